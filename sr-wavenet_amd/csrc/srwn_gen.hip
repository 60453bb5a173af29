// Queue-cached incremental (autoregressive) generation for the mu-law softmax teacher and for the conditioned
// mixture-of-logistics decoder of WaveNetAutoEncoder (the teacher generator.py:150-170 / teacher.py:140-171 sample
// from with a whole-clip pass per sample).
//
// The reference has no fast generator: its only sampler re-runs the whole clip once per
// generated sample (teacher.py:140-171, O(T^2 L)).  This kernel keeps, per layer, a ring of the last
// d_l + 1 layer inputs and produces one sample per step with exactly the arithmetic of the training
// graph (ops.py:23-46, model.py:158-196 with RightShift), so step t's logits equal the full forward's
// logits[:, t] on the same prefix -- the property the parity test pins.
//
// One persistent workgroup (4 waves) serves a group of up to 32 utterances (grid = number of groups): the 32 utterances are the 32 columns of
// the MFMA tiles, so a step costs the same MFMAs for 1 or 32 voices.  Per layer every wave runs the tiny
// conv -> gate -> residual chain redundantly in registers (no intra-layer exchange) and owns one quarter
// (64 channels) of the skip / head products; conv+residual weights of layer l+1 stream into LDS by LDS-DMA
// while layer l computes; skip/head weights are read straight from L2 (each wave uses distinct rows).
#include <cstdlib>
#include <type_traits>
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

constexpr int kGenMaxLayers = 64;

struct GenArgs {
  const void* wcr;      // per layer: [conv image RT x 2KS (tap0 natural, tap1 permuted) | res image RT x KS (permuted)]
  const void* wskip;    // [S/32][L*R/16] permuted k order (B operand = gate tile in registers)
  const void* w1;       // [S/32][S/16] natural
  const void* w2;       // [Cp/32][S/16] natural
  const float* bias_f; const float* bias_r;   // [L][R]
  const float* bs_sum; const float* b1; const float* b2;   // [S], [S], [Cp]
  const float* init_w; const float* init_b;   // [2][R], [R]
  void* ring;           // layer input rings, element offsets ring_off[l], depth dil[l]+1 slots of [32][R]
  float* audio_out; int32_t* codes_out; float* logits_out; const float* forced;
  int B, Tout, nsteps, L, C, mode, Q;
  // conditioning (model.py:180-183): cond [B*frames, cond_ld] holds cb of every layer at columns [l*R, (l+1)*R);
  // layer l adds row (u, t / pool) to its input, rounded to T like the training kernel.  NULL = unconditioned.
  const void* cond; int cond_frames; int pool; long long cond_ld;
  int M;                // > 0: mixture-of-logistics head with M mixtures (C = 4M logits) instead of the softmax
  long long ring_group_elems;
  unsigned long long seed;
  int dil[kGenMaxLayers];
  long long ring_off[kGenMaxLayers];
};

__device__ __forceinline__ float gen_mu_law_decode(int code, int Q) {   // ops.py:96-104, as srwn_mu_law_decode
  const float mu = (float)(Q - 1);
  const float signal = __fadd_rn(__fmul_rn(2.0f, __fdiv_rn((float)code, mu)), -1.0f);
  const float p = (float)pow((double)Q, (double)fabsf(signal));
  const float magnitude = __fmul_rn((float)(1.0 / (double)(Q - 1)), __fadd_rn(p, -1.0f));
  const float sgn = (signal > 0.0f) ? 1.0f : ((signal < 0.0f) ? -1.0f : 0.0f);
  return __fmul_rn(sgn, magnitude);
}

__device__ __forceinline__ float gen_uniform(unsigned long long seed, unsigned u, unsigned t) {
  unsigned long long x = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)u * 0x100000001ull + t + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return (float)((x >> 40) + 0.5) * (1.0f / 16777216.0f);   // (0,1)
}

template <typename T, int RT> struct GenCond { f32x4 cc[RT][4]; };
struct GenNoCond {};

template <typename T, int NBUF, bool COND, int RT, int SS>
__global__ __launch_bounds__(256) void generate_kernel(GenArgs a) {
  constexpr int R = 32 * RT, KS = R / 16, S = SS, SQ = S / 4;   // SQ: skip/head-1 channels per wave
  constexpr int MQ = SQ / 32;                                    // ... = MQ 32-row tiles per wave
  constexpr int LGS = 256;                                       // row stride of the logits exchange (C <= 256)
  constexpr int FB = sizeof(Frag<T>) * 64;
  constexpr int LAYER_FR = RT * 2 * KS + RT * KS;                // 24 fragment images per layer (conv + res)
  constexpr int LAYER_B = LAYER_FR * FB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wbuf = smem;                                             // [NBUF][LAYER_B]
  T* xch = reinterpret_cast<T*>(smem + NBUF * LAYER_B);          // [32][S] activation exchange (r0 / r1)
  float* lgl = reinterpret_cast<float*>(xch + 32 * S);           // [32][LGS] logits
  float* prev = lgl + 32 * LGS;                                  // [2][32] last two samples
  float* cst = prev + 64;                                        // constants: biases of every layer + head + input conv
  float* c_bf = cst;                 // [L][R]
  float* c_br = c_bf + a.L * R;      // [L][R]
  float* c_bs = c_br + a.L * R;      // [S]
  float* c_b1 = c_bs + S;            // [S]
  float* c_b2 = c_b1 + S;            // [LGS]
  float* c_iw = c_b2 + LGS;          // [2][R]
  float* c_ib = c_iw + 2 * R;        // [R]
  float* c_dec = c_ib + R;           // [256] mu-law decode of every code (ops.py:96-104 has a pow(): one table per launch)

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int u0 = blockIdx.x * 32;                                // first utterance of this workgroup's group
  const int ug = u0 + col;                                       // this lane's utterance
  const bool uok = ug < a.B;
  const char* wcr = reinterpret_cast<const char*>(a.wcr);
  const Frag<T>* wskip = reinterpret_cast<const Frag<T>*>(a.wskip);
  const Frag<T>* w1 = reinterpret_cast<const Frag<T>*>(a.w1);
  const Frag<T>* w2 = reinterpret_cast<const Frag<T>*>(a.w2);
  const int ks_skip = a.L * KS;
  T* ring = reinterpret_cast<T*>(a.ring) + (size_t)blockIdx.x * a.ring_group_elems;   // one ring set per group
  int par = 0;   // which weight buffer holds the layer being computed (toggles every layer, across steps)

  for (int i = threadIdx.x; i < a.L * R; i += 256) { c_bf[i] = a.bias_f[i]; c_br[i] = a.bias_r[i]; }
  for (int i = threadIdx.x; i < S; i += 256) { c_bs[i] = a.bs_sum[i]; c_b1[i] = a.b1[i]; }
  for (int i = threadIdx.x; i < LGS; i += 256)
    c_b2[i] = (i < (a.C + 31) / 32 * 32) ? a.b2[i] : 0.0f;     // the last 1x1 has ceil(C/32)*32 rows
  if (threadIdx.x < 2 * R) c_iw[threadIdx.x] = a.init_w[threadIdx.x];
  if (threadIdx.x < R) c_ib[threadIdx.x] = a.init_b[threadIdx.x];
  if (threadIdx.x < 64) prev[threadIdx.x] = 0.0f;
  if (a.Q >= 2) c_dec[threadIdx.x] = gen_mu_law_decode(threadIdx.x < a.Q ? threadIdx.x : a.Q - 1, a.Q);
  lds_dma_copy(wcr, wbuf, LAYER_B, wave, lane, 4);
  __syncthreads();

  // operands of one layer that do not depend on the current step's activations: the tap-0 window from
  // the ring (written d >= 1 steps ago) and this wave's skip-weight fragments (from L2).  Loaded two
  // layers ahead, unconditionally (clamped), so their latency hides behind the dependent MFMA chain.
  // (the conditioning operands exist only in the COND instantiation: they cost 32 VGPRs per operand set)
  struct Pre : std::conditional<COND, GenCond<T, RT>, GenNoCond>::type { Frag<T> xd[KS]; Frag<T> ws[MQ][KS]; };
  const T* condp = COND ? reinterpret_cast<const T*>(a.cond) : nullptr;
  const int ucl = uok ? ug : (a.B - 1);                          // clamped utterance for conditioning loads
  auto preload = [&](int l_, int t, Pre& p) {
    const int l = l_ < a.L ? l_ : a.L - 1;
    const int d = a.dil[l], depth = d + 1;
    const int td = t - d;
    const int slot = (td >= 0 ? td : 0) % depth;
    const T* rp = ring + a.ring_off[l] + ((size_t)slot * 32 + col) * R + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) p.xd[ks] = load_nat(rp + 16 * ks);
    if constexpr (COND) {   // cb_l of the current frame (accumulator layout); the ring holds conditioned inputs
      const int fc = min(t / a.pool, a.cond_frames - 1);
      const T* ccp = condp + ((size_t)ucl * a.cond_frames + fc) * a.cond_ld + (size_t)l * R + 4 * half;
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) p.cc[mt][g] = load4(ccp + 32 * mt + 8 * g);
    }
#pragma unroll
    for (int m = 0; m < MQ; ++m)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) p.ws[m][ks] = wskip[((size_t)(MQ * wave + m) * ks_skip + l * KS + ks) * 64 + lane];
  };

  for (int t = 0; t < a.nsteps; ++t) {
    // ---- input conv with RightShift (model.py:172-173): h0[t] = w[0]*audio[t-2] + w[1]*audio[t-1] + b
    float a1 = 0.0f, a2 = 0.0f;
    if (uok) {
      if (a.forced) {
        if (t >= 1) a1 = a.forced[(size_t)ug * a.Tout + t - 1];
        if (t >= 2) a2 = a.forced[(size_t)ug * a.Tout + t - 2];
      } else {
        a1 = prev[col];
        a2 = prev[32 + col];
      }
    }
    float h[RT][16];
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = 32 * mt + crow(q, half);
        h[mt][q] = fmaf(c_iw[n], a2, fmaf(c_iw[R + n], a1, c_ib[n]));
      }
    f32x16 accS[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m)
#pragma unroll
      for (int q = 0; q < 16; ++q) accS[m][q] = c_bs[SQ * wave + 32 * m + crow(q, half)];

    auto layer = [&](int l, const Pre& p, Pre& pfill, int lfill) {
      const int d = a.dil[l];
      const int depth = d + 1;
      const int buf = (NBUF == 2) ? par : 0;
      if (NBUF == 2) {   // stream the next layer's (or next step's first layer's) conv+res weights
        const int ln = (l + 1 < a.L) ? l + 1 : 0;
        lds_dma_copy(wcr + (size_t)ln * LAYER_B, wbuf + (par ^ 1) * LAYER_B, LAYER_B, wave, lane, 4);
        par ^= 1;
      }
      preload(lfill, t, pfill);   // issued AFTER the LDS-DMA so the counted wait below leaves it in flight
      Frag<T> xd[KS];
      const bool tap0 = (t - d) >= 0;   // zero before the clip starts
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xd[ks] = tap0 ? p.xd[ks] : zero_frag<T>();
      if constexpr (COND) {
        // the layer's complete input = output of the layer below + cb_l, rounded once (srwn_residual_layer_fwd adds
        // the next layer's bias before storing); below layer 0 the input conv's output was stored (rounded) first
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float base = (l == 0) ? (float)(T)h[mt][q] : h[mt][q];
            h[mt][q] = base + p.cc[mt][q >> 2][q & 3];
          }
      }
      // x_l[t] -> ring (one writer), and as the permuted-order B fragments of tap 1
      Frag<T> xc[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) xc[s].set(j, h[s >> 1][8 * (s & 1) + j]);
      if (wave == 0) {   // one writer per ring slot
        T* wp = ring + a.ring_off[l] + ((size_t)(t % depth) * 32 + col) * R;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            store4(wp + 32 * mt + 8 * g + 4 * half, h[mt][4 * g], h[mt][4 * g + 1], h[mt][4 * g + 2], h[mt][4 * g + 3]);
      }
      // in bf16 mode the residual operand is the rounded activation the training graph stored
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) h[mt][q] = xc[2 * mt + (q >> 3)].get(q & 7);

      const Frag<T>* lw = reinterpret_cast<const Frag<T>*>(wbuf + buf * LAYER_B) + lane;
      f32x16 accF[RT];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) accF[mt][q] = c_bf[l * R + 32 * mt + crow(q, half)];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          mma(accF[mt], lw[(mt * 2 * KS + ks) * 64], xd[ks]);
          mma(accF[mt], lw[(mt * 2 * KS + KS + ks) * 64], xc[ks]);
        }
      Frag<T> cf[KS];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float z = Math<T>::tanh_(accF[mt][q]);
          z = (float)(T)z;   // the training graph stores z in T and rebuilds the gate from it
          cf[2 * mt + (q >> 3)].set(q & 7, gate_of_z<T>(z));
        }
      f32x16 accR[RT];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) accR[mt][q] = c_br[l * R + 32 * mt + crow(q, half)];
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) mma(accR[mt], lw[(RT * 2 * KS + mt * KS + s) * 64], cf[s]);
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) h[mt][q] = (h[mt][q] + accR[mt][q]) * kSqrtHalf;
      // this wave's quarter of the skip 1x1 (ops.py:44), accumulated over layers (model.py:50)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < MQ; ++m) mma(accS[m], p.ws[m][ks], cf[ks]);
      __syncthreads();   // next layer's weights landed; ring write of this layer ordered before later reads
      if (NBUF == 1) {
        const int ln = (l + 1 < a.L) ? l + 1 : 0;
        lds_dma_copy(wcr + (size_t)ln * LAYER_B, wbuf, LAYER_B, wave, lane, 4);
        __syncthreads();
      }
    };

    // three rotating operand sets: layer l computes while l+1 and l+2 are in flight
    Pre p0, p1, p2;
    preload(0, t, p0);
    preload(1, t, p1);
    for (int l = 0; l < a.L; l += 3) {
      layer(l, p0, p2, l + 2);
      if (l + 1 >= a.L) break;
      layer(l + 1, p1, p0, l + 3);
      if (l + 2 >= a.L) break;
      layer(l + 2, p2, p1, l + 4);
    }

    // ---- head: relu(sum skip) -> 1x1 + relu -> 1x1 (model.py:51-56); quarters exchanged through LDS
#pragma unroll
    for (int m = 0; m < MQ; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store4(xch + col * S + SQ * wave + 32 * m + 8 * g + 4 * half, fmaxf(accS[m][4 * g], 0.f),
               fmaxf(accS[m][4 * g + 1], 0.f), fmaxf(accS[m][4 * g + 2], 0.f), fmaxf(accS[m][4 * g + 3], 0.f));
    __syncthreads();
    f32x16 acc1[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc1[m][q] = c_b1[SQ * wave + 32 * m + crow(q, half)];
#pragma unroll
    for (int ks = 0; ks < S / 16; ++ks) {
      const Frag<T> bf = load_nat(xch + col * S + 16 * ks + 8 * half);
#pragma unroll
      for (int m = 0; m < MQ; ++m) mma(acc1[m], w1[((size_t)(MQ * wave + m) * (S / 16) + ks) * 64 + lane], bf);
    }
    __syncthreads();   // everyone has read r0
#pragma unroll
    for (int m = 0; m < MQ; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store4(xch + col * S + SQ * wave + 32 * m + 8 * g + 4 * half, fmaxf(acc1[m][4 * g], 0.f),
               fmaxf(acc1[m][4 * g + 1], 0.f), fmaxf(acc1[m][4 * g + 2], 0.f), fmaxf(acc1[m][4 * g + 3], 0.f));
    __syncthreads();
    f32x16 acc2[2];
    const int cp_pad = (a.C + 31) / 32 * 32;   // rows of the last 1x1's image (256 for the softmax head, 4M padded for MoL)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc2[m][q] = c_b2[64 * wave + 32 * m + crow(q, half)];   // 2 class tiles per wave
#pragma unroll
    for (int ks = 0; ks < S / 16; ++ks) {
      const Frag<T> bf = load_nat(xch + col * S + 16 * ks + 8 * half);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        if (32 * (2 * wave + m) < cp_pad) mma(acc2[m], w2[((size_t)(2 * wave + m) * (S / 16) + ks) * 64 + lane], bf);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(lgl + col * LGS + 64 * wave + 32 * m + 8 * g + 4 * half) =
            f32x4{acc2[m][4 * g], acc2[m][4 * g + 1], acc2[m][4 * g + 2], acc2[m][4 * g + 3]};
    __syncthreads();

    if (a.M > 0) {
      // ---- mixture-of-logistics head (model.py:196-198): sample_from_discretized_mix_logistic (ops.py:178-201) with
      //      counter-based uniforms; lanes = utterances (M <= 16 mixtures: a short serial loop)
      if (wave == 0 && half == 0) {
        const float* l = lgl + col * LGS;
        int sel = 0;
        float best = -INFINITY;
        for (int m = 0; m < a.M; ++m) {
          const float u1 = 1e-5f + (1.0f - 2e-5f) * gen_uniform(a.seed, (unsigned)ug, (unsigned)(t * (a.M + 1) + m));
          const float v = l[m] - logf(-logf(u1));
          if (v > best) { best = v; sel = m; }
        }
        float smp = l[a.M + sel];                                  // mode 0: the selected mean (no logistic noise)
        if (a.mode == 1) {
          const float u2 = 1e-5f + (1.0f - 2e-5f) * gen_uniform(a.seed, (unsigned)ug, (unsigned)(t * (a.M + 1) + a.M));
          smp += expf(fmaxf(l[2 * a.M + sel], -7.0f)) * (logf(u2) - logf(1.0f - u2));
        }
        smp = fminf(fmaxf(smp, -1.0f), 1.0f);
        if (uok) {
          a.audio_out[(size_t)ug * a.Tout + t] = smp;
          a.codes_out[(size_t)ug * a.Tout + t] = sel;
        }
        prev[32 + col] = prev[col];
        prev[col] = smp;
      }
      if (a.logits_out) {
        for (int i = threadIdx.x; i < 32 * a.C; i += 256) {
          const int ul = i / a.C, c = i - ul * a.C;
          if (u0 + ul < a.B) a.logits_out[((size_t)(u0 + ul) * a.Tout + t) * a.C + c] = lgl[ul * LGS + c];
        }
      }
      __syncthreads();
      continue;
    }
    // ---- softmax over the C classes, pick a code, mu-law decode: wave w serves utterances 8w..8w+7 with
    //      lanes = classes (4 per lane: conflict-free LDS rows, shuffle reductions instead of serial loops)
    for (int i = 0; i < 8; ++i) {
      const int ul = 8 * wave + i;                        // wave-uniform, local to the group
      const int u = u0 + ul;
      const f32x4 v = *reinterpret_cast<const f32x4*>(lgl + ul * LGS + 4 * lane);
      float m = -INFINITY; int am = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * lane + e < a.C && v[e] > m) { m = v[e]; am = 4 * lane + e; }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float mo = __shfl_xor(m, off); const int ao = __shfl_xor(am, off);
        if (mo > m || (mo == m && ao < am)) { m = mo; am = ao; }
      }
      int code = am;
      if (a.mode == 1) {   // categorical sample from softmax(logits): inclusive prefix sums over the lanes
        float ev[4], loc = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ev[e] = (4 * lane + e < a.C) ? __expf(v[e] - m) : 0.0f; loc += ev[e]; }
        float inc = loc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float o = __shfl_up(inc, off);
          if (lane >= off) inc += o;
        }
        const float total = __shfl(inc, 63);
        const float target = gen_uniform(a.seed, (unsigned)u, (unsigned)t) * total;
        const unsigned long long hit = __ballot(inc > target);
        const int src = hit ? (__ffsll((long long)hit) - 1) : 63;
        float run = inc - loc;
        int pick = 4 * lane + 3;
#pragma unroll
        for (int e = 3; e >= 0; --e) { if (run + ev[0] + (e > 0 ? ev[1] : 0.f) + (e > 1 ? ev[2] : 0.f) + (e > 2 ? ev[3] : 0.f) > target) pick = 4 * lane + e; }
        if (pick >= a.C) pick = a.C - 1;
        code = __shfl(pick, src);
      }
      if (lane == 0) {
        const float smp = c_dec[code];
        if (u < a.B) {
          a.audio_out[(size_t)u * a.Tout + t] = smp;
          a.codes_out[(size_t)u * a.Tout + t] = code;
        }
        prev[32 + ul] = prev[ul];
        prev[ul] = smp;
      }
      if (a.logits_out && u < a.B && 4 * lane < a.C) {
        float* lo = a.logits_out + ((size_t)u * a.Tout + t) * a.C + 4 * lane;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * lane + e < a.C) lo[e] = v[e];
      }
    }
    __syncthreads();
  }
}

extern "C" int64_t srwn_generate_ring_elems(const int32_t* dilations, int32_t nlayers, int32_t R) {
  int64_t n = 0;
  for (int l = 0; l < nlayers; ++l) n += (int64_t)(dilations[l] + 1) * 32 * R;
  return n;   // per group of 32 utterances
}

static int generate_impl(const void* wcr, const void* wskip, const void* w1, const void* w2, const float* bias_f,
                         const float* bias_r, const float* bs_sum, const float* b1, const float* b2,
                         const float* init_w, const float* init_b, void* ring, float* audio_out, int32_t* codes_out,
                         float* logits_out, const float* forced, const int32_t* dilations, int32_t nlayers, int32_t B,
                         int32_t Tout, int32_t nsteps, int32_t R, int32_t S, int32_t C, int32_t K, int32_t mode,
                         uint64_t seed, int32_t dtype, void* stream, const void* cond, int32_t cond_frames,
                         int32_t pool, int64_t cond_ld, int32_t M) {
  if (B == 0 || nsteps == 0) return 0;
  if (!wcr || !wskip || !w1 || !w2 || !bias_f || !bias_r || !bs_sum || !b1 || !b2 || !init_w || !init_b || !ring ||
      !audio_out || !codes_out || !dilations)
    return set_error(SRWN_E_NULL, "generate: null pointer");
  if ((R != 64 && R != 32) || (S != 256 && S != 128) || K != 2 || C < 2 || C > 256)
    return set_error(SRWN_E_UNSUPPORTED, "generate: built for R=64 or 32, S=256 or 128, K=2, C<=256 (got R=%d S=%d K=%d C=%d)", R, S, K, C);
  if (B < 0 || nsteps < 0 || nsteps > Tout || nlayers < 1 || nlayers > kGenMaxLayers || (mode != 0 && mode != 1))
    return set_error(SRWN_E_SHAPE, "generate: B=%d nsteps=%d Tout=%d layers=%d mode=%d", B, nsteps, Tout, nlayers, mode);
  GenArgs a;
  a.wcr = wcr; a.wskip = wskip; a.w1 = w1; a.w2 = w2; a.bias_f = bias_f; a.bias_r = bias_r; a.bs_sum = bs_sum;
  a.b1 = b1; a.b2 = b2; a.init_w = init_w; a.init_b = init_b; a.ring = ring; a.audio_out = audio_out;
  a.codes_out = codes_out; a.logits_out = logits_out; a.forced = forced;
  a.B = B; a.Tout = Tout; a.nsteps = nsteps; a.L = nlayers; a.C = C; a.mode = mode; a.Q = C; a.seed = seed;
  a.cond = cond; a.cond_frames = cond_frames; a.pool = pool; a.cond_ld = cond_ld; a.M = M;
  long long off = 0;
  for (int l = 0; l < kGenMaxLayers; ++l) {
    a.dil[l] = (l < nlayers) ? dilations[l] : 1;
    a.ring_off[l] = off;
    if (l < nlayers) {
      if (dilations[l] < 1) return set_error(SRWN_E_SHAPE, "generate: dilation %d", dilations[l]);
      off += (long long)(dilations[l] + 1) * 32 * R;
    }
  }
  a.ring_group_elems = off;
  const unsigned groups = (unsigned)((B + 31) / 32);
  hipStream_t st = (hipStream_t)stream;
  const size_t lfr = (size_t)(R / 32) * 3 * (R / 16);   // fragment images per layer: conv RT x 2KS + residual RT x KS
  // widths: (64, 256) the north-star stack, (32, 256) generator.py's default teacher, (32, 128) teacher.py's
#define SRWN_GEN_PICK(TT, NB)                                                                                   \
  ((R == 64 && S == 256) ? (cond ? generate_kernel<TT, NB, true, 2, 256> : generate_kernel<TT, NB, false, 2, 256>)  \
   : (R == 32 && S == 256) ? (cond ? generate_kernel<TT, NB, true, 1, 256> : generate_kernel<TT, NB, false, 1, 256>) \
   : (R == 32 && S == 128) ? (cond ? generate_kernel<TT, NB, true, 1, 128> : generate_kernel<TT, NB, false, 1, 128>) \
                           : (cond ? generate_kernel<TT, NB, true, 2, 128> : generate_kernel<TT, NB, false, 2, 128>))
  if (dtype == SRWN_BF16) {
    auto kfn = SRWN_GEN_PICK(bf16_t, 2);
    const size_t sh = 2 * lfr * sizeof(Frag<bf16_t>) * 64 + 32 * S * sizeof(bf16_t) + 32 * 256 * 4 + 64 * 4 +
                      (size_t)(2 * nlayers * R + 2 * S + 256 + 3 * R + 256) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return set_error((int)e, "generate: LDS %zu: %s", sh, hipGetErrorString(e));
    hipLaunchKernelGGL(kfn, dim3(groups), dim3(256), sh, st, a);
  } else if (dtype == SRWN_F32) {
    auto kfn = SRWN_GEN_PICK(float, 1);
    const size_t sh = 1 * lfr * sizeof(Frag<float>) * 64 + 32 * S * sizeof(float) + 32 * 256 * 4 + 64 * 4 +
                      (size_t)(2 * nlayers * R + 2 * S + 256 + 3 * R + 256) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return set_error((int)e, "generate: LDS %zu: %s", sh, hipGetErrorString(e));
    hipLaunchKernelGGL(kfn, dim3(groups), dim3(256), sh, st, a);
  } else {
    return set_error(SRWN_E_DTYPE, "generate: dtype %d", dtype);
  }
  return check_launch("generate");
}

extern "C" int srwn_generate(const void* wcr, const void* wskip, const void* w1, const void* w2, const float* bias_f,
                             const float* bias_r, const float* bs_sum, const float* b1, const float* b2,
                             const float* init_w, const float* init_b, void* ring, float* audio_out,
                             int32_t* codes_out, float* logits_out, const float* forced, const int32_t* dilations,
                             int32_t nlayers, int32_t B, int32_t Tout, int32_t nsteps, int32_t R, int32_t S,
                             int32_t C, int32_t K, int32_t mode, uint64_t seed, int32_t dtype, void* stream) {
  return generate_impl(wcr, wskip, w1, w2, bias_f, bias_r, bs_sum, b1, b2, init_w, init_b, ring, audio_out, codes_out,
                       logits_out, forced, dilations, nlayers, B, Tout, nsteps, R, S, C, K, mode, seed, dtype, stream,
                       nullptr, 1, 1, 0, 0);
}

// The conditioned mixture-of-logistics decoder (WaveNetAutoEncoder.createDecoder, model.py:158-200): cond
// [B*cond_frames, cond_ld] = the per-layer conditioning biases cb_l at columns [l*R, (l+1)*R) (srwn_pw_linear of
// encoding_w_condition, model.py:180); head = 4*num_mixtures logits, sampled as ops.py:178-201.  b2 and the w2
// image cover ceil(4M/32)*32 rows.  codes_out receives the selected mixture index.
extern "C" int srwn_generate_mol(const void* wcr, const void* wskip, const void* w1, const void* w2,
                                 const float* bias_f, const float* bias_r, const float* bs_sum, const float* b1,
                                 const float* b2, const float* init_w, const float* init_b, void* ring,
                                 float* audio_out, int32_t* codes_out, float* logits_out, const float* forced,
                                 const int32_t* dilations, int32_t nlayers, int32_t B, int32_t Tout, int32_t nsteps,
                                 int32_t R, int32_t S, int32_t K, int32_t num_mixtures, const void* cond,
                                 int32_t cond_frames, int32_t pool_stride, int64_t cond_ld, int32_t mode, uint64_t seed,
                                 int32_t dtype, void* stream) {
  if (num_mixtures < 1 || num_mixtures > 16)
    return set_error(SRWN_E_SHAPE, "generate_mol: num_mixtures=%d (1..16)", num_mixtures);
  if (cond && (cond_frames < 1 || pool_stride < 1 || cond_ld < (int64_t)nlayers * R))
    return set_error(SRWN_E_SHAPE, "generate_mol: cond_frames=%d pool_stride=%d cond_ld=%lld", cond_frames, pool_stride,
                     (long long)cond_ld);
  return generate_impl(wcr, wskip, w1, w2, bias_f, bias_r, bs_sum, b1, b2, init_w, init_b, ring, audio_out, codes_out,
                       logits_out, forced, dilations, nlayers, B, Tout, nsteps, R, S, 4 * num_mixtures, K, mode, seed,
                       dtype, stream, cond, cond ? cond_frames : 1, cond ? pool_stride : 1, cond_ld, num_mixtures);
}
