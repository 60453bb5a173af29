// Queue-cached incremental generation, latency-optimised body for the benchmark's teacher (bf16, 64 residual / 256 skip
// channels, mu-law softmax head, unconditioned): the same arithmetic and the same rings as generate_kernel
// (srwn_gen.hip; the reference only has the whole-clip-per-sample loop of teacher.py:140-171 / generator.py:150-170),
// organised for the length of the DEPENDENT chain of one step instead of for throughput.
//
// generate_kernel runs the conv -> tanh -> gate -> 1x1 chain of a layer redundantly in each of its four waves (no
// exchange inside a layer) on 32x32x16 tiles: 32 MFMAs and the transcendental work of all 64 channels per wave and
// layer, every MFMA behind an LDS read of its weight fragment: 3.7 us per layer, real-time factor 1.8 for one stream.
// Here a layer's 64 channels are SPLIT over the four waves (16 each) on v_mfma_f32_16x16x32_bf16 tiles -- the 32
// utterances of a workgroup are two 16-column blocks -- so a wave runs 8 + 4 MFMAs of 16 cycles and the tanh / gate of 8
// values per lane on the chain; the gate output and the layer output cross the waves through two 4.5-KB LDS tiles (two
// barriers per layer); every weight fragment a wave needs (its 16 rows of the conv and residual kernels, its 64 rows of
// the skip kernel: 14 KB per layer) is requested from L2 one layer ahead straight into registers (one wave per SIMD: 512
// registers), so no MFMA waits for LDS or memory; the skip products run in the matrix pipe behind the residual ones
// while the VALU finishes the layer.  Pinned by the same tests as the throughput kernel (tests/test_gpu_generate.py).
#include <cstdlib>
#include <type_traits>
#include "srwn_common.h"
#include "srwn_group.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;
using namespace srwn::grp;

namespace {

constexpr int kG16MaxLayers = 64;

struct Gen16Args {
  const void* wl;       // per layer: [4 waves][conv ks 0..3 | res ks 0..1 | skip (rb 0..3) x (ks 0..1)] 16x32 A fragments
  const void* wh1;      // head 1x1 S -> S: [4 waves][rb 0..3][ks 0..7]
  const void* wh2;      // head 1x1 S -> C (rows >= C zero): same shape
  const float* bias_f; const float* bias_r;   // [L][R]
  const float* bs_sum; const float* b1; const float* b2;   // [S], [S], [256]
  const float* init_w; const float* init_b;   // [2][R], [R]
  void* ring;
  float* audio_out; int32_t* codes_out; float* logits_out; const float* forced;
  int B, Tout, nsteps, L, C, Cp, mode, Q;       // Cp: entries of b2 / rows of the last 1x1 (ceil(C/32)*32)
  int M;                                        // > 0: mixture-of-logistics head with M mixtures (C = 4M)
  const void* cond; int cond_frames, pool; long long cond_ld;   // conditioning biases cb_l (COND instantiation)
  long long ring_group_elems;
  unsigned long long seed;
  int dil[kG16MaxLayers];
  long long ring_off[kG16MaxLayers];
};

__device__ __forceinline__ float g16_mu_law_decode(int code, int Q) {   // ops.py:96-104, as srwn_mu_law_decode
  const float mu = (float)(Q - 1);
  const float signal = __fadd_rn(__fmul_rn(2.0f, __fdiv_rn((float)code, mu)), -1.0f);
  const float p = (float)pow((double)Q, (double)fabsf(signal));
  const float magnitude = __fmul_rn((float)(1.0 / (double)(Q - 1)), __fadd_rn(p, -1.0f));
  const float sgn = (signal > 0.0f) ? 1.0f : ((signal < 0.0f) ? -1.0f : 0.0f);
  return __fmul_rn(sgn, magnitude);
}

__device__ __forceinline__ float g16_uniform(unsigned long long seed, unsigned u, unsigned t) {   // as gen_uniform (srwn_gen.hip)
  unsigned long long x = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)u * 0x100000001ull + t + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return (float)((x >> 40) + 0.5) * (1.0f / 16777216.0f);
}

typedef bf16_t T;
constexpr int LGS = 256;
constexpr int FR = 512;                       // elements of one fragment image (64 lanes x 8)
// widths: R residual channels (64; 32 = the reference scripts' dilation_channels: only two waves have channels of the
// layer chain, all four share the skip and head rows), S skip channels (256 or 128)
template <int R, int S> struct G16W {
  static constexpr int KR = R / 32;           // k-steps of one tap / of the residual 1x1
  static constexpr int KC = 2 * KR;           // k-steps of the conv: [delayed tap | current tap]
  static constexpr int SRB = S / 64;          // 16-row blocks of the skip 1x1 / of the first head 1x1 per wave
  static constexpr int HKS = S / 32;          // k-steps of the head 1x1s
  static constexpr int CW = R / 16;           // waves that own channels of the layer chain
  static constexpr int WFR = KC + KR + SRB * KR;   // fragments of one layer per wave
  static constexpr int LSX = R + 8, LSH = S + 8;
};

// one layer's operands of a wave that do not depend on the step's activations: its weight fragments and the delayed tap
// NCB: 16-utterance column blocks per workgroup (2: one workgroup per ring group of 32 utterances; 1: two workgroups share
// a ring group, each half the work per layer -- the choice whenever the utterances do not fill the chip's CUs otherwise)
struct G16NoCond {};
template <int NCB> struct G16Cond { bf16x4 cc[NCB]; };   // cb_l of the current frame: this wave's channels 16w + 4rq..
template <int NCB, bool COND, int R, int S>
struct PreT : std::conditional<COND, G16Cond<NCB>, G16NoCond>::type {
  Frag<T> wc[G16W<R, S>::KC], wr[G16W<R, S>::KR], ws[G16W<R, S>::SRB][G16W<R, S>::KR];
  Frag<T> x0[G16W<R, S>::KR][NCB];            // [k-step of the delayed tap][column block]
};

template <int NCB, bool COND, bool MOL, int R, int S>
__global__ __launch_bounds__(256) void generate16_kernel(Gen16Args a) {
  constexpr int NU = 16 * NCB;                // utterances of this workgroup
  constexpr int NI = 4 * NCB;                 // utterances a wave samples
  using W = G16W<R, S>;
  constexpr int KR = W::KR, KC = W::KC, SRB = W::SRB, HKS = W::HKS, LSX = W::LSX, LSH = W::LSH;
  using Pre = PreT<NCB, COND, R, S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xb = reinterpret_cast<T*>(smem);                         // [32][LSX] the layer input x_l[t]
  T* cb = xb + 32 * LSX;                                      // [32][LSX] the gate output c_l[t]
  T* hx = cb + 32 * LSX;                                      // [32][LSH] head activations (r0 / r1)
  float* lgl = reinterpret_cast<float*>(hx + 32 * LSH);       // [32][LGS] logits
  float* prev = lgl + 32 * LGS;                               // [2][32] last two samples
  float* c_bf = prev + 64;                                    // [L][R]
  float* c_br = c_bf + a.L * R;                               // [L][R]
  float* c_bs = c_br + a.L * R;                               // [S]
  float* c_b1 = c_bs + S;                                     // [S]
  float* c_b2 = c_b1 + S;                                     // [256]
  float* c_iw = c_b2 + 256;                                   // [2][R]
  float* c_ib = c_iw + 2 * R;                                 // [R]
  float* c_dec = c_ib + R;                                    // [256] mu-law decode of every code (one pow() each, once)

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, rq = lane >> 4;                  // D tile: column (utterance in its block), rows 4 rq + r
  const int u0 = blockIdx.x * NU;
  const int urow = u0 & 31;                                   // this workgroup's first row inside its ring group
  T* ring = reinterpret_cast<T*>(a.ring) + (size_t)(u0 >> 5) * a.ring_group_elems + urow * R;
  const T* wl = reinterpret_cast<const T*>(a.wl);
  const bool chw = W::CW == 4 || wave < W::CW;                // this wave owns 16 channels of the layer chain

  for (int i = threadIdx.x; i < a.L * R; i += 256) { c_bf[i] = a.bias_f[i]; c_br[i] = a.bias_r[i]; }
  for (int i = threadIdx.x; i < S; i += 256) { c_bs[i] = a.bs_sum[i]; c_b1[i] = a.b1[i]; }
  for (int i = threadIdx.x; i < 256; i += 256) c_b2[i] = i < a.Cp ? a.b2[i] : 0.0f;
  if (threadIdx.x < 2 * R) c_iw[threadIdx.x] = a.init_w[threadIdx.x];
  if (threadIdx.x < R) c_ib[threadIdx.x] = a.init_b[threadIdx.x];
  if (threadIdx.x < 64) prev[threadIdx.x] = 0.0f;
  c_dec[threadIdx.x] = g16_mu_law_decode((int)threadIdx.x < a.Q ? (int)threadIdx.x : a.Q - 1, a.Q);
  __syncthreads();

  // per-layer scalars live in lane-indexed registers (lane l: layer l) and are fetched with v_readlane: the ring depth,
  // the ring's element offset, and the slot t % depth the current step writes (kept by increment: no division per layer)
  const int lyr = lane < a.L ? lane : a.L - 1;
  const int depthv = a.dil[lyr] + 1;
  const int roffv = (int)a.ring_off[lyr];
  int curv = 0;
  auto wrap = [](int x, int depth) { return x >= depth ? x - depth : x; };

  // the operands of layer l at the step whose write slots are curv + ahead: the delayed tap x_l[t - d] sits in slot
  // (t - d) mod (d + 1) = (t + 1) mod (d + 1), the slot after the one the step writes -- which, for t < d, nobody has
  // written yet: the caller hands the rings over zero-filled, and that is the zero padding of the causal conv (ops.py:6-10)
  const T* condp = reinterpret_cast<const T*>(a.cond);
  auto preload = [&](int l, int ahead, int t, Pre& p) {
    if constexpr (COND) {   // (first: the epilogue of the layer below needs it before anything else of this set)
      if (chw) {
      const int tt = t + ahead;
      const int fc = tt / a.pool < a.cond_frames ? tt / a.pool : a.cond_frames - 1;
#pragma unroll
      for (int c2 = 0; c2 < NCB; ++c2) {
        const int ug = u0 + 16 * c2 + col;
        const int uc = ug < a.B ? ug : a.B - 1;
        p.cc[c2] = *reinterpret_cast<const bf16x4*>(condp + ((size_t)uc * a.cond_frames + fc) * a.cond_ld + (size_t)l * R + 16 * wave + 4 * rq);
      }
      }
    }
    const T* w = wl + ((size_t)l * 4 + wave) * (W::WFR * FR) + lane * 8;
#pragma unroll
    for (int f = 0; f < KC; ++f) p.wc[f] = load_nat(w + f * FR);
#pragma unroll
    for (int f = 0; f < KR; ++f) p.wr[f] = load_nat(w + (KC + f) * FR);
#pragma unroll
    for (int rb = 0; rb < SRB; ++rb)
#pragma unroll
      for (int ks = 0; ks < KR; ++ks) p.ws[rb][ks] = load_nat(w + (KC + KR + KR * rb + ks) * FR);
    const int depth = __builtin_amdgcn_readlane(depthv, l);
    const int roff = __builtin_amdgcn_readlane(roffv, l);
    int slot = wrap(__builtin_amdgcn_readlane(curv, l) + 1, depth);
    if (ahead) slot = wrap(slot + 1, depth);
    const T* rp = ring + roff + (size_t)slot * (32 * R) + col * R + 8 * rq;
#pragma unroll
    for (int ks = 0; ks < KR; ++ks)
#pragma unroll
      for (int c2 = 0; c2 < NCB; ++c2) p.x0[ks][c2] = load_nat(rp + 16 * c2 * R + 32 * ks);
  };

  Pre pa, pb;
  for (int t = 0; t < a.nsteps; ++t) {
    // (layer 0's operands were requested by the top layer of the step before: its ring slot is at least one step old)
    if (t == 0) preload(0, 0, 0, pa);
    // ---- input conv with RightShift (model.py:172-173): h0[t] = w[0] audio[t-2] + w[1] audio[t-1] + b; this wave's 16 channels
    float xs[NCB][4];                              // the wave's slice of the current layer input (as stored: rounded)
#pragma unroll
    for (int c2 = 0; c2 < NCB; ++c2) {
      const int ug = u0 + 16 * c2 + col;
      float a1 = 0.0f, a2 = 0.0f;
      if (ug < a.B) {
        if (a.forced) {
          if (t >= 1) a1 = a.forced[(size_t)ug * a.Tout + t - 1];
          if (t >= 2) a2 = a.forced[(size_t)ug * a.Tout + t - 2];
        } else {
          a1 = prev[16 * c2 + col];
          a2 = prev[32 + 16 * c2 + col];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = chw ? 16 * wave + 4 * rq + r : 0;
        xs[c2][r] = (float)(T)fmaf(c_iw[n], a2, fmaf(c_iw[R + n], a1, c_ib[n]));
        // (conditioned decoder, model.py:176-189: the layer's input is the output below + cb_l, rounded once more --
        // the input conv's output was stored rounded first, as srwn_residual_layer_fwd sees it)
        if constexpr (COND) xs[c2][r] = (float)(T)(xs[c2][r] + (float)pa.cc[c2][r]);
      }
      if (chw) store4(xb + (size_t)(16 * c2 + col) * LSX + 16 * wave + 4 * rq, xs[c2][0], xs[c2][1], xs[c2][2], xs[c2][3]);
    }
    f32x4 accS[SRB][NCB];
#pragma unroll
    for (int rb = 0; rb < SRB; ++rb)
#pragma unroll
      for (int c2 = 0; c2 < NCB; ++c2)
#pragma unroll
        for (int r = 0; r < 4; ++r) accS[rb][c2][r] = c_bs[16 * SRB * wave + 16 * rb + 4 * rq + r];
    wg_barrier();

    auto layer = [&](int l, const Pre& p, Pre& pnext) {
      // x_l[t] of the workgroup -> the layer's ring (read d steps from now): each wave copies eight utterances' rows
      {
        constexpr int LPRW = R / 8, RPW = 64 / LPRW;          // lanes per row, rows per wave-instruction
        const int ul = RPW * wave + lane / LPRW;
        const int slot = __builtin_amdgcn_readlane(curv, l), roff = __builtin_amdgcn_readlane(roffv, l);
        if (RPW * wave < NU) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)ul * LSX + (lane % LPRW) * 8);
          *reinterpret_cast<f32x4*>(ring + roff + (size_t)slot * (32 * R) + ul * R + (lane % LPRW) * 8) = v;
        }
      }
      // the next layer's weights and delayed tap, one layer ahead (after the top layer: layer 0 of the next step)
      {
        const bool top = l + 1 >= a.L;
        preload(top ? 0 : l + 1, top ? 1 : 0, t, pnext);
      }
      if (chw) {
        f32x4 accF[NCB];
#pragma unroll
        for (int c2 = 0; c2 < NCB; ++c2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) accF[c2][r] = c_bf[l * R + 16 * wave + 4 * rq + r];
        }
#pragma unroll
        for (int ks = 0; ks < KR; ++ks)
#pragma unroll
          for (int c2 = 0; c2 < NCB; ++c2) {
            const Frag<T> x1 = load_nat(xb + (size_t)(16 * c2 + col) * LSX + 32 * ks + 8 * rq);
            mma16(accF[c2], p.wc[KR + ks], x1);
            mma16(accF[c2], p.wc[ks], p.x0[ks][c2]);
          }
        // tanh, gate (ops.py:28-36); z as the training graph stores it
#pragma unroll
        for (int c2 = 0; c2 < NCB; ++c2) {
          float cv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float z = Math<T>::tanh_(accF[c2][r]);
            z = (float)(T)z;
            cv[r] = gate_of_z<T>(z);
          }
          store4(cb + (size_t)(16 * c2 + col) * LSX + 16 * wave + 4 * rq, cv[0], cv[1], cv[2], cv[3]);
        }
      }
      wg_barrier();
      Frag<T> cf[KR][NCB];
#pragma unroll
      for (int ks = 0; ks < KR; ++ks)
#pragma unroll
        for (int c2 = 0; c2 < NCB; ++c2) cf[ks][c2] = load_nat(cb + (size_t)(16 * c2 + col) * LSX + 32 * ks + 8 * rq);
      f32x4 accR[NCB];
      if (chw) {
#pragma unroll
        for (int c2 = 0; c2 < NCB; ++c2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) accR[c2][r] = c_br[l * R + 16 * wave + 4 * rq + r];
        }
#pragma unroll
        for (int ks = 0; ks < KR; ++ks)
#pragma unroll
          for (int c2 = 0; c2 < NCB; ++c2) mma16(accR[c2], p.wr[ks], cf[ks][c2]);
      }
      // this wave's quarter of the skip 1x1 (ops.py:44), accumulated over layers (model.py:50): in the matrix pipe behind
      // the residual products while the VALU finishes the layer
#pragma unroll
      for (int rb = 0; rb < SRB; ++rb)
#pragma unroll
        for (int ks = 0; ks < KR; ++ks)
#pragma unroll
          for (int c2 = 0; c2 < NCB; ++c2) mma16(accS[rb][c2], p.ws[rb][ks], cf[ks][c2]);
      if (chw) {
#pragma unroll
        for (int c2 = 0; c2 < NCB; ++c2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float y = (xs[c2][r] + accR[c2][r]) * kSqrtHalf;
            if constexpr (COND) y += (l + 1 < a.L) ? (float)pnext.cc[c2][r] : 0.0f;   // the next layer's cb, rounded once with it
            xs[c2][r] = (float)(T)y;
          }
          store4(xb + (size_t)(16 * c2 + col) * LSX + 16 * wave + 4 * rq, xs[c2][0], xs[c2][1], xs[c2][2], xs[c2][3]);
        }
      }
      wg_barrier();
    };
    for (int l = 0; l + 1 < a.L; l += 2) {         // the two operand sets alternate
      layer(l, pa, pb);
      layer(l + 1, pb, pa);
    }
    if (a.L & 1) {                                 // an odd stack: the top layer leaves the next step's layer 0 in the other set
      layer(a.L - 1, pa, pb);
      pa = pb;
    }
    curv = wrap(curv + 1, depthv);

    // ---- head: relu(sum skip) -> 1x1 + relu -> 1x1 (model.py:51-56); the activations cross the waves through LDS.
    // First product: wave w owns rows (S/4) w .. of the S outputs.  Second product (up to 256 outputs): wave w owns the
    // 16-row blocks w, w + 4, w + 8, w + 12, so that a head with few outputs (4M mixture parameters: 40 of 256 rows)
    // still splits over the four waves; blocks beyond the padded output count (nrb2 per wave) are skipped.
    Frag<T> hw[4][HKS];                         // the wave's rows of one head 1x1: requested well before their products
    const int nrb2 = MOL ? (a.Cp + 63) / 64 : 4;
    auto head_load1 = [&](const T* wimg) {
#pragma unroll
      for (int rb = 0; rb < SRB; ++rb)
#pragma unroll
        for (int ks = 0; ks < HKS; ++ks) hw[rb][ks] = load_nat(wimg + ((size_t)(wave * SRB + rb) * HKS + ks) * FR + lane * 8);
    };
    auto head_load2 = [&](const T* wimg) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
        if (rb < nrb2) {
#pragma unroll
          for (int ks = 0; ks < HKS; ++ks) hw[rb][ks] = load_nat(wimg + ((size_t)(wave * 4 + rb) * HKS + ks) * FR + lane * 8);
        }
    };
    auto hrow1 = [&](int rb) { return 16 * SRB * wave + 16 * rb + 4 * rq; };
    auto hrow2 = [&](int rb) { return 16 * (4 * rb + wave) + 4 * rq; };    // (the interleaved blocks of the second product)
    auto relu_to_hx = [&](const f32x4 (&acc)[SRB][NCB]) {
#pragma unroll
      for (int rb = 0; rb < SRB; ++rb)
#pragma unroll
        for (int c2 = 0; c2 < NCB; ++c2)
          store4(hx + (size_t)(16 * c2 + col) * LSH + hrow1(rb), fmaxf(acc[rb][c2][0], 0.f),
                 fmaxf(acc[rb][c2][1], 0.f), fmaxf(acc[rb][c2][2], 0.f), fmaxf(acc[rb][c2][3], 0.f));
    };
    head_load1(reinterpret_cast<const T*>(a.wh1));
    relu_to_hx(accS);
    wg_barrier();
    f32x4 acc1[SRB][NCB];
#pragma unroll
    for (int rb = 0; rb < SRB; ++rb)
#pragma unroll
      for (int c2 = 0; c2 < NCB; ++c2)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[rb][c2][r] = c_b1[hrow1(rb) + r];
#pragma unroll
    for (int ks = 0; ks < HKS; ++ks)
#pragma unroll
      for (int c2 = 0; c2 < NCB; ++c2) {
        const Frag<T> bf = load_nat(hx + (size_t)(16 * c2 + col) * LSH + 32 * ks + 8 * rq);
#pragma unroll
        for (int rb = 0; rb < SRB; ++rb) mma16(acc1[rb][c2], hw[rb][ks], bf);
      }
    head_load2(reinterpret_cast<const T*>(a.wh2));   // behind the products, ahead of the exchange
    wg_barrier();                                 // everyone has read r0
    relu_to_hx(acc1);
    wg_barrier();
    f32x4 acc2[4][NCB];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int c2 = 0; c2 < NCB; ++c2)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[rb][c2][r] = c_b2[hrow2(rb) + r];
#pragma unroll
    for (int ks = 0; ks < HKS; ++ks)
#pragma unroll
      for (int c2 = 0; c2 < NCB; ++c2) {
        const Frag<T> bf = load_nat(hx + (size_t)(16 * c2 + col) * LSH + 32 * ks + 8 * rq);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          if (rb < nrb2) mma16(acc2[rb][c2], hw[rb][ks], bf);
      }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int c2 = 0; c2 < NCB; ++c2)
        *reinterpret_cast<f32x4*>(lgl + (size_t)(16 * c2 + col) * LGS + hrow2(rb)) = acc2[rb][c2];
    wg_barrier();

    if constexpr (MOL) {
      // ---- mixture-of-logistics head (model.py:196-198): sample_from_discretized_mix_logistic (ops.py:178-201) with the
      // counter-based uniforms of generate_kernel, lanes = (utterance, mixture): four utterances of 16 mixture slots per
      // pass; Gumbel-max over a 16-lane group (ties to the lower mixture), then the group's first lane draws the sample
#pragma unroll
      for (int ps = 0; ps < NI / 4; ++ps) {
        const int ul = NI * wave + 4 * ps + (lane >> 4);
        const int u = u0 + ul, mx = lane & 15;
        const float* lg = lgl + ul * LGS;
        float v = -INFINITY;
        if (mx < a.M) {
          const float u1 = 1e-5f + (1.0f - 2e-5f) * g16_uniform(a.seed, (unsigned)u, (unsigned)(t * (a.M + 1) + mx));
          v = lg[mx] - logf(-logf(u1));
        }
        int sel = mx;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          const float vo = __shfl_xor(v, off); const int so = __shfl_xor(sel, off);
          if (vo > v || (vo == v && so < sel)) { v = vo; sel = so; }
        }
        if (mx == 0) {
          float smp = lg[a.M + sel];                               // mode 0: the selected mean (no logistic noise)
          if (a.mode == 1) {
            const float u2 = 1e-5f + (1.0f - 2e-5f) * g16_uniform(a.seed, (unsigned)u, (unsigned)(t * (a.M + 1) + a.M));
            smp += expf(fmaxf(lg[2 * a.M + sel], -7.0f)) * (logf(u2) - logf(1.0f - u2));
          }
          smp = fminf(fmaxf(smp, -1.0f), 1.0f);
          if (u < a.B) {
            a.audio_out[(size_t)u * a.Tout + t] = smp;
            a.codes_out[(size_t)u * a.Tout + t] = sel;
          }
          prev[32 + ul] = prev[ul];
          prev[ul] = smp;
        }
      }
      if (a.logits_out) {
        for (int i = threadIdx.x; i < NU * a.C; i += 256) {
          const int ul = i / a.C, c = i - ul * a.C;
          if (u0 + ul < a.B) a.logits_out[((size_t)(u0 + ul) * a.Tout + t) * a.C + c] = lgl[ul * LGS + c];
        }
      }
      wg_barrier();
      continue;
    }
    // ---- softmax over the C classes, pick a code, mu-law decode: wave w serves NI utterances with lanes = classes
    // (4 per lane, as generate_kernel -- same sums in the same order, same counter-based uniforms); the eight utterances'
    // shuffle chains are independent and written round by round so that they overlap
    {
      f32x4 v[NI];
      float m[NI];
      int am[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(lgl + (NI * wave + i) * LGS + 4 * lane);
        m[i] = -INFINITY; am[i] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * lane + e < a.C && v[i][e] > m[i]) { m[i] = v[i][e]; am[i] = 4 * lane + e; }
      }
      int code[NI];
      if (a.mode == 0) {   // argmax, ties to the lower class
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const float mo = __shfl_xor(m[i], off); const int ao = __shfl_xor(am[i], off);
            if (mo > m[i] || (mo == m[i] && ao < am[i])) { m[i] = mo; am[i] = ao; }
          }
#pragma unroll
        for (int i = 0; i < NI; ++i) code[i] = am[i];
      } else {             // categorical sample from softmax(logits): inclusive prefix sums over the lanes
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
          for (int i = 0; i < NI; ++i) m[i] = fmaxf(m[i], __shfl_xor(m[i], off));
        float ev[NI][4], loc[NI], inc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          loc[i] = 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) { ev[i][e] = (4 * lane + e < a.C) ? __expf(v[i][e] - m[i]) : 0.0f; loc[i] += ev[i][e]; }
          inc[i] = loc[i];
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const float o = __shfl_up(inc[i], off);
            if (lane >= off) inc[i] += o;
          }
        // lane j draws the uniform of utterance j & 7 once
        const float uni = g16_uniform(a.seed, (unsigned)(u0 + NI * wave + (lane & (NI - 1))), (unsigned)t);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const float total = __shfl(inc[i], 63);
          const float target = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uni), i)) * total;
          const unsigned long long hit = __ballot(inc[i] > target);
          const int src = hit ? (__ffsll((long long)hit) - 1) : 63;
          const float run = inc[i] - loc[i];
          int pick = 4 * lane + 3;
#pragma unroll
          for (int e = 3; e >= 0; --e) {
            if (run + ev[i][0] + (e > 0 ? ev[i][1] : 0.f) + (e > 1 ? ev[i][2] : 0.f) + (e > 2 ? ev[i][3] : 0.f) > target)
              pick = 4 * lane + e;
          }
          if (pick >= a.C) pick = a.C - 1;
          code[i] = __shfl(pick, src);
        }
      }
      if (lane < NI) {     // lane i publishes utterance i
        int cd = code[0];
#pragma unroll
        for (int i = 1; i < NI; ++i) cd = (lane == i) ? code[i] : cd;
        const int ul = NI * wave + lane, u = u0 + ul;
        const float smp = c_dec[cd];
        if (u < a.B) {
          a.audio_out[(size_t)u * a.Tout + t] = smp;
          a.codes_out[(size_t)u * a.Tout + t] = cd;
        }
        prev[32 + ul] = prev[ul];
        prev[ul] = smp;
      }
      if (a.logits_out) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int u = u0 + NI * wave + i;
          if (u < a.B && 4 * lane < a.C) {
            float* lo = a.logits_out + ((size_t)u * a.Tout + t) * a.C + 4 * lane;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (4 * lane + e < a.C) lo[e] = v[i][e];
          }
        }
      }
    }
    wg_barrier();
  }
}

}  // namespace

// elements of the three weight images srwn_generate16 takes (in the activation type), for nlayers layers of R residual /
// S skip channels: which = 0 the layers' image, 1 the first head 1x1's, 2 the second's
extern "C" int64_t srwn_generate16_image_elems(int32_t nlayers, int32_t which, int32_t R, int32_t S) {
  if ((R != 64 && R != 32) || (S != 256 && S != 128)) return 0;
  const int KR = R / 32, SRB = S / 64, HKS = S / 32;
  if (which == 0) return (int64_t)nlayers * 4 * (2 * KR + KR + SRB * KR) * FR;
  if (which == 1) return (int64_t)4 * SRB * HKS * FR;
  return (int64_t)4 * 4 * HKS * FR;
}

template <int R, int S>
static int generate16_launch(Gen16Args& a, const int32_t* dilations, int32_t nlayers, int32_t B, bool cond, int32_t M, void* stream) {
  long long off = 0;
  for (int l = 0; l < kG16MaxLayers; ++l) {
    a.dil[l] = (l < nlayers) ? dilations[l] : 1;
    a.ring_off[l] = off;
    if (l < nlayers) {
      if (dilations[l] < 1) return set_error(SRWN_E_SHAPE, "generate16: dilation %d", dilations[l]);
      off += (long long)(dilations[l] + 1) * 32 * R;
    }
  }
  a.ring_group_elems = off;
  // half-size workgroups while they still find a CU each (and always for a single stream's latency)
  bool half = (B + 15) / 16 <= num_cus();
  if (const char* e = getenv("SRWN_GEN16_NCB")) half = atoi(e) == 1;   // (tests: both bodies at any batch)
  const unsigned groups = half ? (unsigned)((B + 15) / 16) : (unsigned)((B + 31) / 32);
  using W = G16W<R, S>;
  const size_t sh = (size_t)(2 * 32 * W::LSX + 32 * W::LSH) * sizeof(T) +
                    (size_t)(32 * LGS + 64 + 2 * nlayers * R + 2 * S + 256 + 3 * R + 256) * 4;
  auto kfn = M > 0 ? (cond ? (half ? generate16_kernel<1, true, true, R, S> : generate16_kernel<2, true, true, R, S>)
                           : (half ? generate16_kernel<1, false, true, R, S> : generate16_kernel<2, false, true, R, S>))
                   : (half ? generate16_kernel<1, false, false, R, S> : generate16_kernel<2, false, false, R, S>);
  hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  if (e != hipSuccess) return set_error((int)e, "generate16: LDS %zu: %s", sh, hipGetErrorString(e));
  hipLaunchKernelGGL(kfn, dim3(groups), dim3(256), sh, (hipStream_t)stream, a);
  return check_launch("generate16");
}

static int generate16_impl(const void* wl, const void* wh1, const void* wh2, const float* bias_f, const float* bias_r,
                           const float* bs_sum, const float* b1, const float* b2, const float* init_w,
                           const float* init_b, void* ring, float* audio_out, int32_t* codes_out, float* logits_out,
                           const float* forced, const int32_t* dilations, int32_t nlayers, int32_t B, int32_t Tout,
                           int32_t nsteps, int32_t R, int32_t S, int32_t C, int32_t mode, uint64_t seed, void* stream,
                           int32_t M, const void* cond, int32_t cond_frames, int32_t pool, int64_t cond_ld) {
  if (B == 0 || nsteps == 0) return 0;
  if (!wl || !wh1 || !wh2 || !bias_f || !bias_r || !bs_sum || !b1 || !b2 || !init_w || !init_b || !ring || !audio_out ||
      !codes_out || !dilations)
    return set_error(SRWN_E_NULL, "generate16: null pointer");
  if ((R != 64 && R != 32) || (S != 256 && S != 128) || C < 2 || C > 256)
    return set_error(SRWN_E_UNSUPPORTED, "generate16: built for R = 64 or 32, S = 256 or 128, C <= 256 (got R=%d S=%d C=%d)", R, S, C);
  if (B < 0 || nsteps < 0 || nsteps > Tout || nlayers < 1 || nlayers > kG16MaxLayers || (mode != 0 && mode != 1))
    return set_error(SRWN_E_SHAPE, "generate16: B=%d nsteps=%d Tout=%d layers=%d mode=%d", B, nsteps, Tout, nlayers, mode);
  if (M == 0 && cond) return set_error(SRWN_E_UNSUPPORTED, "generate16: the conditioned softmax teacher is not built");
  if (cond && (cond_frames < 1 || pool < 1 || cond_ld < (int64_t)nlayers * R || (cond_ld % 4)))
    return set_error(SRWN_E_SHAPE, "generate16_mol: cond_frames=%d pool_stride=%d cond_ld=%lld", cond_frames, pool, (long long)cond_ld);
  Gen16Args a;
  a.wl = wl; a.wh1 = wh1; a.wh2 = wh2; a.bias_f = bias_f; a.bias_r = bias_r; a.bs_sum = bs_sum; a.b1 = b1; a.b2 = b2;
  a.init_w = init_w; a.init_b = init_b; a.ring = ring; a.audio_out = audio_out; a.codes_out = codes_out;
  a.logits_out = logits_out; a.forced = forced;
  a.B = B; a.Tout = Tout; a.nsteps = nsteps; a.L = nlayers; a.C = C; a.Cp = (C + 31) / 32 * 32; a.mode = mode; a.Q = C; a.seed = seed;
  a.M = M; a.cond = cond; a.cond_frames = cond_frames; a.pool = pool; a.cond_ld = cond_ld;
  if (R == 64 && S == 256) return generate16_launch<64, 256>(a, dilations, nlayers, B, cond != nullptr, M, stream);
  if (R == 64) return generate16_launch<64, 128>(a, dilations, nlayers, B, cond != nullptr, M, stream);
  if (S == 256) return generate16_launch<32, 256>(a, dilations, nlayers, B, cond != nullptr, M, stream);
  return generate16_launch<32, 128>(a, dilations, nlayers, B, cond != nullptr, M, stream);
}

extern "C" int srwn_generate16(const void* wl, const void* wh1, const void* wh2, const float* bias_f,
                               const float* bias_r, const float* bs_sum, const float* b1, const float* b2,
                               const float* init_w, const float* init_b, void* ring, float* audio_out,
                               int32_t* codes_out, float* logits_out, const float* forced, const int32_t* dilations,
                               int32_t nlayers, int32_t B, int32_t Tout, int32_t nsteps, int32_t R, int32_t S,
                               int32_t C, int32_t mode, uint64_t seed, void* stream) {
  return generate16_impl(wl, wh1, wh2, bias_f, bias_r, bs_sum, b1, b2, init_w, init_b, ring, audio_out, codes_out,
                         logits_out, forced, dilations, nlayers, B, Tout, nsteps, R, S, C, mode, seed, stream, 0, nullptr, 1,
                         1, 0);
}

// the same body for the conditioned mixture-of-logistics decoder (srwn_generate_mol's arguments; wh2 / b2 cover
// ceil(4M/32)*32 rows)
extern "C" int srwn_generate16_mol(const void* wl, const void* wh1, const void* wh2, const float* bias_f,
                                   const float* bias_r, const float* bs_sum, const float* b1, const float* b2,
                                   const float* init_w, const float* init_b, void* ring, float* audio_out,
                                   int32_t* codes_out, float* logits_out, const float* forced,
                                   const int32_t* dilations, int32_t nlayers, int32_t B, int32_t Tout, int32_t nsteps,
                                   int32_t R, int32_t S, int32_t num_mixtures, const void* cond, int32_t cond_frames,
                                   int32_t pool_stride, int64_t cond_ld, int32_t mode, uint64_t seed, void* stream) {
  if (num_mixtures < 1 || num_mixtures > 16)
    return set_error(SRWN_E_SHAPE, "generate16_mol: num_mixtures=%d (1..16)", num_mixtures);
  return generate16_impl(wl, wh1, wh2, bias_f, bias_r, bs_sum, b1, b2, init_w, init_b, ring, audio_out, codes_out,
                         logits_out, forced, dilations, nlayers, B, Tout, nsteps, R, S, 4 * num_mixtures, mode, seed, stream,
                         num_mixtures, cond, cond ? cond_frames : 1, cond ? pool_stride : 1, cond_ld);
}
