// Fused ResidualDilationLayerNC kernels of the auto-encoder's encoder (ops.py:48-58; model.py:141-150), bf16,
// 128 channels, K = 2 taps at t and t+1 (SAME padding).  gfx950 (MI355X) only.
//
// The generic path runs every layer as two row-streaming GEMM launches (srwn_tap_linear): 33 + 21 us forward and
// the same again backward at batch 8 x 16000, each re-streaming its 64 / 32 KB weight image per 128-row workgroup and
// round-tripping the intermediate through HBM.  Here one persistent workgroup per CU keeps BOTH weight images of the
// layer in LDS (96 KB), every wave walks 32-row tiles, and the first product's accumulator tile is the B operand of
// the second (the residual-layer kernels' register chaining, srwn_fwd.hip / srwn_bwd.hip):
//   forward   a = relu(b + sum_k W[k] . r[t+k]) -> a_out ;  r' = relu(br + Wr . a) -> r_out
//   backward  dh = [r > 0] . sum_k W[k]^T . dpre_up[t-k] -> dh_out ;
//             dpre = [a > 0] . (Wr_below^T . dh + frame_add) -> dpre_out
// (the backward kernel pairs the conv data gradient of layer l with the 1x1 data gradient of the layer BELOW it, so
// no tile needs a neighbour's result).  Operand tiles are fetched as whole rows one tile ahead in registers and
// redistributed through a wave-private padded LDS row buffer, which is also the stage of the whole-row stores.
// The relu masks travel as BITS: the forward kernel writes, per 32-row tile and lane, one 64-bit word whose bit
// (mt, q) says "accumulator register q of row tile mt is positive" -- the backward kernel, which holds its
// gradients in the same layout, reads 8 bytes per lane instead of staging two more 128-channel tensors through LDS
// (65 MB of its 164 MB per launch).  srwn_nc_mask_bits makes the words for a tensor some other kernel wrote.
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

namespace {

constexpr int kC = 128, kRT = 4, kKS = 8, kTaps = 2;
constexpr int kLS = RowStage<bf16_t>::stride(kC);   // 136 elements: padded LDS row
constexpr int kBufRows = 34;                        // 33-row tap window (+1 so that every 4-row piece fits)
constexpr int kNI = 9;                              // 16 lanes per 256-byte row, 4 rows per instruction, 36 rows
constexpr size_t kLdsBytes = (size_t)(kRT * kTaps * kKS + kRT * kKS) * 64 * sizeof(Frag<bf16_t>) + 256 * sizeof(float) +
                             (size_t)4 * kBufRows * kLS * sizeof(bf16_t);

// Mask word of one lane's 64 relu'd values v >= 0: two 32-bit halves (row tiles 0,1 | 2,3), value (mt, q) at bit
// 31 - (16*(mt&1) + q) of its half.  Two VALU ops per value: n = 0 - v carries the sign bit exactly when v > 0
// (0 - (+-0) = +0), and v_alignbit shifts it into the word.
__device__ __forceinline__ unsigned long long positive_bits(const float (&v)[kRT][16]) {
  unsigned w[2] = {0u, 0u};
#pragma unroll
  for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) w[mt >> 1] = __builtin_amdgcn_alignbit(w[mt >> 1], __float_as_uint(0.0f - v[mt][q]), 31);
  return (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
}
__device__ __forceinline__ bool mask_bit(unsigned long long w, int mt, int q) {
  const unsigned h = (mt >> 1) ? (unsigned)(w >> 32) : (unsigned)w;
  return (h & (1u << (31 - (16 * (mt & 1) + q)))) != 0u;
}

struct NcFwdArgs {
  const bf16_t* r_in; const bf16_t* wconv; const bf16_t* wres; const float* bias_c; const float* bias_r;
  bf16_t* a_out; bf16_t* r_out; unsigned long long* a_bits; unsigned long long* r_bits; int Tlen, ntb, ntiles;
};

template <bool RES>   // RES: the 1x1 residual product and its output exist (not the last layer)
__global__ __launch_bounds__(256) void nc_layer_fwd_kernel(NcFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<bf16_t>* lds_conv = reinterpret_cast<Frag<bf16_t>*>(smem);          // [RT][taps*KS][64]
  Frag<bf16_t>* lds_res = lds_conv + kRT * kTaps * kKS * 64;                 // [RT][KS][64], permuted k
  float* lds_bias = reinterpret_cast<float*>(lds_res + kRT * kKS * 64);      // [conv 128 | res 128]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16_t* buf = reinterpret_cast<bf16_t*>(lds_bias + 256) + wave * (kBufRows * kLS);
  lds_dma_copy(a.wconv, lds_conv, kRT * kTaps * kKS * 64 * (int)sizeof(Frag<bf16_t>), wave, lane, 4);
  if (RES) lds_dma_copy(a.wres, lds_res, kRT * kKS * 64 * (int)sizeof(Frag<bf16_t>), wave, lane, 4);
  lds_bias[threadIdx.x] = threadIdx.x < kC ? a.bias_c[threadIdx.x] : (RES ? a.bias_r[threadIdx.x - kC] : 0.0f);

  const int col = lane & 31, half = lane >> 5;
  const int rl = lane >> 4, piece = lane & 15;
  f32x4 xr[kNI];   // rows t0 .. t0+35 of the tile (clamped into the clip), whole rows
  auto load_regs = [&](int tile) {
    const int b = tile / a.ntb;
    const int t0 = (tile - b * a.ntb) * 32;
    const bf16_t* xb = a.r_in + (size_t)b * a.Tlen * kC + piece * 8;
#pragma unroll
    for (int i = 0; i < kNI; ++i) {
      int t = t0 + 4 * i + rl;
      t = t < a.Tlen ? t : a.Tlen - 1;
      xr[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)t * kC);
    }
  };
  auto put_regs = [&]() {
    wave_lds_order();                     // the previous tile's reads of the buffer are done
#pragma unroll
    for (int i = 0; i < kNI; ++i) {
      const int rr = 4 * i + rl;
      if (rr < kBufRows) *reinterpret_cast<f32x4*>(buf + rr * kLS + piece * 8) = xr[i];
    }
    wave_lds_order();
  };

  auto process = [&](int tile) {
    const int b = tile / a.ntb;
    const int t0 = (tile - b * a.ntb) * 32;
    const int tc = t0 + col;
    const int rows_valid = a.Tlen - t0;
    const size_t out0 = ((size_t)b * a.Tlen + t0) * kC;
    f32x16 accF[kRT];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lds_bias + 32 * mt + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) accF[mt][4 * g + e] = bv[e];
      }
    // operands of k-step s+1 are read from LDS before the MFMAs of k-step s issue (one wave per SIMD here: nothing
    // else hides the ds_read -> v_mfma round trip hipcc otherwise leaves in front of every MFMA)
    {
      constexpr int NS = kTaps * kKS;
      Frag<bf16_t> af[2][kRT], bfr[2];
      auto fetch = [&](int s, int slot) {
        const int k = s / kKS, ks = s % kKS;
        bfr[slot] = load_nat(buf + (col + k) * kLS + 16 * ks + 8 * half);
#pragma unroll
        for (int mt = 0; mt < kRT; ++mt) af[slot][mt] = lds_conv[(mt * NS + s) * 64 + lane];
      };
      fetch(0, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) fetch(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool valid = tc + s / kKS < a.Tlen;   // a tap beyond the clip contributes 0 (SAME padding)
        const Frag<bf16_t> bf = valid ? bfr[s & 1] : zero_frag<bf16_t>();
#pragma unroll
        for (int mt = 0; mt < kRT; ++mt) mma(accF[mt], af[s & 1][mt], bf);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    Frag<bf16_t> cf[kKS];
    {
      float av[kRT][16];
#pragma unroll
      for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float v = fmaxf(accF[mt][q], 0.0f);
          av[mt][q] = v;
          cf[2 * mt + (q >> 3)].set(q & 7, v);
        }
      if (a.a_bits) a.a_bits[(size_t)tile * 64 + lane] = positive_bits(av);
      store_rows_via_lds<bf16_t, kRT, true>(buf, a.a_out + out0, kC, av, rows_valid, lane);
    }
    if (!RES) return;
    f32x16 accR[kRT];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lds_bias + kC + 32 * mt + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) accR[mt][4 * g + e] = bv[e];
      }
    {
      Frag<bf16_t> af[2][kRT];
#pragma unroll
      for (int mt = 0; mt < kRT; ++mt) af[0][mt] = lds_res[(mt * kKS) * 64 + lane];
#pragma unroll
      for (int s = 0; s < kKS; ++s) {
        if (s + 1 < kKS) {
#pragma unroll
          for (int mt = 0; mt < kRT; ++mt) af[(s + 1) & 1][mt] = lds_res[(mt * kKS + s + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < kRT; ++mt) mma(accR[mt], af[s & 1][mt], cf[s]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float rv[kRT][16];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) rv[mt][q] = fmaxf(accR[mt][q], 0.0f);
    if (a.r_bits) a.r_bits[(size_t)tile * 64 + lane] = positive_bits(rv);
    store_rows_via_lds<bf16_t, kRT, true>(buf, a.r_out + out0, kC, rv, rows_valid, lane);
  };

  const int stride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  load_regs(tile < a.ntiles ? tile : a.ntiles - 1);
  __syncthreads();   // weights and biases landed (vmcnt(0) + barrier)
  while (tile + stride < a.ntiles) {   // (last tile peeled: the prefetch inside the loop is unconditional)
    put_regs();
    load_regs(tile + stride);
    process(tile);
    tile += stride;
  }
  if (tile < a.ntiles) {
    put_regs();
    process(tile);
  }
}

struct NcBwdArgs {
  const bf16_t* dpre_up; const bf16_t* wconvT; const unsigned long long* r_bits; bf16_t* dh_out;
  const bf16_t* wresT; const float* fadd; int64_t fadd_ld; int frames, pool; float fadd_scale;
  const unsigned long long* a_bits; bf16_t* dpre_out; int Tlen, ntb, ntiles;
};

template <bool DOWN, bool FADD>   // DOWN: the 1x1 data gradient of the layer below follows the conv's; FADD: + frame_add
__global__ __launch_bounds__(256) void nc_layer_bwd_kernel(NcBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<bf16_t>* lds_conv = reinterpret_cast<Frag<bf16_t>*>(smem);          // [RT][taps*KS][64]: rows = in channel
  Frag<bf16_t>* lds_res = lds_conv + kRT * kTaps * kKS * 64;                 // [RT][KS][64], permuted k
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16_t* buf = reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(lds_res + kRT * kKS * 64) + 256) + wave * (kBufRows * kLS);
  constexpr bool down = DOWN;
  lds_dma_copy(a.wconvT, lds_conv, kRT * kTaps * kKS * 64 * (int)sizeof(Frag<bf16_t>), wave, lane, 4);
  if (down) lds_dma_copy(a.wresT, lds_res, kRT * kKS * 64 * (int)sizeof(Frag<bf16_t>), wave, lane, 4);

  const int col = lane & 31, half = lane >> 5;
  const int rl = lane >> 4, piece = lane & 15;
  struct Regs {
    f32x4 up[kNI];                    // dpre_up rows t0-1 .. t0+34 (clamped)
    unsigned long long rbits, abits;  // relu masks of this lane's 64 accumulator positions
    f32x4 fa[kRT][4];                 // frame_add of this lane's time row, accumulator layout (fetched with the tile:
                                      // a load issued mid-tile would make its s_waitcnt drain the prefetch as well)
  };
  auto load_regs = [&](int tile, Regs& r) {
    const int b = tile / a.ntb;
    const int t0 = (tile - b * a.ntb) * 32;
    const size_t boff = (size_t)b * a.Tlen * kC + piece * 8;
#pragma unroll
    for (int i = 0; i < kNI; ++i) {
      int t = t0 - 1 + 4 * i + rl;
      t = t < 0 ? 0 : (t < a.Tlen ? t : a.Tlen - 1);
      r.up[i] = *reinterpret_cast<const f32x4*>(a.dpre_up + boff + (size_t)t * kC);
    }
    r.rbits = a.r_bits[(size_t)tile * 64 + lane];
    r.abits = down ? a.a_bits[(size_t)tile * 64 + lane] : 0ull;
    if (down && FADD) {
      int tc = t0 + col;
      tc = tc < a.Tlen ? tc : a.Tlen - 1;
      const float* frow = a.fadd + ((size_t)b * a.frames + tc / a.pool) * a.fadd_ld + 4 * half;
#pragma unroll
      for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) r.fa[mt][g] = *reinterpret_cast<const f32x4*>(frow + 32 * mt + 8 * g);
    }
  };

  auto process = [&](int tile, const Regs& r) {
    const int b = tile / a.ntb;
    const int t0 = (tile - b * a.ntb) * 32;
    const int tc = t0 + col;
    const bool ok = tc < a.Tlen;
    const int rows_valid = a.Tlen - t0;
    const size_t out0 = ((size_t)b * a.Tlen + t0) * kC;
    // tap window: buffer row rr holds time t0 - 1 + rr
    wave_lds_order();
#pragma unroll
    for (int i = 0; i < kNI; ++i) {
      const int rr = 4 * i + rl;
      if (rr < kBufRows) *reinterpret_cast<f32x4*>(buf + rr * kLS + piece * 8) = r.up[i];
    }
    wave_lds_order();
    f32x16 accG[kRT];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) accG[mt][q] = 0.0f;
    {   // (operands one k-step ahead of their MFMAs, as in the forward kernel)
      constexpr int NS = kTaps * kKS;
      Frag<bf16_t> af[2][kRT], bfr[2];
      auto fetch = [&](int s, int slot) {
        const int k = s / kKS, ks = s % kKS;
        bfr[slot] = load_nat(buf + (col + 1 - k) * kLS + 16 * ks + 8 * half);
#pragma unroll
        for (int mt = 0; mt < kRT; ++mt) af[slot][mt] = lds_conv[(mt * NS + s) * 64 + lane];
      };
      fetch(0, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) fetch(s + 1, (s + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool valid = ok && (tc - s / kKS >= 0);   // dh[t] = sum_k W[k]^T dpre[t-k]; before the clip: 0
        const Frag<bf16_t> bf = valid ? bfr[s & 1] : zero_frag<bf16_t>();
#pragma unroll
        for (int mt = 0; mt < kRT; ++mt) mma(accG[mt], af[s & 1][mt], bf);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    Frag<bf16_t> gf[kKS];
    {
      float dv[kRT][16];
#pragma unroll
      for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float v = (ok && mask_bit(r.rbits, mt, q)) ? accG[mt][q] : 0.0f;
          dv[mt][q] = v;
          gf[2 * mt + (q >> 3)].set(q & 7, v);
        }
      store_rows_via_lds<bf16_t, kRT, true>(buf, a.dh_out + out0, kC, dv, rows_valid, lane);
    }
    if (!down) return;
    f32x16 accC[kRT];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) accC[mt][4 * g + e] = FADD ? r.fa[mt][g][e] * a.fadd_scale : 0.0f;
    {
      Frag<bf16_t> af[2][kRT];
#pragma unroll
      for (int mt = 0; mt < kRT; ++mt) af[0][mt] = lds_res[(mt * kKS) * 64 + lane];
#pragma unroll
      for (int s = 0; s < kKS; ++s) {
        if (s + 1 < kKS) {
#pragma unroll
          for (int mt = 0; mt < kRT; ++mt) af[(s + 1) & 1][mt] = lds_res[(mt * kKS + s + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < kRT; ++mt) mma(accC[mt], af[s & 1][mt], gf[s]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float pv[kRT][16];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) pv[mt][q] = (ok && mask_bit(r.abits, mt, q)) ? accC[mt][q] : 0.0f;
    store_rows_via_lds<bf16_t, kRT, true>(buf, a.dpre_out + out0, kC, pv, rows_valid, lane);
  };

  // two register tiles: the next tile's rows are in flight while this one is processed
  const int stride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  Regs ra, rb;
  load_regs(tile < a.ntiles ? tile : a.ntiles - 1, ra);
  __syncthreads();   // weights landed (vmcnt(0) + barrier)
  while (tile < a.ntiles) {
    if (tile + stride < a.ntiles) load_regs(tile + stride, rb);
    process(tile, ra);
    tile += stride;
    if (tile >= a.ntiles) break;
    if (tile + stride < a.ntiles) load_regs(tile + stride, ra);
    process(tile, rb);
    tile += stride;
  }
}

// mask words of a [B,T,128] tensor in the kernels' tile/lane layout (one wave per 32-row tile)
__global__ __launch_bounds__(256) void nc_mask_bits_kernel(const bf16_t* __restrict__ x, unsigned long long* __restrict__ bits,
                                                           int Tlen, int ntb, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;
  const int b = tile / ntb;
  int t = (tile - b * ntb) * 32 + col;
  t = t < Tlen ? t : Tlen - 1;
  const bf16_t* row = x + ((size_t)b * Tlen + t) * kC + 4 * half;
  float v[kRT][16];
#pragma unroll
  for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 x4 = load4(row + 32 * mt + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[mt][4 * g + e] = fmaxf(x4[e], 0.0f);
    }
  bits[(size_t)tile * 64 + lane] = positive_bits(v);
}

template <typename KFN, typename ARGS>
int launch_nc(KFN kfn, ARGS& a, int B, int T, const char* what, hipStream_t st) {
  a.Tlen = T;
  a.ntb = (T + 31) / 32;
  const long long ntiles = (long long)B * a.ntb;
  if (ntiles > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "%s: B*T too large", what);
  a.ntiles = (int)ntiles;
  long long blocks = (ntiles + 3) / 4;
  if (blocks > 256) blocks = 256;   // 134 KB of LDS: one workgroup per CU
  hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
  if (e != hipSuccess) return set_error((int)e, "%s: LDS %zu: %s", what, kLdsBytes, hipGetErrorString(e));
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(256), kLdsBytes, st, a);
  return check_launch(what);
}

}  // namespace

extern "C" int64_t srwn_nc_mask_words(int32_t B, int32_t T) { return (int64_t)B * ((T + 31) / 32) * 64; }

extern "C" int srwn_nc_mask_bits(const void* x, uint64_t* bits, int32_t B, int32_t T, int32_t C, int32_t dtype,
                                 void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!x || !bits) return set_error(SRWN_E_NULL, "nc_mask_bits: null pointer");
  if (B < 0 || T < 0) return set_error(SRWN_E_SHAPE, "nc_mask_bits: B=%d T=%d", B, T);
  if (C != kC || dtype != SRWN_BF16) return set_error(SRWN_E_UNSUPPORTED, "nc_mask_bits: built for 128 channels, bf16");
  const int ntb = (T + 31) / 32;
  const long long ntiles = (long long)B * ntb;
  if (ntiles > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "nc_mask_bits: B*T too large");
  hipLaunchKernelGGL(nc_mask_bits_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (unsigned long long*)bits, T, ntb, (int)ntiles);
  return check_launch("nc_mask_bits");
}

extern "C" int srwn_nc_layer_fwd(const void* r_in, const void* wconv, const void* wres, const float* bias_c,
                                 const float* bias_r, void* a_out, void* r_out, uint64_t* a_bits, uint64_t* r_bits,
                                 int32_t B, int32_t T, int32_t C, int32_t K, int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!r_in || !wconv || !bias_c || !a_out) return set_error(SRWN_E_NULL, "nc_layer_fwd: null pointer");
  if (r_out && (!wres || !bias_r)) return set_error(SRWN_E_NULL, "nc_layer_fwd: r_out needs wres and bias_r");
  if (B < 0 || T < 0) return set_error(SRWN_E_SHAPE, "nc_layer_fwd: B=%d T=%d", B, T);
  if (C != kC || K != kTaps || dtype != SRWN_BF16)
    return set_error(SRWN_E_UNSUPPORTED, "nc_layer_fwd: built for 128 channels, K=2, bf16 (got C=%d K=%d dtype=%d)", C, K, dtype);
  NcFwdArgs a;
  a.r_in = (const bf16_t*)r_in; a.wconv = (const bf16_t*)wconv; a.wres = (const bf16_t*)wres;
  a.bias_c = bias_c; a.bias_r = bias_r; a.a_out = (bf16_t*)a_out; a.r_out = (bf16_t*)r_out;
  a.a_bits = (unsigned long long*)a_bits; a.r_bits = r_out ? (unsigned long long*)r_bits : nullptr;
  return r_out ? launch_nc(nc_layer_fwd_kernel<true>, a, B, T, "nc_layer_fwd", (hipStream_t)stream)
               : launch_nc(nc_layer_fwd_kernel<false>, a, B, T, "nc_layer_fwd", (hipStream_t)stream);
}

extern "C" int srwn_nc_layer_bwd(const void* dpre_up, const void* wconvT, const uint64_t* r_bits, void* dh_out,
                                 const void* wresT, const float* frame_add, int64_t frame_add_ld, int32_t frames,
                                 int32_t pool_stride, float frame_add_scale, const uint64_t* a_bits, void* dpre_out,
                                 int32_t B, int32_t T, int32_t C, int32_t K, int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!dpre_up || !wconvT || !r_bits || !dh_out) return set_error(SRWN_E_NULL, "nc_layer_bwd: null pointer");
  if (dpre_out && (!wresT || !a_bits)) return set_error(SRWN_E_NULL, "nc_layer_bwd: dpre_out needs wresT and a_bits");
  if (B < 0 || T < 0) return set_error(SRWN_E_SHAPE, "nc_layer_bwd: B=%d T=%d", B, T);
  if (C != kC || K != kTaps || dtype != SRWN_BF16)
    return set_error(SRWN_E_UNSUPPORTED, "nc_layer_bwd: built for 128 channels, K=2, bf16 (got C=%d K=%d dtype=%d)", C, K, dtype);
  if (frame_add && (pool_stride < 1 || frame_add_ld < kC || (int64_t)frames * pool_stride < T))
    return set_error(SRWN_E_SHAPE, "nc_layer_bwd: frames %d x pool %d < T %d", frames, pool_stride, T);
  NcBwdArgs a;
  a.dpre_up = (const bf16_t*)dpre_up; a.wconvT = (const bf16_t*)wconvT; a.r_bits = (const unsigned long long*)r_bits;
  a.dh_out = (bf16_t*)dh_out; a.wresT = (const bf16_t*)wresT; a.fadd = frame_add; a.fadd_ld = frame_add_ld;
  a.frames = frames; a.pool = pool_stride > 0 ? pool_stride : 1; a.fadd_scale = frame_add_scale;
  a.a_bits = (const unsigned long long*)a_bits; a.dpre_out = (bf16_t*)dpre_out;
  if (!dpre_out) return launch_nc(nc_layer_bwd_kernel<false, false>, a, B, T, "nc_layer_bwd", (hipStream_t)stream);
  return frame_add ? launch_nc(nc_layer_bwd_kernel<true, true>, a, B, T, "nc_layer_bwd", (hipStream_t)stream)
                   : launch_nc(nc_layer_bwd_kernel<true, false>, a, B, T, "nc_layer_bwd", (hipStream_t)stream);
}
