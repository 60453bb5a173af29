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

struct NcFwdArgs {
  const bf16_t* r_in; const bf16_t* wconv; const bf16_t* wres; const float* bias_c; const float* bias_r;
  bf16_t* a_out; bf16_t* r_out; int Tlen, ntb, ntiles;
};

__global__ __launch_bounds__(256) void nc_layer_fwd_kernel(NcFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<bf16_t>* lds_conv = reinterpret_cast<Frag<bf16_t>*>(smem);          // [RT][taps*KS][64]
  Frag<bf16_t>* lds_res = lds_conv + kRT * kTaps * kKS * 64;                 // [RT][KS][64], permuted k
  float* lds_bias = reinterpret_cast<float*>(lds_res + kRT * kKS * 64);      // [conv 128 | res 128]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16_t* buf = reinterpret_cast<bf16_t*>(lds_bias + 256) + wave * (kBufRows * kLS);
  lds_dma_copy(a.wconv, lds_conv, kRT * kTaps * kKS * 64 * (int)sizeof(Frag<bf16_t>), wave, lane, 4);
  if (a.r_out) lds_dma_copy(a.wres, lds_res, kRT * kKS * 64 * (int)sizeof(Frag<bf16_t>), wave, lane, 4);
  lds_bias[threadIdx.x] = threadIdx.x < kC ? a.bias_c[threadIdx.x] : (a.r_out ? a.bias_r[threadIdx.x - kC] : 0.0f);

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
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      const bool valid = tc + k < a.Tlen;   // a tap beyond the clip contributes 0 (SAME padding)
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) {
        Frag<bf16_t> bf = load_nat(buf + (col + k) * kLS + 16 * ks + 8 * half);
        bf = valid ? bf : zero_frag<bf16_t>();
#pragma unroll
        for (int mt = 0; mt < kRT; ++mt) mma(accF[mt], lds_conv[(mt * (kTaps * kKS) + k * kKS + ks) * 64 + lane], bf);
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
      store_rows_via_lds<bf16_t, kRT>(buf, a.a_out + out0, kC, av, rows_valid, lane);
    }
    if (!a.r_out) return;
    f32x16 accR[kRT];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(lds_bias + kC + 32 * mt + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) accR[mt][4 * g + e] = bv[e];
      }
#pragma unroll
    for (int s = 0; s < kKS; ++s)
#pragma unroll
      for (int mt = 0; mt < kRT; ++mt) mma(accR[mt], lds_res[(mt * kKS + s) * 64 + lane], cf[s]);
    float rv[kRT][16];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) rv[mt][q] = fmaxf(accR[mt][q], 0.0f);
    store_rows_via_lds<bf16_t, kRT>(buf, a.r_out + out0, kC, rv, rows_valid, lane);
  };

  const int stride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  load_regs(tile < a.ntiles ? tile : a.ntiles - 1);
  __syncthreads();   // weights and biases landed (vmcnt(0) + barrier)
  while (tile < a.ntiles) {
    put_regs();
    if (tile + stride < a.ntiles) load_regs(tile + stride);
    process(tile);
    tile += stride;
  }
}

struct NcBwdArgs {
  const bf16_t* dpre_up; const bf16_t* wconvT; const bf16_t* r_mask; bf16_t* dh_out;
  const bf16_t* wresT; const float* fadd; int64_t fadd_ld; int frames, pool; float fadd_scale;
  const bf16_t* a_mask; bf16_t* dpre_out; int Tlen, ntb, ntiles;
};

__global__ __launch_bounds__(256) void nc_layer_bwd_kernel(NcBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<bf16_t>* lds_conv = reinterpret_cast<Frag<bf16_t>*>(smem);          // [RT][taps*KS][64]: rows = in channel
  Frag<bf16_t>* lds_res = lds_conv + kRT * kTaps * kKS * 64;                 // [RT][KS][64], permuted k
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16_t* buf = reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(lds_res + kRT * kKS * 64) + 256) + wave * (kBufRows * kLS);
  const bool down = a.dpre_out != nullptr;
  lds_dma_copy(a.wconvT, lds_conv, kRT * kTaps * kKS * 64 * (int)sizeof(Frag<bf16_t>), wave, lane, 4);
  if (down) lds_dma_copy(a.wresT, lds_res, kRT * kKS * 64 * (int)sizeof(Frag<bf16_t>), wave, lane, 4);

  const int col = lane & 31, half = lane >> 5;
  const int rl = lane >> 4, piece = lane & 15;
  typedef bf16x4 raw4;
  struct Regs {
    f32x4 up[kNI];      // dpre_up rows t0-1 .. t0+34 (clamped)
    f32x4 rm[8], am[8]; // mask tiles r, a: rows t0 .. t0+31
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
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int t = t0 + 4 * i + rl;
      t = t < a.Tlen ? t : a.Tlen - 1;
      r.rm[i] = *reinterpret_cast<const f32x4*>(a.r_mask + boff + (size_t)t * kC);
      if (down) r.am[i] = *reinterpret_cast<const f32x4*>(a.a_mask + boff + (size_t)t * kC);
    }
  };
  auto put_tile = [&](const f32x4 (&v)[8]) {   // a 32-row tile into rows 0..31 of the buffer
    wave_lds_order();
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(buf + (4 * i + rl) * kLS + piece * 8) = v[i];
    wave_lds_order();
  };
  auto get_acc = [&](raw4 (&o)[kRT][4]) {
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) o[mt][g] = *reinterpret_cast<const raw4*>(buf + col * kLS + 32 * mt + 8 * g + 4 * half);
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
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      const bool valid = ok && (tc - k >= 0);   // dh[t] = sum_k W[k]^T dpre[t-k]; before the clip: 0
#pragma unroll
      for (int ks = 0; ks < kKS; ++ks) {
        Frag<bf16_t> bf = load_nat(buf + (col + 1 - k) * kLS + 16 * ks + 8 * half);
        bf = valid ? bf : zero_frag<bf16_t>();
#pragma unroll
        for (int mt = 0; mt < kRT; ++mt) mma(accG[mt], lds_conv[(mt * (kTaps * kKS) + k * kKS + ks) * 64 + lane], bf);
      }
    }
    raw4 m[kRT][4];
    put_tile(r.rm);
    get_acc(m);
    Frag<bf16_t> gf[kKS];
    {
      float dv[kRT][16];
#pragma unroll
      for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float v = (ok && (float)m[mt][q >> 2][q & 3] > 0.0f) ? accG[mt][q] : 0.0f;
          dv[mt][q] = v;
          gf[2 * mt + (q >> 3)].set(q & 7, v);
        }
      store_rows_via_lds<bf16_t, kRT>(buf, a.dh_out + out0, kC, dv, rows_valid, lane);
    }
    if (!down) return;
    f32x16 accC[kRT];
    const float* frow = a.fadd ? a.fadd + ((size_t)b * a.frames + (ok ? tc : 0) / a.pool) * a.fadd_ld : nullptr;
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 fa = {0.f, 0.f, 0.f, 0.f};
        if (frow) fa = *reinterpret_cast<const f32x4*>(frow + 32 * mt + 8 * g + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) accC[mt][4 * g + e] = fa[e] * a.fadd_scale;
      }
#pragma unroll
    for (int s = 0; s < kKS; ++s)
#pragma unroll
      for (int mt = 0; mt < kRT; ++mt) mma(accC[mt], lds_res[(mt * kKS + s) * 64 + lane], gf[s]);
    put_tile(r.am);
    get_acc(m);
    float pv[kRT][16];
#pragma unroll
    for (int mt = 0; mt < kRT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) pv[mt][q] = (ok && (float)m[mt][q >> 2][q & 3] > 0.0f) ? accC[mt][q] : 0.0f;
    store_rows_via_lds<bf16_t, kRT>(buf, a.dpre_out + out0, kC, pv, rows_valid, lane);
  };

  // two register tiles (the a mask of a tile is consumed late in `process`): the next tile's rows are in flight
  // while this one is processed
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

extern "C" int srwn_nc_layer_fwd(const void* r_in, const void* wconv, const void* wres, const float* bias_c,
                                 const float* bias_r, void* a_out, void* r_out, int32_t B, int32_t T, int32_t C,
                                 int32_t K, int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!r_in || !wconv || !bias_c || !a_out) return set_error(SRWN_E_NULL, "nc_layer_fwd: null pointer");
  if (r_out && (!wres || !bias_r)) return set_error(SRWN_E_NULL, "nc_layer_fwd: r_out needs wres and bias_r");
  if (B < 0 || T < 0) return set_error(SRWN_E_SHAPE, "nc_layer_fwd: B=%d T=%d", B, T);
  if (C != kC || K != kTaps || dtype != SRWN_BF16)
    return set_error(SRWN_E_UNSUPPORTED, "nc_layer_fwd: built for 128 channels, K=2, bf16 (got C=%d K=%d dtype=%d)", C, K, dtype);
  NcFwdArgs a;
  a.r_in = (const bf16_t*)r_in; a.wconv = (const bf16_t*)wconv; a.wres = (const bf16_t*)wres;
  a.bias_c = bias_c; a.bias_r = bias_r; a.a_out = (bf16_t*)a_out; a.r_out = (bf16_t*)r_out;
  return launch_nc(nc_layer_fwd_kernel, a, B, T, "nc_layer_fwd", (hipStream_t)stream);
}

extern "C" int srwn_nc_layer_bwd(const void* dpre_up, const void* wconvT, const void* r_mask, void* dh_out,
                                 const void* wresT, const float* frame_add, int64_t frame_add_ld, int32_t frames,
                                 int32_t pool_stride, float frame_add_scale, const void* a_mask, void* dpre_out,
                                 int32_t B, int32_t T, int32_t C, int32_t K, int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!dpre_up || !wconvT || !r_mask || !dh_out) return set_error(SRWN_E_NULL, "nc_layer_bwd: null pointer");
  if (dpre_out && (!wresT || !a_mask)) return set_error(SRWN_E_NULL, "nc_layer_bwd: dpre_out needs wresT and a_mask");
  if (B < 0 || T < 0) return set_error(SRWN_E_SHAPE, "nc_layer_bwd: B=%d T=%d", B, T);
  if (C != kC || K != kTaps || dtype != SRWN_BF16)
    return set_error(SRWN_E_UNSUPPORTED, "nc_layer_bwd: built for 128 channels, K=2, bf16 (got C=%d K=%d dtype=%d)", C, K, dtype);
  if (frame_add && (pool_stride < 1 || frame_add_ld < kC || (int64_t)frames * pool_stride < T))
    return set_error(SRWN_E_SHAPE, "nc_layer_bwd: frames %d x pool %d < T %d", frames, pool_stride, T);
  NcBwdArgs a;
  a.dpre_up = (const bf16_t*)dpre_up; a.wconvT = (const bf16_t*)wconvT; a.r_mask = (const bf16_t*)r_mask;
  a.dh_out = (bf16_t*)dh_out; a.wresT = (const bf16_t*)wresT; a.fadd = frame_add; a.fadd_ld = frame_add_ld;
  a.frames = frames; a.pool = pool_stride > 0 ? pool_stride : 1; a.fadd_scale = frame_add_scale;
  a.a_mask = (const bf16_t*)a_mask; a.dpre_out = (bf16_t*)dpre_out;
  return launch_nc(nc_layer_bwd_kernel, a, B, T, "nc_layer_bwd", (hipStream_t)stream);
}
