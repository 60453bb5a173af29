// Forward kernels: fused residual layer, pointwise linear (channels GEMM), softmax-CE head.
// gfx950 (MI355X) only; see srwn_common.h for the MFMA orientation and lane maps.
#include <cstdlib>
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

// ------------------------------------------------------------------------------------------
// fused residual layer forward (ops.py:23-46 + model.py:180-183)
//   Persistent waves: grid = O(#CUs) workgroups of 4 waves; a wave walks 32-step time tiles
//   (tile id = b * tiles_per_clip + t/32) with stride = total waves.
//   Activations enter through LDS as WHOLE ROWS: each tap's 32-row window of x is one contiguous
//   4 KiB block of the channels-last tensor, fetched by LDS-DMA (global_load_lds_dwordx4, 8 rows x
//   128 B per instruction) into a wave-private, XOR-swizzled image (swizzle on the per-lane SOURCE
//   address, LDS destination linear) and double-buffered, so tile n+1 streams in while tile n is in
//   the MFMAs / gate / stores.  (Fragment-shaped global loads touch 32-64 cache lines per instruction
//   and measured ~5 us of the 17 us kernel.)  Outputs leave as whole rows through a wave-private
//   transposing LDS stage.  LDS also holds the layer's packed weights (A fragments), filled by LDS-DMA.
// ------------------------------------------------------------------------------------------
template <typename T> struct XTile {
  static constexpr int VEC = 16 / (int)sizeof(T);
  // element offset of 16-byte chunk c of row r in a swizzled [32][R] image
  template <int R> static __device__ __forceinline__ int off(int r, int c) {
    constexpr int CPR = R / VEC;
    return r * R + ((c ^ (r & (CPR - 1))) * VEC);
  }
};

// B fragment (natural k order) of k-step ks for time row `r`, lane half h, from a swizzled x image
template <int R> __device__ __forceinline__ Frag<bf16_t> xfrag(const bf16_t* img, int r, int ks, int h) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(img + XTile<bf16_t>::off<R>(r, 2 * ks + h));
  return f;
}
template <int R> __device__ __forceinline__ Frag<float> xfrag(const float* img, int r, int ks, int h) {
  Frag<float> f;
  f.lo = *reinterpret_cast<const f32x4*>(img + XTile<float>::off<R>(r, 4 * ks + 2 * h));
  f.hi = *reinterpret_cast<const f32x4*>(img + XTile<float>::off<R>(r, 4 * ks + 2 * h + 1));
  return f;
}
// 4 consecutive channels (accumulator group) starting at channel ch0 (multiple of 4) of row r
template <int R> __device__ __forceinline__ f32x4 xquad(const bf16_t* img, int r, int ch0) {
  return load4(img + XTile<bf16_t>::off<R>(r, ch0 >> 3) + (ch0 & 7));
}
template <int R> __device__ __forceinline__ f32x4 xquad(const float* img, int r, int ch0) {
  return load4(img + XTile<float>::off<R>(r, ch0 >> 2));
}

template <typename T, int RT, int K, bool COND, int NBUF>
__global__ __launch_bounds__(256) void layer_fwd_kernel(const T* __restrict__ x, const T* __restrict__ cond,
                                                        const T* __restrict__ wconv, const T* __restrict__ wres,
                                                        const float* __restrict__ bias_f,
                                                        const float* __restrict__ bias_r, T* __restrict__ h_out,
                                                        T* __restrict__ z_out, int Tlen, int dilation,
                                                        int cond_frames, int pool, int cond_stride, int ntb,
                                                        int ntiles) {
  constexpr int R = 32 * RT;
  constexpr int KS = R / 16;  // k-steps per tap
  constexpr int NCONV = RT * K * KS, NRES = RT * KS;
  constexpr int VEC = 16 / (int)sizeof(T), CPR = R / VEC;       // 16-B chunks per row
  constexpr int TILE_E = 32 * R;                                // elements of one tap image
  constexpr int PIECES = TILE_E * (int)sizeof(T) / 1024;        // 1-KiB LDS-DMA pieces per tap image
  constexpr int RPP = 32 / PIECES;                              // rows per piece
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<T>* lds_conv = reinterpret_cast<Frag<T>*>(smem);
  Frag<T>* lds_res = lds_conv + NCONV * 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  T* stage = reinterpret_cast<T*>(lds_res + NRES * 64) + wave * (32 * RowStage<T>::stride(R));   // wave-private
  constexpr int NIMG = NBUF ? NBUF : 1;   // NBUF 0: one image set, the next tile waits in registers
  T* ximg = reinterpret_cast<T*>(lds_res + NRES * 64) + 4 * (32 * RowStage<T>::stride(R)) +
            wave * (NIMG * K * TILE_E);                                                          // [NIMG][K][32][R]
  lds_dma_copy(wconv, lds_conv, NCONV * 64 * (int)sizeof(Frag<T>), wave, lane, 4);
  lds_dma_copy(wres, lds_res, NRES * 64 * (int)sizeof(Frag<T>), wave, lane, 4);

  const int col = lane & 31, half = lane >> 5;
  float bf[RT][16], br[RT][16];
#pragma unroll
  for (int mt = 0; mt < RT; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      bf[mt][q] = bias_f[32 * mt + crow(q, half)];
      br[mt][q] = bias_r[32 * mt + crow(q, half)];
    }

  // LDS-DMA of the K tap windows of one tile (rows clamped into the clip; out-of-range taps are
  // zeroed at use).  Always exactly K*PIECES instructions, so the vmcnt bookkeeping below is exact.
  auto fetch = [&](int tile_, int buf) {
    const int tile = tile_ < ntiles ? tile_ : ntiles - 1;
    const int b = tile / ntb;
    const int t0 = (tile - b * ntb) * 32;
    const T* xb = x + (size_t)b * Tlen * R;
    const int rl0 = lane / CPR, slot = lane % CPR;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      T* img = ximg + (buf * K + k) * TILE_E;
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        const int rl = p * RPP + rl0;                       // row within the tile image
        int tr = t0 + rl - (K - 1 - k) * dilation;          // source time row
        tr = tr < 0 ? 0 : (tr < Tlen ? tr : Tlen - 1);
        const int c = slot ^ (rl & (CPR - 1));              // source chunk (swizzle on the source side)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xb + (size_t)tr * R + c * VEC),
                                         (__attribute__((address_space(3))) void*)(img + p * (1024 / (int)sizeof(T))), 16, 0, 0);
      }
    }
  };

  // NBUF 0: the same whole-row pieces fetched into registers (in flight while the previous tile is processed) and
  // dropped into the image with ds_write_b128 -- the compiler tracks these loads per register, so the prefetch
  // costs no LDS and needs no hand-counted vmcnt.
  f32x4 xr[K][PIECES];
  auto load_regs = [&](int tile) {
    const int b = tile / ntb;
    const int t0 = (tile - b * ntb) * 32;
    const T* xb = x + (size_t)b * Tlen * R;
    const int rl0 = lane / CPR, slot = lane % CPR;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        const int rl = p * RPP + rl0;
        int tr = t0 + rl - (K - 1 - k) * dilation;
        tr = tr < 0 ? 0 : (tr < Tlen ? tr : Tlen - 1);
        const int c = slot ^ (rl & (CPR - 1));
        xr[k][p] = *reinterpret_cast<const f32x4*>(xb + (size_t)tr * R + c * VEC);
      }
  };
  auto put_regs = [&]() {
    wave_lds_order();                     // the previous tile's reads of the image are done
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int p = 0; p < PIECES; ++p)
        *reinterpret_cast<f32x4*>(ximg + k * TILE_E + p * (1024 / (int)sizeof(T)) + lane * VEC) = xr[k][p];
    wave_lds_order();
  };

  // one tile: conv MFMAs -> tanh/gate -> residual MFMAs -> whole-row stores
  auto process = [&](int tile, int buf) {
    const int b = tile / ntb;
    const int t0 = (tile - b * ntb) * 32;
    const int tc = t0 + col;
    const int rows_valid = Tlen - t0;   // >= 1; rows beyond it do not exist
    const T* cb = COND ? cond + (size_t)b * cond_frames * cond_stride : nullptr;
    Frag<T> cur[K][KS];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tk = tc - (K - 1 - k) * dilation;
      const bool valid = (tc < Tlen) && (tk >= 0);
      const T* img = ximg + (buf * K + k) * TILE_E;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const Frag<T> f = xfrag<R>(img, col, ks, half);
        cur[k][ks] = valid ? f : zero_frag<T>();
      }
    }

    // ---- dilated causal conv as one (K*R)-deep contraction; accumulators start at the bias
    f32x16 accF[RT];
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) accF[mt][q] = bf[mt][q];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) {
          const Frag<T> a = lds_conv[(mt * (K * KS) + k * KS + ks) * 64 + lane];
          mma(accF[mt], a, cur[k][ks]);
        }

    // ---- tanh, gate, 1x1 residual from registers, scaled residual add
    T* ztile = z_out + ((size_t)b * Tlen + t0) * R;
    T* htile = h_out + ((size_t)b * Tlen + t0) * R;
    Frag<T> cf[KS];
    {
      float zz[RT][16];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float z = Math<T>::tanh_(accF[mt][q]);
          zz[mt][q] = z;
          cf[2 * mt + (q >> 3)].set(q & 7, gate_of_z<T>(z));
        }
      store_rows_via_lds<T, RT>(stage, ztile, R, zz, rows_valid, lane);
    }
    f32x16 accR[RT];
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) accR[mt][q] = br[mt][q];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int mt = 0; mt < RT; ++mt) {
        const Frag<T> a = lds_res[(mt * KS + s) * 64 + lane];
        mma(accR[mt], a, cf[s]);
      }
    {
      // residual operand x(+cond) in accumulator layout straight from the tap-(K-1) image
      const T* img = ximg + (buf * K + (K - 1)) * TILE_E;
      const bool ok = tc < Tlen;
      float hv[RT][16];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 xv = xquad<R>(img, col, 32 * mt + 8 * g + 4 * half);
          f32x4 cv = {0.f, 0.f, 0.f, 0.f};
          if (COND)   // the NEXT layer's conditioning bias, so that the stored row is that layer's complete input
            cv = load4(cb + (size_t)((ok ? tc : 0) / pool) * cond_stride + 32 * mt + 8 * g + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[mt][4 * g + e] = (xv[e] + accR[mt][4 * g + e]) * kSqrtHalf + cv[e];
        }
      store_rows_via_lds<T, RT>(stage, htile, R, hv, rows_valid, lane);
    }
  };

  const int stride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  if (NBUF == 0) {
    load_regs(tile < ntiles ? tile : ntiles - 1);
    __syncthreads();   // weights landed
    while (tile < ntiles) {
      put_regs();
      if (tile + stride < ntiles) load_regs(tile + stride);
      process(tile, 0);
      tile += stride;
    }
    return;
  }
  fetch(tile, 0);
  __syncthreads();   // weights + first tile landed (vmcnt(0) + barrier)
  if (NBUF == 2) {
    int buf = 0;
    bool prev_full = true;
    while (tile < ntiles) {
      fetch(tile + stride, buf ^ 1);                  // K*PIECES LDS-DMA ops for the next tile
      // tile `tile` was fetched one iteration ago; younger ops: the previous tile's 2*32/RPI-row-store
      // instructions (all issued only if that tile was full) and the K*PIECES ops just issued.
      if (prev_full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K * PIECES + 2 * (32 * R * (int)sizeof(T) / 1024)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K * PIECES) : "memory");
      process(tile, buf);
      {
        const int b = tile / ntb;
        prev_full = (Tlen - (tile - b * ntb) * 32) >= 32;
      }
      tile += stride;
      buf ^= 1;
    }
  } else {
    while (tile < ntiles) {
      process(tile, 0);
      tile += stride;
      if (tile < ntiles) {
        fetch(tile, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
}

static int layer_blocks_per_cu(const char* env, int dflt) {
  const char* e = getenv(env);
  int v = e ? atoi(e) : dflt;
  return v < 1 ? 1 : (v > 8 ? 8 : v);
}

template <typename T, int RT, int K, int NBUF>
static int launch_layer_fwd_n(const void* x, const void* cond, const void* wconv, const void* wres,
                              const float* bias_f, const float* bias_r, void* h_out, void* z_out, int B, int Tlen,
                              int dilation, int cond_frames, int pool, int cond_stride, hipStream_t st) {
  constexpr int R = 32 * RT, KS = R / 16;
  const size_t sh = (size_t)(RT * K * KS + RT * KS) * 64 * sizeof(Frag<T>) +
                    (size_t)4 * 32 * RowStage<T>::stride(R) * sizeof(T) +
                    (size_t)4 * (NBUF ? NBUF : 1) * K * 32 * R * sizeof(T);
  const int ntb = (Tlen + 31) / 32;
  const long long ntiles = (long long)B * ntb;
  static const int bpc = layer_blocks_per_cu("SRWN_FWD_BPC", 2);   // single-buffered images: 74 KB LDS -> 2 blocks/CU
  long long blocks = (ntiles + 3) / 4;
  if (blocks > 256LL * bpc) blocks = 256LL * bpc;
  dim3 grid((unsigned)blocks), block(256);
  if (cond) {
    auto kfn = layer_fwd_kernel<T, RT, K, true, NBUF>;
    if (sh > 32768) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(kfn, grid, block, sh, st, (const T*)x, (const T*)cond, (const T*)wconv, (const T*)wres, bias_f,
                       bias_r, (T*)h_out, (T*)z_out, Tlen, dilation, cond_frames, pool, cond_stride, ntb, (int)ntiles);
  } else {
    auto kfn = layer_fwd_kernel<T, RT, K, false, NBUF>;
    if (sh > 32768) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(kfn, grid, block, sh, st, (const T*)x, (const T*)nullptr, (const T*)wconv, (const T*)wres,
                       bias_f, bias_r, (T*)h_out, (T*)z_out, Tlen, dilation, 1, 1, R, ntb, (int)ntiles);
  }
  return check_launch("residual_layer_fwd");
}

template <typename T, int RT, int K>
static int launch_layer_fwd(const void* x, const void* cond, const void* wconv, const void* wres,
                            const float* bias_f, const float* bias_r, void* h_out, void* z_out, int B, int Tlen,
                            int dilation, int cond_frames, int pool, int cond_stride, hipStream_t st) {
  // bf16: operand windows prefetched into registers one tile ahead (NBUF = 0; the LDS-DMA variants measured no faster);
  // fp32: one LDS buffer
  if (sizeof(T) == 2) return launch_layer_fwd_n<T, RT, K, 0>(x, cond, wconv, wres, bias_f, bias_r, h_out, z_out, B, Tlen, dilation, cond_frames, pool, cond_stride, st);
  return launch_layer_fwd_n<T, RT, K, 1>(x, cond, wconv, wres, bias_f, bias_r, h_out, z_out, B, Tlen, dilation, cond_frames, pool, cond_stride, st);
}


extern "C" int srwn_residual_layer_fwd(const void* x, const void* cond, const void* wconv, const void* wres,
                                       const float* bias_f, const float* bias_r, void* h_out, void* z_out,
                                       int32_t B, int32_t T, int32_t R, int32_t K, int32_t dilation,
                                       int32_t cond_frames, int32_t pool_stride, int32_t cond_row_stride,
                                       int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!x || !wconv || !wres || !bias_f || !bias_r || !h_out || !z_out)
    return set_error(SRWN_E_NULL, "residual_layer_fwd: null pointer");
  if (B < 0 || T < 0 || dilation < 1) return set_error(SRWN_E_SHAPE, "residual_layer_fwd: B=%d T=%d d=%d", B, T, dilation);
  if ((long long)B * ((T + 31) / 32) > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "residual_layer_fwd: too many tiles");
  if (cond && (pool_stride < 1 || cond_row_stride < R || cond_row_stride % 8 || (int64_t)cond_frames * pool_stride < T))
    return set_error(SRWN_E_SHAPE, "residual_layer_fwd: cond frames %d x pool %d < T %d", cond_frames, pool_stride, T);
  if (K != 2) return set_error(SRWN_E_UNSUPPORTED, "residual_layer_fwd: filter_width %d (only 2 is built)", K);
  hipStream_t st = (hipStream_t)stream;
#define SRWN_LF(TT, RT_) \
  return launch_layer_fwd<TT, RT_, 2>(x, cond, wconv, wres, bias_f, bias_r, h_out, z_out, B, T, dilation, cond_frames, pool_stride, cond_row_stride, st)
  if (dtype == SRWN_BF16) {
    if (R == 32) SRWN_LF(bf16_t, 1);
    if (R == 64) SRWN_LF(bf16_t, 2);
  } else if (dtype == SRWN_F32) {
    if (R == 32) SRWN_LF(float, 1);
    if (R == 64) SRWN_LF(float, 2);
  } else {
    return set_error(SRWN_E_DTYPE, "residual_layer_fwd: dtype %d", dtype);
  }
#undef SRWN_LF
  return set_error(SRWN_E_UNSUPPORTED, "residual_layer_fwd: dilation_channels %d (built: 32, 64)", R);
}

// ------------------------------------------------------------------------------------------
// pointwise linear: y^T[n][row] = epi(bias[n] + sum_k W[k][n] * pro(x[row][k]))
//   grid = (ceil(rows / (4*32*NT)), cout_pad / (32*MT)); wave = NT column tiles x MT row tiles.
//   A fragments stream from the packed global image (L2-resident; identical for all waves).
// ------------------------------------------------------------------------------------------
struct PwArgs {
  const void* x; int64_t x_row_stride; int64_t x_chunk_stride; int chunk_len; int ks_total;
  const void* wpack; const float* bias; void* y; int64_t y_row_stride; int cout_valid; int64_t rows;
  const void* aux; int64_t aux_row_stride;
  int ks_per_split; int64_t y_split_stride;   // grid.z = k-splits: slice z covers k-steps [z*ks_per_split, ...) -> y + z*stride
  int y_chunk_len; int64_t y_chunk_stride;    // > 0: output channel n at y + (n / y_chunk_len)*y_chunk_stride + row*y_row_stride + n % y_chunk_len
};

template <typename T, int PRO>
__device__ __forceinline__ Frag<T> pw_load_b(const PwArgs& a, int64_t row, bool valid, int ks, int half) {
  if (!valid) return zero_frag<T>();
  const int kg = 16 * ks;
  const int chunk = kg / a.chunk_len, within = kg - chunk * a.chunk_len;
  const T* p = reinterpret_cast<const T*>(a.x) + (int64_t)chunk * a.x_chunk_stride + row * a.x_row_stride + within +
               8 * half;
  Frag<T> f = load_nat(p);
  if (PRO == SRWN_PRO_GATE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.set(j, gate_of_z<T>(f.get(j)));
  }
  return f;
}

template <typename T, int MT, int NT, int PRO, int EPI>
__global__ __launch_bounds__(256) void pw_linear_kernel(PwArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int64_t row_wave = ((int64_t)blockIdx.x * 4 + wave) * (32 * NT);
  if (row_wave >= a.rows) return;
  const int mb = blockIdx.y;  // block of MT row tiles
  const Frag<T>* wp = reinterpret_cast<const Frag<T>*>(a.wpack) + (size_t)mb * MT * a.ks_total * 64 + lane;
  const int ks_lo = blockIdx.z * a.ks_per_split;
  const int ks_hi = (ks_lo + a.ks_per_split < a.ks_total) ? ks_lo + a.ks_per_split : a.ks_total;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n = 32 * (mb * MT + mt) + crow(q, half);
      const float bv = (a.bias && n < a.cout_valid && blockIdx.z == 0) ? a.bias[n] : 0.0f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt][q] = bv;
    }

  int64_t rowc[NT];
  bool valid[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    rowc[nt] = row_wave + 32 * nt + col;
    valid[nt] = rowc[nt] < a.rows;
  }

  for (int ks = ks_lo; ks < ks_hi; ++ks) {
    Frag<T> bf[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[nt] = pw_load_b<T, PRO>(a, rowc[nt], valid[nt], ks, half);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const Frag<T> af = wp[((size_t)mt * a.ks_total + ks) * 64];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) mma(acc[mt][nt], af, bf[nt]);
    }
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (!valid[nt]) continue;
    T* yrow = reinterpret_cast<T*>(a.y) + rowc[nt] * a.y_row_stride;
    const T* arow = (EPI == SRWN_EPI_MASK) ? reinterpret_cast<const T*>(a.aux) + rowc[nt] * a.aux_row_stride : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = 32 * (mb * MT + mt) + 8 * g + 4 * half;
        if (n0 >= a.cout_valid) continue;  // cout_valid is a multiple of 4 (checked on the host)
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * g + e];
        if (EPI == SRWN_EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        } else if (EPI == SRWN_EPI_MASK) {
          const f32x4 m = load4(arow + n0);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (m[e] > 0.0f) ? v[e] : 0.0f;
        }
        if (EPI == SRWN_EPI_F32)
          store4(reinterpret_cast<float*>(a.y) + (int64_t)blockIdx.z * a.y_split_stride + rowc[nt] * a.y_row_stride + n0,
                 v[0], v[1], v[2], v[3]);
        else if (a.y_chunk_len > 0)
          store4(yrow + (int64_t)(n0 / a.y_chunk_len) * a.y_chunk_stride + n0 % a.y_chunk_len, v[0], v[1], v[2], v[3]);
        else store4(yrow + n0, v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <typename T, int MT, int NT>
static int launch_pw(const PwArgs& a, int cout_pad, int pro, int epi, hipStream_t st) {
  const int nsplit = (a.ks_total + a.ks_per_split - 1) / a.ks_per_split;
  dim3 grid((unsigned)((a.rows + 4 * 32 * NT - 1) / (4 * 32 * NT)), (unsigned)(cout_pad / (32 * MT)), (unsigned)nsplit),
      block(256);
#define SRWN_PW(P, E)                                                                                   \
  if (pro == P && epi == E) {                                                                           \
    hipLaunchKernelGGL((pw_linear_kernel<T, MT, NT, P, E>), grid, block, 0, st, a);                     \
    return check_launch("pw_linear");                                                                   \
  }
  SRWN_PW(SRWN_PRO_NONE, SRWN_EPI_NONE)
  SRWN_PW(SRWN_PRO_NONE, SRWN_EPI_RELU)
  SRWN_PW(SRWN_PRO_NONE, SRWN_EPI_MASK)
  SRWN_PW(SRWN_PRO_GATE, SRWN_EPI_NONE)
  SRWN_PW(SRWN_PRO_GATE, SRWN_EPI_RELU)
  SRWN_PW(SRWN_PRO_NONE, SRWN_EPI_F32)
#undef SRWN_PW
  return set_error(SRWN_E_UNSUPPORTED, "pw_linear: pro %d / epi %d combination not built", pro, epi);
}

static int pw_dispatch(const PwArgs& a, int cout_pad, int pro, int epi, int dtype, hipStream_t st);

extern "C" int srwn_pw_linear(const void* x, int64_t x_row_stride, int64_t x_chunk_stride, int32_t chunk_len,
                              int32_t Cin, const void* wpack, const float* bias, void* y, int64_t y_row_stride,
                              int32_t cout_pad, int32_t cout_valid, int64_t rows, const void* aux,
                              int64_t aux_row_stride, int32_t pro, int32_t epi, int32_t dtype, void* stream) {
  if (rows == 0) return 0;
  if (!x || !wpack || !y) return set_error(SRWN_E_NULL, "pw_linear: null pointer");
  if (epi == SRWN_EPI_MASK && !aux) return set_error(SRWN_E_NULL, "pw_linear: EPI_MASK needs aux");
  if (rows < 0 || Cin < 16 || Cin % 16 || chunk_len < 16 || chunk_len % 16 || Cin % chunk_len || cout_pad < 32 ||
      cout_pad % 32 || cout_valid < 4 || cout_valid % 4 || cout_valid > cout_pad)
    return set_error(SRWN_E_SHAPE, "pw_linear: rows=%lld Cin=%d chunk=%d cout_pad=%d cout_valid=%d", (long long)rows,
                     Cin, chunk_len, cout_pad, cout_valid);
  hipStream_t st = (hipStream_t)stream;
  if (epi != SRWN_EPI_F32) {
    int rc = 0;
    if (rowgemm_dispatch(x, x_row_stride, x_chunk_stride, chunk_len, Cin, wpack, bias, y, y_row_stride, cout_pad,
                         cout_valid, rows, aux, aux_row_stride, nullptr, nullptr, nullptr, 0.0f, pro, epi, dtype, st,
                         &rc))
      return rc;
  }
  PwArgs a{x, x_row_stride, x_chunk_stride, chunk_len, Cin / 16, wpack, bias, y, y_row_stride, cout_valid, rows, aux,
           aux_row_stride, Cin / 16, 0, 0, 0};
  return pw_dispatch(a, cout_pad, pro, epi, dtype, st);
}

// The same product with its outputs in chunks: channel n at y + (n / y_chunk_len)*y_chunk_stride + row*y_row_stride +
// n % y_chunk_len.  The conditioning biases of all layers (model.py:180) are one product with L*R outputs; stored as
// [rows, L*R] a layer's 128-byte rows are 3 840 bytes apart, stored layer by layer ([L][rows][R]) they are as dense as
// every other operand of the layer kernels (the conditioned forward group kernel: 83 -> 73 us per launch).
extern "C" int srwn_pw_linear_ychunks(const void* x, int64_t x_row_stride, int32_t Cin, const void* wpack,
                                      const float* bias, void* y, int64_t y_row_stride, int32_t y_chunk_len,
                                      int64_t y_chunk_stride, int32_t cout_pad, int32_t cout_valid, int64_t rows,
                                      int32_t dtype, void* stream) {
  if (rows == 0) return 0;
  if (!x || !wpack || !y) return set_error(SRWN_E_NULL, "pw_linear_ychunks: null pointer");
  if (rows < 0 || Cin < 16 || Cin % 16 || cout_pad < 32 || cout_pad % 32 || cout_valid < 4 || cout_valid % 4 ||
      cout_valid > cout_pad || y_chunk_len < 4 || y_chunk_len % 4 || y_row_stride < y_chunk_len || y_chunk_stride < 0)
    return set_error(SRWN_E_SHAPE, "pw_linear_ychunks: rows=%lld Cin=%d cout_pad=%d cout_valid=%d chunk=%d", (long long)rows,
                     Cin, cout_pad, cout_valid, y_chunk_len);
  PwArgs a{x, x_row_stride, 0, Cin, Cin / 16, wpack, bias, y, y_row_stride, cout_valid, rows, nullptr, 0, Cin / 16, 0,
           y_chunk_len, y_chunk_stride};
  return pw_dispatch(a, cout_pad, SRWN_PRO_NONE, SRWN_EPI_NONE, dtype, (hipStream_t)stream);
}

static int pw_dispatch(const PwArgs& a, int cout_pad, int pro, int epi, int dtype, hipStream_t st) {
  const int tiles = cout_pad / 32;
  // few rows (the conditioning product: B*frames = 1 024 rows, K = 16): one row tile per wave, or the launch is 60
  // workgroups of 128 stores per lane -- 19 us for 63 MFLOP
  const bool few = ((a.rows + 255) / 256) * (tiles / 4) < 128;
  if (dtype == SRWN_BF16) {
    if (few) return launch_pw<bf16_t, 1, 2>(a, cout_pad, pro, epi, st);
    if (tiles % 4 == 0) return launch_pw<bf16_t, 4, 2>(a, cout_pad, pro, epi, st);
    if (tiles % 2 == 0) return launch_pw<bf16_t, 2, 2>(a, cout_pad, pro, epi, st);
    return launch_pw<bf16_t, 1, 2>(a, cout_pad, pro, epi, st);
  } else if (dtype == SRWN_F32) {
    if (tiles % 4 == 0) return launch_pw<float, 4, 1>(a, cout_pad, pro, epi, st);
    if (tiles % 2 == 0) return launch_pw<float, 2, 1>(a, cout_pad, pro, epi, st);
    return launch_pw<float, 1, 1>(a, cout_pad, pro, epi, st);
  }
  return set_error(SRWN_E_DTYPE, "pw_linear: dtype %d", dtype);
}

// The same product split over the contraction axis: slice z of `nsplit` covers Cin/nsplit inputs and writes its
// fp32 partial to y_partials + z*rows*y_row_stride (bias in slice 0); finish with srwn_reduce_partials.  For
// few-row products with a long contraction (the pooled skip sum of the encoder: 1024 rows x K = L*128).
extern "C" int srwn_pw_linear_ksplit(const void* x, int64_t x_row_stride, int64_t x_chunk_stride, int32_t chunk_len,
                                     int32_t Cin, const void* wpack, const float* bias, float* y_partials,
                                     int64_t y_row_stride, int32_t cout_pad, int32_t cout_valid, int64_t rows,
                                     int32_t nsplit, int32_t dtype, void* stream) {
  if (rows == 0) return 0;
  if (!x || !wpack || !y_partials) return set_error(SRWN_E_NULL, "pw_linear_ksplit: null pointer");
  if (rows < 0 || Cin < 16 || Cin % 16 || chunk_len < 16 || chunk_len % 16 || Cin % chunk_len || cout_pad < 32 ||
      cout_pad % 32 || cout_valid < 4 || cout_valid % 4 || cout_valid > cout_pad || nsplit < 1 || nsplit > 1024 ||
      (Cin / 16) % nsplit)
    return set_error(SRWN_E_SHAPE, "pw_linear_ksplit: rows=%lld Cin=%d chunk=%d cout_pad=%d cout_valid=%d nsplit=%d",
                     (long long)rows, Cin, chunk_len, cout_pad, cout_valid, nsplit);
  PwArgs a{x, x_row_stride, x_chunk_stride, chunk_len, Cin / 16, wpack, bias, y_partials, y_row_stride, cout_valid, rows,
           nullptr, 0, (Cin / 16) / nsplit, rows * y_row_stride, 0, 0};
  return pw_dispatch(a, cout_pad, SRWN_PRO_NONE, SRWN_EPI_F32, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// last 1x1 + per-time-step softmax cross-entropy, fused in registers.
//   One wave = one column tile of 32 time steps x ALL classes (MT = cout_pad/32 row tiles): a lane
//   holds half of its time step's logits, lane^32 the other half -> one cross-half exchange.
// ------------------------------------------------------------------------------------------
struct HeadArgs {
  const void* x; int64_t x_row_stride; int ks_total; const void* wpack; const float* bias;
  const int32_t* targets; float* loss_partials; void* dlogits; float* logits_out;
  int cout_valid; int64_t rows; float grad_scale;
};

template <typename T, int MT>
__global__ __launch_bounds__(256) void head_softmax_ce_kernel(HeadArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t row_wave = tile * 32;
  if (row_wave >= a.rows) return;
  const int64_t row = row_wave + col;
  const bool valid = row < a.rows;
  const Frag<T>* wp = reinterpret_cast<const Frag<T>*>(a.wpack) + lane;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n = 32 * mt + crow(q, half);
      acc[mt][q] = (a.bias && n < a.cout_valid) ? a.bias[n] : 0.0f;
    }
  const T* xr = reinterpret_cast<const T*>(a.x) + (valid ? row : 0) * a.x_row_stride + 8 * half;

  for (int ks = 0; ks < a.ks_total; ++ks) {
    const Frag<T> bf = valid ? load_nat(xr + 16 * ks) : zero_frag<T>();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const Frag<T> af = wp[((size_t)mt * a.ks_total + ks) * 64];
      mma(acc[mt], af, bf);
    }
  }

  // log-softmax over the class axis (ops.py:111-115): registers x two lane halves
  const int tgt = valid ? a.targets[row] : -1;
  float m = -INFINITY, vt = 0.0f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n = 32 * mt + crow(q, half);
      if (n < a.cout_valid) m = fmaxf(m, acc[mt][q]);
      if (n == tgt) vt = acc[mt][q];
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  vt += __shfl_xor(vt, 32);
  float s = 0.0f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n = 32 * mt + crow(q, half);
      if (n < a.cout_valid) s += expf(acc[mt][q] - m);
    }
  s += __shfl_xor(s, 32);
  const float lse = m + logf(s);
  float loss = (valid && half == 0) ? (lse - vt) : 0.0f;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) loss += __shfl_xor(loss, off);  // sums within each 32-lane half
  if (lane == 0) a.loss_partials[tile] = loss;

  if (!valid) return;
  if (a.logits_out) {
    float* lr = a.logits_out + row * (int64_t)a.cout_valid;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = 32 * mt + crow(q, half);
        if (n < a.cout_valid) lr[n] = acc[mt][q];
      }
  }
  if (a.dlogits) {
    T* dr = reinterpret_cast<T*>(a.dlogits) + row * (int64_t)(32 * MT);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = 32 * mt + 8 * g + 4 * half + e;
          const float p = (n < a.cout_valid) ? expf(acc[mt][4 * g + e] - lse) : 0.0f;
          v[e] = (p - (n == tgt ? 1.0f : 0.0f)) * a.grad_scale;
        }
        store4(dr + 32 * mt + 8 * g + 4 * half, v[0], v[1], v[2], v[3]);
      }
  }
}

extern "C" int64_t srwn_softmax_ce_partials(int64_t rows) { return (rows + 31) / 32; }

extern "C" int srwn_head_softmax_ce(const void* x, int64_t x_row_stride, int32_t Cin, const void* wpack,
                                    const float* bias, const int32_t* targets, float* loss_partials, void* dlogits,
                                    float* logits_out, int32_t cout_pad, int32_t cout_valid, int64_t rows,
                                    float grad_scale, int32_t dtype, void* stream) {
  if (rows == 0) return 0;
  if (!x || !wpack || !targets || !loss_partials) return set_error(SRWN_E_NULL, "head_softmax_ce: null pointer");
  if (rows < 0 || Cin < 16 || Cin % 16 || cout_pad % 32 || cout_pad < 32 || cout_pad > 256 || cout_valid < 1 ||
      cout_valid > cout_pad)
    return set_error(SRWN_E_SHAPE, "head_softmax_ce: rows=%lld Cin=%d cout_pad=%d cout_valid=%d (cout_pad <= 256)",
                     (long long)rows, Cin, cout_pad, cout_valid);
  {
    int rc = 0;
    if (rowgemm_dispatch(x, x_row_stride, 0, Cin, Cin, wpack, bias, dlogits, cout_pad, cout_pad, cout_valid, rows,
                         nullptr, 0, targets, loss_partials, logits_out, grad_scale, SRWN_PRO_NONE, 3, dtype,
                         (hipStream_t)stream, &rc))
      return rc;
  }
  HeadArgs a{x, x_row_stride, Cin / 16, wpack, bias, targets, loss_partials, dlogits, logits_out, cout_valid, rows,
             grad_scale};
  const int64_t tiles = (rows + 31) / 32;
  dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int mt = cout_pad / 32;
#define SRWN_HD(TT, M)                                                                  \
  if (mt == M) {                                                                        \
    hipLaunchKernelGGL((head_softmax_ce_kernel<TT, M>), grid, block, 0, st, a);         \
    return check_launch("head_softmax_ce");                                             \
  }
  if (dtype == SRWN_BF16) {
    SRWN_HD(bf16_t, 1) SRWN_HD(bf16_t, 2) SRWN_HD(bf16_t, 3) SRWN_HD(bf16_t, 4) SRWN_HD(bf16_t, 5) SRWN_HD(bf16_t, 6)
    SRWN_HD(bf16_t, 7) SRWN_HD(bf16_t, 8)
  } else if (dtype == SRWN_F32) {
    SRWN_HD(float, 1) SRWN_HD(float, 2) SRWN_HD(float, 3) SRWN_HD(float, 4) SRWN_HD(float, 5) SRWN_HD(float, 6)
    SRWN_HD(float, 7) SRWN_HD(float, 8)
  } else {
    return set_error(SRWN_E_DTYPE, "head_softmax_ce: dtype %d", dtype);
  }
#undef SRWN_HD
  return set_error(SRWN_E_UNSUPPORTED, "head_softmax_ce: cout_pad %d", cout_pad);
}
