// Weight gradient of 256-wide products as one time-contraction GEMM:
//   out[m][n] = sum_rows pro(A[row][m]) * D[row][n],   n < 256,  m in chunks of 64 channels
// Used for the sum of all skip 1x1s (A = stack of 30 z tensors -> out = dWs of every layer, 1920 x 256,
// dtotal shared by all layers and therefore re-read only once per group of 4 layers) and for the
// two head 1x1s (A = r0 / r1, 256 x 256).
//
// One workgroup = 8 waves = 256 (m) x 256 (n) output tile over one slab of rows; 32-row chunks of A and
// D are register-staged (loads for chunk i+1 issued before the MFMAs of chunk i, written to the other
// LDS buffer after them: one barrier per chunk) in natural [row][channel] layout with a row pad that
// makes the transposing reads (ds_read_b64_tr_b16) bank-conflict free.  fp32 partials per slab, summed
// in fixed order by srwn_reduce_partials (deterministic; no atomics).
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

struct Wg2Args {
  const void* a; int64_t a_chunk_stride; int64_t a_row_stride; int m_chunks;
  const void* d; int64_t d_row_stride;
  float* partials; float* bias_partials;
  int64_t rows; int rows_per_slab; int nslabs;
};
// up to two independent products of the same shape in one launch (grid.z): the head's two 1x1s are 46 us each alone,
// 256 workgroups of 4-16 MFMAs per barrier -- side by side their workgroups share the CUs and hide each other's waits
struct Wg2Pair { Wg2Args p[2]; };

namespace {

template <typename T> struct Ld2;
template <> struct Ld2<bf16_t> {
  static __device__ __forceinline__ Frag<bf16_t> load(const bf16_t* tile, int stride, int row0, int col0, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1;
    const bf16_t* base = tile + (size_t)(row0 + 8 * h + q) * stride + col0 + 16 * (g & 1) + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * stride));
    Frag<bf16_t> f;
    f.v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    return f;
  }
};
template <> struct Ld2<float> {
  static __device__ __forceinline__ Frag<float> load(const float* tile, int stride, int row0, int col0, int lane) {
    const int c = col0 + (lane & 31), h = lane >> 5;
    const float* base = tile + (size_t)(row0 + 8 * h) * stride + c;
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.set(j, base[(size_t)j * stride]);
    return f;
  }
};

constexpr int kRows = 32;      // rows per staged chunk (64 rows per chunk for the single-chunk blocks measured the same: 46 us)
constexpr int kStride = 288;   // LDS row stride in elements: 576 B (bf16) puts 4 consecutive rows on disjoint banks

// DW = width of D (256, or 128 for the reference scripts' skip_channels); CW = width of one A chunk (64, or 32 for
// their dilation_channels): the block still stages MB*64 A columns, i.e. MB*64/CW chunks.
template <typename T, int PRO, int MB, int DW, int CW>
__global__ __launch_bounds__(512) void wgrad256_kernel(Wg2Pair pp) {
  const Wg2Args& a = pp.p[blockIdx.z];
  constexpr int kW = DW;
  constexpr int NTW = (MB == 4) ? DW / 64 : 1;   // n-tiles per wave (MB = 1: one tile per wave, waves >= DW/32 idle)
  constexpr int AW = MB * 64;                // staged A tile width (channels)
  constexpr int VEC = 16 / sizeof(T);
  constexpr int VPR = kW / VEC;                  // 16-byte vectors per D tile row
  constexpr int NV = kRows * VPR / 512;          // D vectors per thread (2 bf16, 4 f32)
  constexpr int VPRA = AW / VEC;                 // 16-byte vectors per A tile row
  constexpr int NVA = (kRows * VPRA + 511) / 512;  // A vectors per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* lds = reinterpret_cast<T*>(smem);           // [2 buffers][A tile | D tile][kRows][kStride]
  auto tileA = [&](int buf) { return lds + (size_t)(buf * 2 + 0) * kRows * kStride; };
  auto tileD = [&](int buf) { return lds + (size_t)(buf * 2 + 1) * kRows * kStride; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (MB == 4) ? (wave >> 1) : 0;    // wave -> A chunk
  const int wn = (MB == 4) ? (wave & 1) : wave;  // wave -> group of NTW n-tiles
  const bool wave_live = (MB == 4) || (wave < DW / 32);
  const int slab = blockIdx.x, mblk = blockIdx.y;
  const int64_t r_begin = (int64_t)slab * a.rows_per_slab;
  const int64_t r_end = (r_begin + a.rows_per_slab < a.rows) ? r_begin + a.rows_per_slab : a.rows;
  const int nit = (r_end > r_begin) ? (int)((r_end - r_begin + kRows - 1) / kRows) : 0;
  const T* abase = reinterpret_cast<const T*>(a.a);
  const T* dbase = reinterpret_cast<const T*>(a.d);

  f32x16 acc[2][NTW];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.0f;
  // column sums of D (the bias gradient): every wave sums kRows/8 rows of the chunk for CPL columns per lane with one
  // vector read per row, and the eight partial sums meet in LDS at the end.  (One thread per column walking all 32
  // rows kept four waves ~600 cycles longer per chunk than the 512 cycles of MFMA work: with one barrier per chunk
  // and one workgroup per CU, the bias-carrying workgroups set the kernel time.)
  constexpr int CPL = kW / 64;                   // columns per lane
  float bs[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) bs[j] = 0.0f;
  const bool do_bias = (a.bias_partials != nullptr) && (mblk == 0);

  f32x4 ra[NVA], rd[NV];   // raw 16-byte vectors in flight (bit containers)
  auto gload = [&](int it) {
    const int64_t r0 = r_begin + (int64_t)it * kRows;
#pragma unroll
    for (int v = 0; v < NVA; ++v) {
      const int idx = tid + v * 512;
      const int rr = idx / VPRA, cv = (idx % VPRA) * VEC;
      const int64_t row = r0 + rr;
      const int chunk = (mblk * AW + cv) / CW;
      const bool ok = (idx < kRows * VPRA) && (row < r_end) && (chunk < a.m_chunks);
      ra[v] = ok ? *reinterpret_cast<const f32x4*>(abase + (int64_t)chunk * a.a_chunk_stride + row * a.a_row_stride + (cv % CW))
                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = tid + v * 512;
      const int rr = idx / VPR, cv = (idx % VPR) * VEC;
      const int64_t row = r0 + rr;
      rd[v] = (row < r_end) ? *reinterpret_cast<const f32x4*>(dbase + row * a.d_row_stride + cv) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int buf) {
    T* ta = tileA(buf); T* td = tileD(buf);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = tid + v * 512;
      const int rr = idx / VPR, cv = (idx % VPR) * VEC;
      *reinterpret_cast<f32x4*>(td + rr * kStride + cv) = rd[v];
    }
#pragma unroll
    for (int v = 0; v < NVA; ++v) {
      const int idx = tid + v * 512;
      if (idx >= kRows * VPRA) continue;
      const int rr = idx / VPRA, cv = (idx % VPRA) * VEC;
      f32x4 x = ra[v];
      if (PRO == SRWN_PRO_GATE) {
        if (sizeof(T) == 2) {
          bf16x8 b = __builtin_bit_cast(bf16x8, x);
#pragma unroll
          for (int e = 0; e < 8; ++e) b[e] = (bf16_t)gate_of_z<T>((float)b[e]);
          x = __builtin_bit_cast(f32x4, b);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = gate_of_z<T>(x[e]);
        }
      }
      *reinterpret_cast<f32x4*>(ta + rr * kStride + cv) = x;
    }
  };

  auto multiply = [&](int buf) {
    const T* ta = tileA(buf); const T* td = tileD(buf);
    if (do_bias) {
#pragma unroll
      for (int rr = 0; rr < kRows / 8; ++rr) {
        const T* p = td + (wave * (kRows / 8) + rr) * kStride + lane * CPL;
#pragma unroll
        for (int j = 0; j < CPL; ++j) bs[j] += (float)p[j];
      }
    }
#pragma unroll
    for (int ks = 0; ks < (wave_live ? kRows / 16 : 0); ++ks) {
      Frag<T> af[2], bf[NTW];
#pragma unroll
      for (int m = 0; m < 2; ++m) af[m] = Ld2<T>::load(ta, kStride, 16 * ks, wm * 64 + 32 * m, lane);
#pragma unroll
      for (int n = 0; n < NTW; ++n) bf[n] = Ld2<T>::load(td, kStride, 16 * ks, (wn * NTW + n) * 32, lane);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) mma(acc[m][n], af[m], bf[n]);
    }
  };
  if constexpr (sizeof(T) == 2) {
    // bf16: a ring of three register sets -- chunk it+4 is requested while chunk it is multiplied, three steps before
    // it moves to LDS (as wgrad_layer_kernel).  With one set in flight a workgroup had 20-32 KB outstanding against an
    // HBM round trip of ~2 us: 2.1 TB/s for the head gradients, 2.8 for the skip gradients.  The loads are
    // unconditional (rows and chunks clamped into the tensors, zeroed on their way to LDS) so that hipcc counts them.
    struct Ring { f32x4 a[NVA], d[NV]; };
    auto rload = [&](int it, Ring& r) {
      const int64_t r0 = r_begin + (int64_t)it * kRows;
#pragma unroll
      for (int v = 0; v < NVA; ++v) {
        int idx = tid + v * 512;
        idx = idx < kRows * VPRA ? idx : kRows * VPRA - 1;
        const int rr = idx / VPRA, cv = (idx % VPRA) * VEC;
        int64_t row = r0 + rr;
        row = row < a.rows ? row : a.rows - 1;
        int chunk = (mblk * AW + cv) / CW;
        chunk = chunk < a.m_chunks ? chunk : a.m_chunks - 1;
        r.a[v] = *reinterpret_cast<const f32x4*>(abase + (int64_t)chunk * a.a_chunk_stride + row * a.a_row_stride + (cv % CW));
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * 512;
        const int rr = idx / VPR, cv = (idx % VPR) * VEC;
        int64_t row = r0 + rr;
        row = row < a.rows ? row : a.rows - 1;
        r.d[v] = *reinterpret_cast<const f32x4*>(dbase + row * a.d_row_stride + cv);
      }
    };
    auto rstore = [&](int it, const Ring& r) {
      const int64_t r0 = r_begin + (int64_t)it * kRows;
      T* ta = tileA(it & 1); T* td = tileD(it & 1);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * 512;
        const int rr = idx / VPR, cv = (idx % VPR) * VEC;
        *reinterpret_cast<f32x4*>(td + rr * kStride + cv) = (r0 + rr < r_end) ? r.d[v] : zero;
      }
#pragma unroll
      for (int v = 0; v < NVA; ++v) {
        const int idx = tid + v * 512;
        if (idx >= kRows * VPRA) continue;
        const int rr = idx / VPRA, cv = (idx % VPRA) * VEC;
        const bool ok = (r0 + rr < r_end) && ((mblk * AW + cv) / CW < a.m_chunks);
        f32x4 x = ok ? r.a[v] : zero;
        if (PRO == SRWN_PRO_GATE) {
          bf16x8 b = __builtin_bit_cast(bf16x8, x);
#pragma unroll
          for (int e = 0; e < 8; ++e) b[e] = (bf16_t)gate_of_z<T>((float)b[e]);
          x = __builtin_bit_cast(f32x4, b);
        }
        *reinterpret_cast<f32x4*>(ta + rr * kStride + cv) = x;
      }
    };
    Ring q0, q1, q2;            // chunk c waits in set c % 3
    rload(0, q0);
    rload(1, q1);
    rload(2, q2);
    if (nit > 0) rstore(0, q0);
    rload(3, q0);
    __syncthreads();
    auto step = [&](int it, Ring& nxt) {
      if (it < nit) multiply(it & 1);
      if (it + 1 < nit) rstore(it + 1, nxt);
      rload(it + 4, nxt);
      __syncthreads();
    };
    for (int it = 0; it < nit; it += 3) {
      step(it, q1);
      step(it + 1, q2);
      step(it + 2, q0);
    }
  } else {
    if (nit > 0) {
      gload(0);
      lstore(0);
    }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      const int buf = it & 1;
      if (it + 1 < nit) gload(it + 1);
      multiply(buf);
      if (it + 1 < nit) lstore(buf ^ 1);
      __syncthreads();
    }
  }

  const int col = lane & 31, half = lane >> 5;
  const int64_t mrow0 = (int64_t)mblk * AW + wm * 64;          // first output row (flat A column) of this wave
  const int64_t mtotal = (int64_t)a.m_chunks * CW;
  if (wave_live && mrow0 < mtotal) {                             // (m_chunks*CW is a multiple of 64: host check)
    float* pbase = a.partials + ((int64_t)slab * mtotal + mrow0) * kW;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        const int o = (wn * NTW + n) * 32 + col;
#pragma unroll
        for (int q = 0; q < 16; ++q) pbase[(int64_t)(32 * m + crow(q, half)) * kW + o] = acc[m][n][q];
      }
  }
  if (do_bias) {   // (block-uniform) the loop's last barrier has passed: the tiles are free
    float* red = reinterpret_cast<float*>(smem);   // [8 waves][kW]
#pragma unroll
    for (int j = 0; j < CPL; ++j) red[wave * kW + lane * CPL + j] = bs[j];
    __syncthreads();
    if (tid < kW) {
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[w * kW + tid];
      a.bias_partials[(int64_t)slab * kW + tid] = t;
    }
  }
}

}  // namespace

static int wg2_chunks_per_block(int m64) { return m64 >= 8 ? 4 : 1; }   // m64 = A width in units of 64 columns

extern "C" int32_t srwn_wgrad_wide_slabs(int64_t rows, int32_t m_chunks, int32_t chunk_width) {
  const int m64 = (int)(((int64_t)m_chunks * chunk_width + 63) / 64);
  const int mb = wg2_chunks_per_block(m64);
  const int mblocks = (m64 + mb - 1) / mb;
  int64_t target = 256 / (mblocks > 0 ? mblocks : 1);
  if (target < 1) target = 1;
  int64_t maxs = (rows + 255) / 256;   // at least 256 rows per slab
  if (maxs < 1) maxs = 1;
  return (int32_t)(target < maxs ? target : maxs);
}

extern "C" int32_t srwn_wgrad256_slabs(int64_t rows, int32_t m_chunks) { return srwn_wgrad_wide_slabs(rows, m_chunks, 64); }

extern "C" int srwn_wgrad256(const void* a, int64_t a_chunk_stride, int64_t a_row_stride, int32_t m_chunks,
                             const void* d, int64_t d_row_stride, float* partials, float* bias_partials,
                             int64_t rows, int32_t nslabs, int32_t pro, int32_t dtype, void* stream) {
  return srwn_wgrad_wide(a, a_chunk_stride, a_row_stride, m_chunks, 64, d, d_row_stride, 256, partials, bias_partials,
                         rows, nslabs, pro, dtype, stream);
}

static int wgrad_wide_launch(const void* const* a, const void* const* d, float* const* partials,
                             float* const* bias_partials, int nsets, int64_t a_chunk_stride, int64_t a_row_stride,
                             int32_t m_chunks, int32_t chunk_width, int64_t d_row_stride, int32_t d_width, int64_t rows,
                             int32_t nslabs, int32_t pro, int32_t dtype, void* stream) {
  if (rows == 0 || m_chunks == 0) return 0;
  for (int i = 0; i < nsets; ++i)
    if (!a[i] || !d[i] || !partials[i]) return set_error(SRWN_E_NULL, "wgrad_wide: null pointer");
  if ((chunk_width != 64 && chunk_width != 32) || (d_width != 256 && d_width != 128))
    return set_error(SRWN_E_UNSUPPORTED, "wgrad_wide: chunk_width %d (64, 32) / d_width %d (256, 128)", chunk_width, d_width);
  if (rows < 0 || m_chunks < 0 || nslabs < 1 || d_row_stride < d_width || a_row_stride < chunk_width ||
      ((int64_t)m_chunks * chunk_width) % 64)
    return set_error(SRWN_E_SHAPE, "wgrad_wide: rows=%lld m_chunks=%d nslabs=%d strides a=%lld d=%lld", (long long)rows,
                     m_chunks, nslabs, (long long)a_row_stride, (long long)d_row_stride);
  int64_t rps = (rows + nslabs - 1) / nslabs;
  rps = (rps + kRows - 1) / kRows * kRows;
  Wg2Pair g;
  for (int i = 0; i < 2; ++i) {
    const int j = i < nsets ? i : 0;
    g.p[i] = Wg2Args{a[j], a_chunk_stride, a_row_stride, m_chunks, d[j], d_row_stride, partials[j], bias_partials[j], rows,
                     (int)rps, nslabs};
  }
  const int m64 = (int)((int64_t)m_chunks * chunk_width / 64);
  const int mb = wg2_chunks_per_block(m64);
  dim3 grid((unsigned)nslabs, (unsigned)((m64 + mb - 1) / mb), (unsigned)nsets), block(512);
  hipStream_t st = (hipStream_t)stream;
#define SRWN_W2(TT, P)                                                                                        \
  if (mb == 4) SRWN_W2C(TT, P, 4) else SRWN_W2C(TT, P, 1)
#define SRWN_W2C(TT, P, MBV)                                                                                  \
  {                                                                                                           \
    if (d_width == 256 && chunk_width == 64) SRWN_W2B(TT, P, MBV, 256, 64)                                    \
    if (d_width == 256 && chunk_width == 32) SRWN_W2B(TT, P, MBV, 256, 32)                                    \
    if (d_width == 128 && chunk_width == 64) SRWN_W2B(TT, P, MBV, 128, 64)                                    \
    SRWN_W2B(TT, P, MBV, 128, 32)                                                                             \
  }
#define SRWN_W2B(TT, P, MBV, DWV, CWV)                                                                        \
  {                                                                                                           \
    auto kfn = wgrad256_kernel<TT, P, MBV, DWV, CWV>;                                                         \
    const size_t sh = (size_t)4 * kRows * kStride * sizeof(TT);                                               \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    if (e != hipSuccess) return set_error((int)e, "wgrad256: LDS %zu: %s", sh, hipGetErrorString(e));         \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, g);                                                          \
    return check_launch("wgrad256");                                                                          \
  }
  if (dtype == SRWN_BF16) {
    if (pro == SRWN_PRO_GATE) SRWN_W2(bf16_t, SRWN_PRO_GATE)
    if (pro == SRWN_PRO_NONE) SRWN_W2(bf16_t, SRWN_PRO_NONE)
  } else if (dtype == SRWN_F32) {
    if (pro == SRWN_PRO_GATE) SRWN_W2(float, SRWN_PRO_GATE)
    if (pro == SRWN_PRO_NONE) SRWN_W2(float, SRWN_PRO_NONE)
  } else {
    return set_error(SRWN_E_DTYPE, "wgrad256: dtype %d", dtype);
  }
#undef SRWN_W2
#undef SRWN_W2B
#undef SRWN_W2C
  return set_error(SRWN_E_UNSUPPORTED, "wgrad_wide: pro %d", pro);
}

extern "C" int srwn_wgrad_wide(const void* a, int64_t a_chunk_stride, int64_t a_row_stride, int32_t m_chunks,
                               int32_t chunk_width, const void* d, int64_t d_row_stride, int32_t d_width,
                               float* partials, float* bias_partials, int64_t rows, int32_t nslabs, int32_t pro,
                               int32_t dtype, void* stream) {
  return wgrad_wide_launch(&a, &d, &partials, &bias_partials, 1, a_chunk_stride, a_row_stride, m_chunks, chunk_width,
                           d_row_stride, d_width, rows, nslabs, pro, dtype, stream);
}

// two products of one shape as one launch (model.py:53,56: the head's two 1x1s -- out0 = a0^T . d0, out1 = a1^T . d1)
extern "C" int srwn_wgrad_wide_pair(const void* a0, const void* d0, float* partials0, float* bias_partials0,
                                    const void* a1, const void* d1, float* partials1, float* bias_partials1,
                                    int64_t a_chunk_stride, int64_t a_row_stride, int32_t m_chunks, int32_t chunk_width,
                                    int64_t d_row_stride, int32_t d_width, int64_t rows, int32_t nslabs, int32_t pro,
                                    int32_t dtype, void* stream) {
  const void* a[2] = {a0, a1};
  const void* d[2] = {d0, d1};
  float* p[2] = {partials0, partials1};
  float* b[2] = {bias_partials0, bias_partials1};
  return wgrad_wide_launch(a, d, p, b, 2, a_chunk_stride, a_row_stride, m_chunks, chunk_width, d_row_stride, d_width, rows,
                           nslabs, pro, dtype, stream);
}

// ------------------------------------------------------------------------------------------
// Per-layer weight gradients of the residual layer in ONE pass over the saved tensors, batched over
// layers (grid.y):   dWf_l[k] = x_l[t-(K-1-k)d]^T . df_l      (K = 2 taps; x_l optionally + cond)
//                    dbf_l    = colsum(df_l)
//                    dWr_l    = c_l^T . G_{l+1}               (c = z sigmoid(z); scaled by sqrt(.5) at reduce)
//                    dbr_l    = colsum(G_{l+1})
// x, z, df, G are each read once (the separate per-product kernels read x and df twice and z again).
// One workgroup = 4 waves over one slab of rows of one layer; 32-row chunks register-staged into LDS
// (one barrier per chunk), fragments by transposing LDS reads.  Wave w owns conv row tile w
// (rows 32w..32w+31 of [x(t-d) | x(t)]) x both df column tiles, and dWr tile (w>>1, w&1).
// ------------------------------------------------------------------------------------------
struct WgLArgs {
  const void* x; const void* z; const void* df; const void* g; int64_t layer_stride;   // [L][rows][64]
  const void* cond; int64_t cond_layer_stride; int cond_frames; int pool; int cond_stride;
  float* part_f; float* part_r; float* part_bf; float* part_br;   // [L][ns][2*64*64], [L][ns][64*64], [L][ns][64] x2
  int64_t rows; int Tlen; int rows_per_slab; int nslabs;
  int dil[64];
};

namespace {

// RC = residual channels (64, or 32 -- the width the reference's own scripts use): A tile [x(t-d) | x(t) | c], each
// RC wide; D tile [df | G].  RC = 64: wave w owns conv row tile w (rows 32w.. of [xd|xc]) x both df column tiles and
// dWr tile (w>>1, w&1).  RC = 32: waves 0/1 own the two conv row tiles, wave 2 the single dWr tile (the pass is
// HBM-bound; the idle wave costs nothing), and 64 rows are staged per step so every thread still moves one vector.
// NG = row groups per workgroup: each group of 4 waves streams its own contiguous share of the slab through its
// own pair of LDS tiles (all groups in step, one barrier per chunk) and the groups' accumulators are summed
// through LDS before the partial is written.  A launch covers a few layers x ~42 slabs, i.e. about one workgroup per
// CU: with a single group that leaves one 20-KB chunk in flight per CU and the pass is latency-bound (3.5 TB/s;
// the same kernel at 3 workgroups per CU streams 4.7 TB/s) -- more groups, not more slabs, because every slab is
// another fp32 partial for reduce_partials to read.
template <typename T, bool COND, int RC, int NG>
__global__ __launch_bounds__(256 * NG) void wgrad_layer_kernel(WgLArgs a) {
  constexpr int KR = (RC == 64) ? 32 : 64;      // rows staged per step
  constexpr int LA = 3 * RC + 16, LD = 2 * RC + 16;   // LDS row strides (elements, 16-byte multiples)
  constexpr int CT = RC / 32;                   // column tiles of df / G
  constexpr int VEC = 16 / sizeof(T);
  constexpr int VPC = RC / VEC;                 // 16-byte vectors per RC-channel row
  constexpr int NV = KR * VPC / 256;            // vectors per thread per tensor tile (1 bf16, 2 f32)
  static_assert(NV >= 1, "staging shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int grp = threadIdx.x >> 8;             // row group
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // within the group
  T* lds = reinterpret_cast<T*>(smem) + (size_t)grp * 2 * KR * (LA + LD);   // [NG][2 buffers][A tile KR x LA | D tile KR x LD]
  auto tileA = [&](int buf) { return lds + (size_t)buf * KR * (LA + LD); };
  auto tileD = [&](int buf) { return lds + (size_t)buf * KR * (LA + LD) + KR * LA; };

  const int slab = blockIdx.x, layer = blockIdx.y;
  const int d = a.dil[layer];
  const int64_t s_begin = (int64_t)slab * a.rows_per_slab;
  const int64_t s_end = (s_begin + a.rows_per_slab < a.rows) ? s_begin + a.rows_per_slab : a.rows;
  const int rows_per_group = (a.rows_per_slab / KR + NG - 1) / NG * KR;     // rows_per_slab is a multiple of KR
  const int nit_all = rows_per_group / KR;                                   // barrier count, the same for every group
  const int64_t r_begin = s_begin + (int64_t)grp * rows_per_group;
  const int64_t r_end = (r_begin + rows_per_group < s_end) ? r_begin + rows_per_group : s_end;
  const int nit = (r_end > r_begin) ? (int)((r_end - r_begin + KR - 1) / KR) : 0;
  const T* xb = reinterpret_cast<const T*>(a.x) + (int64_t)layer * a.layer_stride;
  const T* zb = reinterpret_cast<const T*>(a.z) + (int64_t)layer * a.layer_stride;
  const T* fb = reinterpret_cast<const T*>(a.df) + (int64_t)layer * a.layer_stride;
  const T* gb = reinterpret_cast<const T*>(a.g) + (int64_t)layer * a.layer_stride;
  const T* cb = COND ? reinterpret_cast<const T*>(a.cond) + (int64_t)layer * a.cond_layer_stride : nullptr;
  // tile ownership
  const bool has_conv = wave < 2 * CT;                       // conv row tile `wave` of [xd|xc]
  const bool has_res = (RC == 64) ? true : (wave == 2);      // dWr tile (rt, ct)
  const int res_rt = (RC == 64) ? (wave >> 1) : 0, res_ct = (RC == 64) ? (wave & 1) : 0;

  f32x16 accF[CT], accR;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
#pragma unroll
    for (int n = 0; n < CT; ++n) accF[n][q] = 0.0f;
    accR[q] = 0.0f;
  }
  // Column sums of the D tile [df | G] (the two bias gradients): wave w < 2 CT sums 32-column tile w with one MFMA per
  // k-step against a fragment of ONES (every row of the 32 x 32 result holds the sums).  Before, threads 0..2RC-1 -- two of a
  // group's four waves -- walked the tile's 32 rows per chunk (~100 instructions against the ~30 of the products) and the
  // other two waited at the chunk's barrier.
  f32x16 accB;
#pragma unroll
  for (int q = 0; q < 16; ++q) accB[q] = 0.0f;
  Frag<T> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.set(e, 1.0f);
  const bool has_bias = wave < 2 * CT;
  const int bias_col = wave < CT ? 32 * wave : RC + 32 * (wave - CT);

  f32x4 rxd[NV], rxc[NV], rz[NV], rf[NV], rg[NV];
  f32x4 rcd[COND ? NV : 1], rcc[COND ? NV : 1];
  bool okd[NV];
  auto gload = [&](int it) {
    const int64_t r0 = r_begin + (int64_t)it * KR;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = tid + v * 256;
      const int rr = idx / VPC, cv = (idx % VPC) * VEC;
      const int64_t row = r0 + rr;
      const bool okr = row < r_end;
      const int64_t rowc = okr ? row : (a.rows - 1);
      const int t = (int)(rowc % a.Tlen);
      okd[v] = okr && (t - d >= 0);
      const int64_t rowd = (t - d >= 0) ? rowc - d : rowc;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      rxd[v] = *reinterpret_cast<const f32x4*>(xb + rowd * RC + cv);
      rxc[v] = *reinterpret_cast<const f32x4*>(xb + rowc * RC + cv);
      rz[v] = *reinterpret_cast<const f32x4*>(zb + rowc * RC + cv);
      rf[v] = *reinterpret_cast<const f32x4*>(fb + rowc * RC + cv);
      rg[v] = *reinterpret_cast<const f32x4*>(gb + rowc * RC + cv);
      if (COND) {
        const int64_t bidx = rowc / a.Tlen;
        const int td = (t - d >= 0) ? t - d : t;
        rcd[v] = *reinterpret_cast<const f32x4*>(cb + (bidx * a.cond_frames + td / a.pool) * a.cond_stride + cv);
        rcc[v] = *reinterpret_cast<const f32x4*>(cb + (bidx * a.cond_frames + t / a.pool) * a.cond_stride + cv);
      }
      if (!okr) { rxc[v] = zero; rz[v] = zero; rf[v] = zero; rg[v] = zero; }
    }
  };
  auto addc = [&](f32x4 x, f32x4 c) {   // x + c in T precision
    if (sizeof(T) == 2) {
      bf16x8 xb_ = __builtin_bit_cast(bf16x8, x), cb_ = __builtin_bit_cast(bf16x8, c);
#pragma unroll
      for (int e = 0; e < 8; ++e) xb_[e] = (bf16_t)((float)xb_[e] + (float)cb_[e]);
      return __builtin_bit_cast(f32x4, xb_);
    } else {
      return x + c;
    }
  };
  auto lstore = [&](int buf) {
    T* ta = tileA(buf); T* td = tileD(buf);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = tid + v * 256;
      const int rr = idx / VPC, cv = (idx % VPC) * VEC;
      f32x4 xd = rxd[v], xc = rxc[v], zc = rz[v];
      if (COND) { xd = addc(xd, rcd[v]); xc = addc(xc, rcc[v]); }
      if (!okd[v]) xd = f32x4{0.f, 0.f, 0.f, 0.f};
      if (sizeof(T) == 2) {
        bf16x8 b = __builtin_bit_cast(bf16x8, zc);
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = (bf16_t)gate_of_z<T>((float)b[e]);
        zc = __builtin_bit_cast(f32x4, b);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) zc[e] = gate_of_z<T>(zc[e]);
      }
      *reinterpret_cast<f32x4*>(ta + rr * LA + cv) = xd;
      *reinterpret_cast<f32x4*>(ta + rr * LA + RC + cv) = xc;
      *reinterpret_cast<f32x4*>(ta + rr * LA + 2 * RC + cv) = zc;
      *reinterpret_cast<f32x4*>(td + rr * LD + cv) = rf[v];
      *reinterpret_cast<f32x4*>(td + rr * LD + RC + cv) = rg[v];
    }
  };

  if constexpr (sizeof(T) == 2) {
    // bf16: a ring of three register sets -- chunk it+4 is requested while chunk it is
    // multiplied, three steps before it moves to LDS.  With one set the request preceded its use by one chunk's
    // MFMAs (~0.2 us) against an HBM round trip of > 1 us, so every step waited for memory.  The loads are
    // unconditional (rows clamped into the tensor; rows outside the group's range are zeroed on their way to LDS),
    // which lets hipcc count them instead of draining.
    struct Ring { f32x4 xd[NV], xc[NV], z[NV], f[NV], g[NV], cd[COND ? NV : 1], cc[COND ? NV : 1]; };
    auto rload = [&](int it, Ring& r) {
      const int64_t r0 = r_begin + (int64_t)it * KR;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * 256;
        const int rr = idx / VPC, cv = (idx % VPC) * VEC;
        int64_t rowc = r0 + rr;
        rowc = rowc < a.rows ? rowc : a.rows - 1;
        const int t = (int)(rowc % a.Tlen);
        const int64_t rowd = (t - d >= 0) ? rowc - d : rowc;
        r.xd[v] = *reinterpret_cast<const f32x4*>(xb + rowd * RC + cv);
        r.xc[v] = *reinterpret_cast<const f32x4*>(xb + rowc * RC + cv);
        r.z[v] = *reinterpret_cast<const f32x4*>(zb + rowc * RC + cv);
        r.f[v] = *reinterpret_cast<const f32x4*>(fb + rowc * RC + cv);
        r.g[v] = *reinterpret_cast<const f32x4*>(gb + rowc * RC + cv);
        if (COND) {
          const int64_t bidx = rowc / a.Tlen;
          const int td = (t - d >= 0) ? t - d : t;
          r.cd[v] = *reinterpret_cast<const f32x4*>(cb + (bidx * a.cond_frames + td / a.pool) * a.cond_stride + cv);
          r.cc[v] = *reinterpret_cast<const f32x4*>(cb + (bidx * a.cond_frames + t / a.pool) * a.cond_stride + cv);
        }
      }
    };
    auto addc2 = [&](f32x4 x, f32x4 c) {   // x + c in T precision (the conditioned input the forward pass saw)
      bf16x8 xb_ = __builtin_bit_cast(bf16x8, x), cb_ = __builtin_bit_cast(bf16x8, c);
#pragma unroll
      for (int e = 0; e < 8; ++e) xb_[e] = (bf16_t)((float)xb_[e] + (float)cb_[e]);
      return __builtin_bit_cast(f32x4, xb_);
    };
    auto rstore = [&](int it, const Ring& r) {
      const int64_t r0 = r_begin + (int64_t)it * KR;
      T* ta = tileA(it & 1); T* td = tileD(it & 1);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * 256;
        const int rr = idx / VPC, cv = (idx % VPC) * VEC;
        const int64_t row = r0 + rr;
        const bool okr = row < r_end;
        const bool okdd = okr && ((int)(row % a.Tlen) - d >= 0);
        bf16x8 b = __builtin_bit_cast(bf16x8, okr ? r.z[v] : zero);
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = (bf16_t)gate_of_z<T>((float)b[e]);
        f32x4 xd = r.xd[v], xc = r.xc[v];
        if (COND) { xd = addc2(xd, r.cd[v]); xc = addc2(xc, r.cc[v]); }
        *reinterpret_cast<f32x4*>(ta + rr * LA + cv) = okdd ? xd : zero;
        *reinterpret_cast<f32x4*>(ta + rr * LA + RC + cv) = okr ? xc : zero;
        *reinterpret_cast<f32x4*>(ta + rr * LA + 2 * RC + cv) = __builtin_bit_cast(f32x4, b);
        *reinterpret_cast<f32x4*>(td + rr * LD + cv) = okr ? r.f[v] : zero;
        *reinterpret_cast<f32x4*>(td + rr * LD + RC + cv) = okr ? r.g[v] : zero;
      }
    };
    auto multiply = [&](int buf) {
      const T* ta = tileA(buf); const T* td = tileD(buf);
#pragma unroll
      for (int ks = 0; ks < KR / 16; ++ks) {
        if (has_bias) mma(accB, ones, Ld2<T>::load(td, LD, 16 * ks, bias_col, lane));
        if (has_conv) {
          const Frag<T> a_conv = Ld2<T>::load(ta, LA, 16 * ks, 32 * wave, lane);
#pragma unroll
          for (int n = 0; n < CT; ++n) mma(accF[n], a_conv, Ld2<T>::load(td, LD, 16 * ks, 32 * n, lane));
        }
        if (has_res) {
          const Frag<T> a_c = Ld2<T>::load(ta, LA, 16 * ks, 2 * RC + 32 * res_rt, lane);
          mma(accR, a_c, Ld2<T>::load(td, LD, 16 * ks, RC + 32 * res_ct, lane));
        }
      }
    };
    Ring q0, q1, q2;            // chunk c waits in set c % 3
    rload(0, q0);
    rload(1, q1);
    rload(2, q2);
    if (nit > 0) rstore(0, q0);
    rload(3, q0);
    __syncthreads();
    auto step = [&](int it, Ring& nxt) {   // every group runs nit_all steps (the barrier count is per workgroup)
      if (it < nit) multiply(it & 1);
      if (it + 1 < nit) rstore(it + 1, nxt);
      rload(it + 4, nxt);
      __syncthreads();
    };
    for (int it = 0; it < nit_all; it += 3) {
      step(it, q1);
      if (it + 1 >= nit_all) break;
      step(it + 1, q2);
      if (it + 2 >= nit_all) break;
      step(it + 2, q0);
    }
  } else {
  if (nit > 0) { gload(0); lstore(0); }
  __syncthreads();
  for (int it = 0; it < nit_all; ++it) {
    if (it >= nit) { __syncthreads(); continue; }
    const int buf = it & 1;
    if (it + 1 < nit) gload(it + 1);
    const T* ta = tileA(buf); const T* td = tileD(buf);
#pragma unroll
    for (int ks = 0; ks < KR / 16; ++ks) {
      if (has_bias) mma(accB, ones, Ld2<T>::load(td, LD, 16 * ks, bias_col, lane));
      if (has_conv) {
        const Frag<T> a_conv = Ld2<T>::load(ta, LA, 16 * ks, 32 * wave, lane);          // rows 32w.. of [xd|xc]
#pragma unroll
        for (int n = 0; n < CT; ++n) mma(accF[n], a_conv, Ld2<T>::load(td, LD, 16 * ks, 32 * n, lane));
      }
      if (has_res) {
        const Frag<T> a_c = Ld2<T>::load(ta, LA, 16 * ks, 2 * RC + 32 * res_rt, lane);  // c row tile
        mma(accR, a_c, Ld2<T>::load(td, LD, 16 * ks, RC + 32 * res_ct, lane));
      }
    }
    if (it + 1 < nit) lstore(buf ^ 1);
    __syncthreads();
  }
  }

  if (NG > 1) {
    // sum the groups: groups 1.. park their accumulators in LDS ([g-1][value][thread], the tiles are dead after the
    // loop's last barrier), group 0 adds them in group order (deterministic)
    constexpr int NVAL = 16 * (CT + 1) + 1;
    float* red = reinterpret_cast<float*>(smem);
    static_assert((size_t)(NG - 1) * NVAL * 256 * 4 <= (size_t)NG * 2 * KR * (LA + LD) * sizeof(T), "reduction scratch");
    if (grp > 0) {
      float* r = red + (size_t)(grp - 1) * NVAL * 256 + tid;
#pragma unroll
      for (int n = 0; n < CT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) r[(16 * n + q) * 256] = accF[n][q];
#pragma unroll
      for (int q = 0; q < 16; ++q) r[(16 * CT + q) * 256] = accR[q];
      r[(16 * (CT + 1)) * 256] = accB[0];
    }
    __syncthreads();
    if (grp > 0) return;
    for (int g = 1; g < NG; ++g) {
      const float* r = red + (size_t)(g - 1) * NVAL * 256 + tid;
#pragma unroll
      for (int n = 0; n < CT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) accF[n][q] += r[(16 * n + q) * 256];
#pragma unroll
      for (int q = 0; q < 16; ++q) accR[q] += r[(16 * CT + q) * 256];
      accB[0] += r[(16 * (CT + 1)) * 256];
    }
  }
  const int col = lane & 31, half = lane >> 5;
  const int64_t ls = (int64_t)layer * a.nslabs + slab;
  float* pf = a.part_f + ls * (2 * RC * RC);   // [k*RC + i][o]: conv row tile w covers rows 32w..32w+31
  float* pr = a.part_r + ls * (RC * RC);
  if (has_conv) {
#pragma unroll
    for (int n = 0; n < CT; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) pf[(int64_t)(32 * wave + crow(q, half)) * RC + 32 * n + col] = accF[n][q];
  }
  if (has_res) {
#pragma unroll
    for (int q = 0; q < 16; ++q) pr[(int64_t)(32 * res_rt + crow(q, half)) * RC + 32 * res_ct + col] = accR[q];
  }
  if (has_bias && lane < 32) {      // row 0 of the result: lanes 0..31, register 0
    if (wave < CT) a.part_bf[ls * RC + 32 * wave + lane] = accB[0];
    else a.part_br[ls * RC + 32 * (wave - CT) + lane] = accB[0];
  }
}

}  // namespace

extern "C" int srwn_wgrad_layers(const void* x, const void* z, const void* df, const void* g, int64_t layer_stride,
                                 const void* cond, int64_t cond_layer_stride, int32_t cond_frames,
                                 int32_t pool_stride, int32_t cond_row_stride, const int32_t* dilations,
                                 int32_t nlayers, float* part_f, float* part_r, float* part_bf, float* part_br,
                                 int64_t rows, int32_t T, int32_t nslabs, int32_t R, int32_t K, int32_t dtype,
                                 void* stream) {
  if (rows == 0 || nlayers == 0) return 0;
  if (!x || !z || !df || !g || !dilations || !part_f || !part_r || !part_bf || !part_br)
    return set_error(SRWN_E_NULL, "wgrad_layers: null pointer");
  if ((R != 64 && R != 32) || K != 2)
    return set_error(SRWN_E_UNSUPPORTED, "wgrad_layers: built for R=64 or 32, K=2 (got R=%d K=%d)", R, K);
  if (nlayers < 0 || nlayers > 64 || rows < 0 || T < 1 || rows % T || nslabs < 1)
    return set_error(SRWN_E_SHAPE, "wgrad_layers: nlayers=%d rows=%lld T=%d nslabs=%d", nlayers, (long long)rows, T, nslabs);
  if (cond && (pool_stride < 1 || cond_row_stride < R || (int64_t)cond_frames * pool_stride < T))
    return set_error(SRWN_E_SHAPE, "wgrad_layers: cond frames %d x pool %d < T %d", cond_frames, pool_stride, T);
  WgLArgs a;
  a.x = x; a.z = z; a.df = df; a.g = g; a.layer_stride = layer_stride;
  a.cond = cond; a.cond_layer_stride = cond_layer_stride; a.cond_frames = cond_frames;
  a.pool = pool_stride > 0 ? pool_stride : 1; a.cond_stride = cond_row_stride;
  a.part_f = part_f; a.part_r = part_r; a.part_bf = part_bf; a.part_br = part_br;
  a.rows = rows; a.Tlen = T; a.nslabs = nslabs;
  const int kr = (R == 64) ? 32 : 64;
  int64_t rps = (rows + nslabs - 1) / nslabs;
  rps = (rps + kr - 1) / kr * kr;
  a.rows_per_slab = (int)rps;
  for (int i = 0; i < 64; ++i) a.dil[i] = (i < nlayers) ? dilations[i] : 1;
  for (int i = 0; i < nlayers; ++i)
    if (a.dil[i] < 1) return set_error(SRWN_E_SHAPE, "wgrad_layers: dilation %d", a.dil[i]);
  hipStream_t st = (hipStream_t)stream;
  // row groups per workgroup: 2 in bf16 (90 KB of LDS; still fits beside a layer_bwd workgroup of the main stream),
  // 1 in fp32 (one group's two tiles are already 90 KB)
  const int ng = (dtype == SRWN_F32) ? 1 : 2;
  dim3 grid((unsigned)nslabs, (unsigned)nlayers), block(256 * ng);
#define SRWN_WLN(TT, C, NGV)                                                                                   \
  {                                                                                                            \
    auto kfn = (R == 64) ? wgrad_layer_kernel<TT, C, 64, NGV> : wgrad_layer_kernel<TT, C, 32, NGV>;            \
    const size_t sh = (size_t)NGV * 2 * kr * (5 * R + 32) * sizeof(TT);                                        \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    if (e != hipSuccess) return set_error((int)e, "wgrad_layers: LDS %zu: %s", sh, hipGetErrorString(e));      \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, a);                                                           \
    return check_launch("wgrad_layers");                                                                       \
  }
#define SRWN_WL(TT, C)                                                                                         \
  {                                                                                                            \
    if (ng == 2) SRWN_WLN(TT, C, 2)                                                                            \
    SRWN_WLN(TT, C, 1)                                                                                         \
  }
  if (dtype == SRWN_BF16) {
    if (cond) SRWN_WL(bf16_t, true) else SRWN_WL(bf16_t, false)
  } else if (dtype == SRWN_F32) {
    if (cond) SRWN_WLN(float, true, 1) else SRWN_WLN(float, false, 1)
  }
#undef SRWN_WLN
#undef SRWN_WL
  return set_error(SRWN_E_DTYPE, "wgrad_layers: dtype %d", dtype);
}

// ------------------------------------------------------------------------------------------
// Weight gradients of the encoder's ResidualDilationLayerNC chain (ops.py:48-58) in ONE pass over the saved tensors,
// batched over layers (grid.y), 128 channels, K = 2 taps at t and t+1 (SAME padding):
//   dW_l[k]  = r_l[t+k]^T . dpre_l[t]        (r_l = relu'd layer input; rows beyond the clip contribute 0)
//   db_l     = colsum(dpre_l)
//   dWr_l    = a_l^T . dh_l                   (a_l = relu(conv) of the layer; dh_l = gradient at the 1x1's output)
//   dbr_l    = colsum(dh_l)
// r, a, dpre, dh are each read once (three separate time-contraction launches read r and dpre twice).
// One workgroup = 8 waves over one slab of rows of one layer; 32-row chunks register-staged into LDS, fragments by
// transposing LDS reads.  Wave w owns conv row tile w (rows 32w.. of [r(t) | r(t+1)]) x the 4 dpre column tiles, and
// dWr tiles (w>>1, 2(w&1)) and (w>>1, 2(w&1)+1).
// ------------------------------------------------------------------------------------------
struct WgNcArgs {
  const void* r; const void* a; const void* dpre; const void* dh; int64_t layer_stride;   // [L][rows][128] each
  float* part_w; float* part_r; float* part_b; float* part_br;   // [L][ns][2*128*128], [L][ns][128*128], [L][ns][128] x2
  int64_t rows; int Tlen; int rows_per_slab; int nslabs;
};

namespace {

template <typename T>
__global__ __launch_bounds__(512) void wgrad_nc_kernel(WgNcArgs a) {
  constexpr int C = 128, KR = 32;
  constexpr int LA = 3 * C + 16, LD = 2 * C + 16;   // LDS row strides: A tile [r(t) | r(t+1) | a], D tile [dpre | dh]
  constexpr int VEC = 16 / sizeof(T);
  constexpr int VPC = C / VEC;                      // 16-byte vectors per 128-channel row
  constexpr int NV = KR * VPC / 512;                // vectors per thread per tensor tile (1 bf16, 2 f32)
  constexpr int NB = (sizeof(T) == 2) ? 2 : 1;      // LDS buffers (two fp32 tiles of this size exceed 160 KB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* lds = reinterpret_cast<T*>(smem);
  auto tileA = [&](int buf) { return lds + (size_t)buf * KR * (LA + LD); };
  auto tileD = [&](int buf) { return lds + (size_t)buf * KR * (LA + LD) + KR * LA; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slab = blockIdx.x, layer = blockIdx.y;
  const int64_t r_begin = (int64_t)slab * a.rows_per_slab;
  const int64_t r_end = (r_begin + a.rows_per_slab < a.rows) ? r_begin + a.rows_per_slab : a.rows;
  const int nit = (r_end > r_begin) ? (int)((r_end - r_begin + KR - 1) / KR) : 0;
  const T* rb = reinterpret_cast<const T*>(a.r) + (int64_t)layer * a.layer_stride;
  const T* ab = reinterpret_cast<const T*>(a.a) + (int64_t)layer * a.layer_stride;
  const T* pb = reinterpret_cast<const T*>(a.dpre) + (int64_t)layer * a.layer_stride;
  const T* hb = reinterpret_cast<const T*>(a.dh) + (int64_t)layer * a.layer_stride;

  f32x16 accW[4], accR[2];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
#pragma unroll
    for (int n = 0; n < 4; ++n) accW[n][q] = 0.0f;
    accR[0][q] = 0.0f; accR[1][q] = 0.0f;
  }
  // column sums of [dpre | dh] (the two bias gradients) as ones-MFMAs: wave w sums 32-column tile w (see wgrad_layer_kernel)
  f32x16 accB;
#pragma unroll
  for (int q = 0; q < 16; ++q) accB[q] = 0.0f;
  Frag<T> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.set(e, 1.0f);

  f32x4 r0v[NV], r1v[NV], av[NV], pv[NV], hv[NV];
  auto gload = [&](int it) {
    const int64_t base = r_begin + (int64_t)it * KR;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = tid + v * 512;
      const int rr = idx / VPC, cv = (idx % VPC) * VEC;
      const int64_t row = base + rr;
      const bool okr = row < r_end;
      const int64_t rowc = okr ? row : (a.rows - 1);
      const int t = (int)(rowc % a.Tlen);
      const bool ok1 = okr && (t + 1 < a.Tlen);           // tap t+1 stays inside the clip
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      r0v[v] = *reinterpret_cast<const f32x4*>(rb + rowc * C + cv);
      r1v[v] = *reinterpret_cast<const f32x4*>(rb + (ok1 ? rowc + 1 : rowc) * C + cv);
      av[v] = *reinterpret_cast<const f32x4*>(ab + rowc * C + cv);
      pv[v] = *reinterpret_cast<const f32x4*>(pb + rowc * C + cv);
      hv[v] = *reinterpret_cast<const f32x4*>(hb + rowc * C + cv);
      if (!ok1) r1v[v] = zero;
      if (!okr) { r0v[v] = zero; av[v] = zero; pv[v] = zero; hv[v] = zero; }
    }
  };
  auto lstore = [&](int buf) {
    T* ta = tileA(buf); T* td = tileD(buf);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int idx = tid + v * 512;
      const int rr = idx / VPC, cv = (idx % VPC) * VEC;
      *reinterpret_cast<f32x4*>(ta + rr * LA + cv) = r0v[v];
      *reinterpret_cast<f32x4*>(ta + rr * LA + C + cv) = r1v[v];
      *reinterpret_cast<f32x4*>(ta + rr * LA + 2 * C + cv) = av[v];
      *reinterpret_cast<f32x4*>(td + rr * LD + cv) = pv[v];
      *reinterpret_cast<f32x4*>(td + rr * LD + C + cv) = hv[v];
    }
  };

  auto multiply = [&](int buf) {
    const T* ta = tileA(buf); const T* td = tileD(buf);
#pragma unroll
    for (int ks = 0; ks < KR / 16; ++ks) {
      mma(accB, ones, Ld2<T>::load(td, LD, 16 * ks, 32 * wave, lane));                        // columns 32w.. of [dpre | dh]
      const Frag<T> a_conv = Ld2<T>::load(ta, LA, 16 * ks, 32 * wave, lane);                  // rows 32w.. of [r(t)|r(t+1)]
      const Frag<T> a_res = Ld2<T>::load(ta, LA, 16 * ks, 2 * C + 32 * (wave >> 1), lane);    // a row tile
#pragma unroll
      for (int n = 0; n < 4; ++n) mma(accW[n], a_conv, Ld2<T>::load(td, LD, 16 * ks, 32 * n, lane));
#pragma unroll
      for (int n = 0; n < 2; ++n)
        mma(accR[n], a_res, Ld2<T>::load(td, LD, 16 * ks, C + 32 * (2 * (wave & 1) + n), lane));
    }
  };
  if constexpr (sizeof(T) == 2) {
    // bf16: a ring of three register sets -- chunk it+4 is requested while chunk it is multiplied, three steps before it
    // moves to LDS (as wgrad_layer_kernel / wgrad256_kernel).  With one chunk (40 KB) in flight per workgroup and an HBM
    // round trip of ~2 us the pass streamed its 3.9 GB at 4.2 TB/s.  The loads are unconditional (rows clamped into the
    // tensors, zeroed on their way to LDS) so that hipcc counts them.
    struct Ring { f32x4 r0[NV], r1[NV], av[NV], pv[NV], hv[NV]; };
    auto rows_of = [&](int it, int v, int64_t& row, bool& okr, bool& ok1, int& rr, int& cv) {
      const int idx = tid + v * 512;
      rr = idx / VPC; cv = (idx % VPC) * VEC;
      row = r_begin + (int64_t)it * KR + rr;
      okr = row < r_end;
      const int64_t rowc = okr ? row : (a.rows - 1);
      ok1 = okr && ((int)(rowc % a.Tlen) + 1 < a.Tlen);       // tap t+1 stays inside the clip
      row = rowc;
    };
    auto rload = [&](int it, Ring& q) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        int64_t rowc; bool okr, ok1; int rr, cv;
        rows_of(it, v, rowc, okr, ok1, rr, cv);
        q.r0[v] = *reinterpret_cast<const f32x4*>(rb + rowc * C + cv);
        q.r1[v] = *reinterpret_cast<const f32x4*>(rb + (ok1 ? rowc + 1 : rowc) * C + cv);
        q.av[v] = *reinterpret_cast<const f32x4*>(ab + rowc * C + cv);
        q.pv[v] = *reinterpret_cast<const f32x4*>(pb + rowc * C + cv);
        q.hv[v] = *reinterpret_cast<const f32x4*>(hb + rowc * C + cv);
      }
    };
    auto rstore = [&](int it, const Ring& q) {
      T* ta = tileA(it & 1); T* td = tileD(it & 1);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        int64_t rowc; bool okr, ok1; int rr, cv;
        rows_of(it, v, rowc, okr, ok1, rr, cv);
        *reinterpret_cast<f32x4*>(ta + rr * LA + cv) = okr ? q.r0[v] : zero;
        *reinterpret_cast<f32x4*>(ta + rr * LA + C + cv) = ok1 ? q.r1[v] : zero;
        *reinterpret_cast<f32x4*>(ta + rr * LA + 2 * C + cv) = okr ? q.av[v] : zero;
        *reinterpret_cast<f32x4*>(td + rr * LD + cv) = okr ? q.pv[v] : zero;
        *reinterpret_cast<f32x4*>(td + rr * LD + C + cv) = okr ? q.hv[v] : zero;
      }
    };
    Ring q0, q1, q2;            // chunk c waits in set c % 3
    rload(0, q0);
    rload(1, q1);
    rload(2, q2);
    if (nit > 0) rstore(0, q0);
    rload(3, q0);
    __syncthreads();
    auto step = [&](int it, Ring& nxt) {
      if (it < nit) multiply(it & 1);
      if (it + 1 < nit) rstore(it + 1, nxt);
      rload(it + 4, nxt);
      __syncthreads();
    };
    for (int it = 0; it < nit; it += 3) {
      step(it, q1);
      step(it + 1, q2);
      step(it + 2, q0);
    }
  } else {
    if (nit > 0) { gload(0); lstore(0); }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
      if (it + 1 < nit) gload(it + 1);
      multiply(0);
      __syncthreads();   // every wave is done reading the only buffer
      if (it + 1 < nit) lstore(0);
      __syncthreads();
    }
  }


  const int col = lane & 31, half = lane >> 5;
  const int64_t ls = (int64_t)layer * a.nslabs + slab;
  float* pw = a.part_w + ls * (2 * C * C);   // [k*C + i][o]
  float* pr = a.part_r + ls * (C * C);
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int q = 0; q < 16; ++q) pw[(int64_t)(32 * wave + crow(q, half)) * C + 32 * n + col] = accW[n][q];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      pr[(int64_t)(32 * (wave >> 1) + crow(q, half)) * C + 32 * (2 * (wave & 1) + n) + col] = accR[n][q];
  if (lane < 32) {      // row 0 of the result: lanes 0..31, register 0; tiles 0..3 = dpre, 4..7 = dh
    if (wave < 4) a.part_b[ls * C + 32 * wave + lane] = accB[0];
    else a.part_br[ls * C + 32 * (wave - 4) + lane] = accB[0];
  }
}

}  // namespace

extern "C" int srwn_wgrad_nc_layers(const void* r, const void* a, const void* dpre, const void* dh, int64_t layer_stride,
                                    int32_t nlayers, float* part_w, float* part_r, float* part_b, float* part_br,
                                    int64_t rows, int32_t T, int32_t nslabs, int32_t C, int32_t K, int32_t dtype,
                                    void* stream) {
  if (rows == 0 || nlayers == 0) return 0;
  if (!r || !a || !dpre || !dh || !part_w || !part_r || !part_b || !part_br)
    return set_error(SRWN_E_NULL, "wgrad_nc_layers: null pointer");
  if (C != 128 || K != 2) return set_error(SRWN_E_UNSUPPORTED, "wgrad_nc_layers: built for 128 channels, K=2 (got C=%d K=%d)", C, K);
  if (nlayers < 0 || nlayers > 65535 || rows < 0 || T < 1 || rows % T || nslabs < 1)
    return set_error(SRWN_E_SHAPE, "wgrad_nc_layers: nlayers=%d rows=%lld T=%d nslabs=%d", nlayers, (long long)rows, T, nslabs);
  WgNcArgs g{r, a, dpre, dh, layer_stride, part_w, part_r, part_b, part_br, rows, T, 0, nslabs};
  int64_t rps = (rows + nslabs - 1) / nslabs;
  rps = (rps + 31) / 32 * 32;
  g.rows_per_slab = (int)rps;
  dim3 grid((unsigned)nslabs, (unsigned)nlayers), block(512);
  hipStream_t st = (hipStream_t)stream;
#define SRWN_WNC(TT)                                                                                           \
  {                                                                                                            \
    auto kfn = wgrad_nc_kernel<TT>;                                                                            \
    const size_t sh = (size_t)(sizeof(TT) == 2 ? 2 : 1) * 32 * (5 * 128 + 32) * sizeof(TT);                    \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    if (e != hipSuccess) return set_error((int)e, "wgrad_nc_layers: LDS %zu: %s", sh, hipGetErrorString(e));   \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, g);                                                           \
    return check_launch("wgrad_nc_layers");                                                                    \
  }
  if (dtype == SRWN_BF16) SRWN_WNC(bf16_t)
  if (dtype == SRWN_F32) SRWN_WNC(float)
#undef SRWN_WNC
  return set_error(SRWN_E_DTYPE, "wgrad_nc_layers: dtype %d", dtype);
}
