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

constexpr int kRows = 32;      // rows per staged chunk
constexpr int kW = 256;        // tile width (channels) of both operands
constexpr int kStride = 288;   // LDS row stride in elements: 576 B (bf16) puts 4 consecutive rows on disjoint banks

template <typename T, int PRO, int MB>
__global__ __launch_bounds__(512) void wgrad256_kernel(Wg2Args a) {
  constexpr int NTW = (MB == 4) ? 4 : 1;   // n-tiles per wave
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
  float bsum = 0.0f;
  const bool do_bias = (a.bias_partials != nullptr) && (mblk == 0) && (tid < kW);

  f32x4 ra[NVA], rd[NV];   // raw 16-byte vectors in flight (bit containers)
  auto gload = [&](int it) {
    const int64_t r0 = r_begin + (int64_t)it * kRows;
#pragma unroll
    for (int v = 0; v < NVA; ++v) {
      const int idx = tid + v * 512;
      const int rr = idx / VPRA, cv = (idx % VPRA) * VEC;
      const int64_t row = r0 + rr;
      const int chunk = mblk * MB + cv / 64;
      const bool ok = (idx < kRows * VPRA) && (row < r_end) && (chunk < a.m_chunks);
      ra[v] = ok ? *reinterpret_cast<const f32x4*>(abase + (int64_t)chunk * a.a_chunk_stride + row * a.a_row_stride + (cv & 63))
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

  if (nit > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    if (it + 1 < nit) gload(it + 1);
    const T* ta = tileA(buf); const T* td = tileD(buf);
    if (do_bias) {
#pragma unroll 8
      for (int rr = 0; rr < kRows; ++rr) bsum += (float)td[rr * kStride + tid];
    }
#pragma unroll
    for (int ks = 0; ks < kRows / 16; ++ks) {
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
    if (it + 1 < nit) lstore(buf ^ 1);
    __syncthreads();
  }

  const int col = lane & 31, half = lane >> 5;
  const int chunk = mblk * MB + wm;
  if (chunk < a.m_chunks) {
    float* pbase = a.partials + ((int64_t)slab * a.m_chunks * 64 + (int64_t)chunk * 64) * kW;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        const int o = (wn * NTW + n) * 32 + col;
#pragma unroll
        for (int q = 0; q < 16; ++q) pbase[(int64_t)(32 * m + crow(q, half)) * kW + o] = acc[m][n][q];
      }
  }
  if (do_bias) a.bias_partials[(int64_t)slab * kW + tid] = bsum;
}

}  // namespace

static int wg2_chunks_per_block(int m_chunks) { return m_chunks >= 8 ? 4 : 1; }

extern "C" int32_t srwn_wgrad256_slabs(int64_t rows, int32_t m_chunks) {
  const int mb = wg2_chunks_per_block(m_chunks);
  const int mblocks = (m_chunks + mb - 1) / mb;
  int64_t target = 256 / (mblocks > 0 ? mblocks : 1);
  if (target < 1) target = 1;
  int64_t maxs = (rows + 255) / 256;   // at least 256 rows per slab
  if (maxs < 1) maxs = 1;
  return (int32_t)(target < maxs ? target : maxs);
}

extern "C" int srwn_wgrad256(const void* a, int64_t a_chunk_stride, int64_t a_row_stride, int32_t m_chunks,
                             const void* d, int64_t d_row_stride, float* partials, float* bias_partials,
                             int64_t rows, int32_t nslabs, int32_t pro, int32_t dtype, void* stream) {
  if (rows == 0 || m_chunks == 0) return 0;
  if (!a || !d || !partials) return set_error(SRWN_E_NULL, "wgrad256: null pointer");
  if (rows < 0 || m_chunks < 0 || nslabs < 1 || d_row_stride < 256 || a_row_stride < 64)
    return set_error(SRWN_E_SHAPE, "wgrad256: rows=%lld m_chunks=%d nslabs=%d strides a=%lld d=%lld", (long long)rows,
                     m_chunks, nslabs, (long long)a_row_stride, (long long)d_row_stride);
  Wg2Args g{a, a_chunk_stride, a_row_stride, m_chunks, d, d_row_stride, partials, bias_partials, rows, 0, nslabs};
  int64_t rps = (rows + nslabs - 1) / nslabs;
  rps = (rps + kRows - 1) / kRows * kRows;
  g.rows_per_slab = (int)rps;
  const int mb = wg2_chunks_per_block(m_chunks);
  dim3 grid((unsigned)nslabs, (unsigned)((m_chunks + mb - 1) / mb)), block(512);
  hipStream_t st = (hipStream_t)stream;
#define SRWN_W2(TT, P)                                                                                        \
  if (mb == 4) SRWN_W2B(TT, P, 4) else SRWN_W2B(TT, P, 1)
#define SRWN_W2B(TT, P, MBV)                                                                                  \
  {                                                                                                           \
    auto kfn = wgrad256_kernel<TT, P, MBV>;                                                                   \
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
  return set_error(SRWN_E_UNSUPPORTED, "wgrad256: pro %d", pro);
}
