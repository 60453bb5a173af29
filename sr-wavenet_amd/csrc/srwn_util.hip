// Utility kernels + C-ABI glue: error reporting, mu-law, weight packing, generic causal conv,
// input-conv weight gradient, loss reduction.  gfx950 only.
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
namespace srwn {
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error((int)e, "%s: %s", what, hipGetErrorString(e));
  return 0;
}
}  // namespace srwn

extern "C" int srwn_version(void) { return 100; }
extern "C" const char* srwn_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------
// mu-law (ops.py:82-104)
// ------------------------------------------------------------------------------------------
__global__ void mu_law_encode_kernel(const float* __restrict__ audio, int32_t* __restrict__ codes, int64_t n,
                                     float mu, float log1p_mu) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = audio[i];
  float safe = fminf(fabsf(a), 1.0f);                              // ops.py:88
  float lm = (float)log1p((double)__fmul_rn(mu, safe));             // correctly rounded f32 log1p
  float magnitude = __fdiv_rn(lm, log1p_mu);                        // ops.py:89
  float sgn = (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f);      // tf.sign
  float signal = __fmul_rn(sgn, magnitude);                         // ops.py:90
  float q = __fadd_rn(__fmul_rn(__fdiv_rn(__fadd_rn(signal, 1.0f), 2.0f), mu), 0.5f);  // ops.py:92
  codes[i] = (int32_t)q;                                            // tf.to_int32 truncates
}

__global__ void mu_law_decode_kernel(const int32_t* __restrict__ codes, float* __restrict__ audio, int64_t n,
                                     float mu, float inv_mu, double base) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o = (float)codes[i];
  float signal = __fadd_rn(__fmul_rn(2.0f, __fdiv_rn(o, mu)), -1.0f);  // ops.py:101
  float p = (float)pow(base, (double)fabsf(signal));                    // (1+mu)**|signal|
  float magnitude = __fmul_rn(inv_mu, __fadd_rn(p, -1.0f));             // ops.py:103
  float sgn = (signal > 0.0f) ? 1.0f : ((signal < 0.0f) ? -1.0f : 0.0f);
  audio[i] = __fmul_rn(sgn, magnitude);
}

extern "C" int srwn_mu_law_encode(const float* audio, int32_t* codes, int64_t n, int32_t Q, void* stream) {
  if (n == 0) return 0;
  if (!audio || !codes) return set_error(SRWN_E_NULL, "mu_law_encode: null pointer");
  if (n < 0 || Q < 2) return set_error(SRWN_E_SHAPE, "mu_law_encode: n=%lld Q=%d", (long long)n, Q);
  float mu = (float)(Q - 1);
  float l1p = (float)log1p((double)mu);
  hipLaunchKernelGGL(mu_law_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     audio, codes, n, mu, l1p);
  return check_launch("mu_law_encode");
}

extern "C" int srwn_mu_law_decode(const int32_t* codes, float* audio, int64_t n, int32_t Q, void* stream) {
  if (n == 0) return 0;
  if (!audio || !codes) return set_error(SRWN_E_NULL, "mu_law_decode: null pointer");
  if (n < 0 || Q < 2) return set_error(SRWN_E_SHAPE, "mu_law_decode: n=%lld Q=%d", (long long)n, Q);
  int mu = Q - 1;
  hipLaunchKernelGGL(mu_law_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     codes, audio, n, (float)mu, (float)(1.0 / (double)mu), (double)(1 + mu));
  return check_launch("mu_law_decode");
}

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
__global__ void pack_a_index_kernel(int32_t* __restrict__ dst, int32_t src_offset, int32_t rows_valid,
                                    int32_t k_valid, int32_t row_stride, int32_t k_stride, int32_t mt_count,
                                    int32_t ks_total, int32_t ks_offset, int32_t ks_count, int32_t perm_from_ks) {
  // one thread per (mt, ks_local, lane, j)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)mt_count * ks_count * 64 * 8;
  if (i >= total) return;
  int j = (int)(i & 7);
  int lane = (int)((i >> 3) & 63);
  int ksl = (int)((i >> 9) % ks_count);
  int mt = (int)((i >> 9) / ks_count);
  int r = lane & 31, h = lane >> 5;
  int kin = (ksl >= perm_from_ks) ? (8 * (j >> 2) + 4 * h + (j & 3)) : (8 * h + j);
  int k = 16 * ksl + kin;
  int row = 32 * mt + r;
  int32_t v = (row < rows_valid && k < k_valid) ? (src_offset + row * row_stride + k * k_stride) : -1;
  int64_t o = ((((int64_t)mt * ks_total + (ks_offset + ksl)) * 64) + lane) * 8 + j;
  dst[o] = v;
}

extern "C" int srwn_pack_a_index(int32_t* dst_idx, int32_t src_offset, int32_t rows_valid, int32_t k_valid,
                                 int32_t row_stride, int32_t k_stride, int32_t mt_count, int32_t ks_total,
                                 int32_t ks_offset, int32_t ks_count, int32_t perm_from_ks, void* stream) {
  if (!dst_idx) return set_error(SRWN_E_NULL, "pack_a_index: null dst");
  if (mt_count <= 0 || ks_count <= 0 || ks_offset < 0 || ks_offset + ks_count > ks_total)
    return set_error(SRWN_E_SHAPE, "pack_a_index: mt=%d ks=[%d,+%d) of %d", mt_count, ks_offset, ks_count, ks_total);
  int64_t total = (int64_t)mt_count * ks_count * 512;
  hipLaunchKernelGGL(pack_a_index_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dst_idx, src_offset, rows_valid, k_valid, row_stride, k_stride, mt_count, ks_total, ks_offset,
                     ks_count, perm_from_ks);
  return check_launch("pack_a_index");
}

// eight consecutive outputs per thread: two 16-byte index reads, eight gathered parameters (L2 hits: the parameter buffer
// is 4 MB), one 16-byte (bf16) or two (fp32) stores -- one element per thread spent 14 us on 3 M two-byte stores
template <typename T>
__global__ __launch_bounds__(256) void pack_gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          T* __restrict__ dst, int64_t n, unsigned gather_blocks,
                                                          const float* __restrict__ sum_src, int sum_rows, int sum_cols,
                                                          float* __restrict__ sum_out) {
  if (blockIdx.x >= gather_blocks) {      // the blocks behind the gather: column sums of [sum_rows, sum_cols] (rows in order, fp64)
    const int c = (int)(blockIdx.x - gather_blocks) * 256 + (int)threadIdx.x;
    if (c < sum_cols) {
      double s = 0.0;
      for (int l = 0; l < sum_rows; ++l) s += (double)sum_src[(size_t)l * sum_cols + c];
      sum_out[c] = (float)s;
    }
    return;
  }
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i0 >= n) return;
  if (i0 + 8 <= n) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 a = *reinterpret_cast<const i32x4*>(idx + i0), b = *reinterpret_cast<const i32x4*>(idx + i0 + 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j] >= 0 ? src[a[j]] : 0.0f; v[4 + j] = b[j] >= 0 ? src[b[j]] : 0.0f; }
    if constexpr (sizeof(T) == 2) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x8*>(dst + i0) = o;
    } else {
      *reinterpret_cast<f32x4*>(dst + i0) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst + i0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  } else {
    for (int64_t i = i0; i < n; ++i) {
      const int32_t s = idx[i];
      dst[i] = (T)(s >= 0 ? src[s] : 0.0f);
    }
  }
}

extern "C" int srwn_pack_gather_rowsum(const float* src, const int32_t* idx, void* dst, int64_t n, int32_t dtype,
                                       const float* sum_src, int32_t sum_rows, int32_t sum_cols, float* sum_out,
                                       void* stream) {
  const bool sum = sum_rows > 0 && sum_cols > 0;
  if (n == 0 && !sum) return 0;
  if (n < 0 || sum_rows < 0 || sum_cols < 0) return set_error(SRWN_E_SHAPE, "pack_gather: n=%lld sum %d x %d", (long long)n, sum_rows, sum_cols);
  if (n > 0 && (!src || !idx || !dst)) return set_error(SRWN_E_NULL, "pack_gather: null pointer");
  if (sum && (!sum_src || !sum_out)) return set_error(SRWN_E_NULL, "pack_gather_rowsum: null pointer");
  if (((uintptr_t)idx | (uintptr_t)dst) & 15) return set_error(SRWN_E_SHAPE, "pack_gather: idx and dst must be 16-byte aligned");
  const unsigned gb = (unsigned)((n + 2047) / 2048), sb = sum ? (unsigned)((sum_cols + 255) / 256) : 0u;
  dim3 grid(gb + sb), block(256);
  if (dtype == SRWN_F32)
    hipLaunchKernelGGL(pack_gather_kernel<float>, grid, block, 0, (hipStream_t)stream, src, idx, (float*)dst, n, gb, sum_src,
                       sum ? sum_rows : 0, sum ? sum_cols : 0, sum_out);
  else if (dtype == SRWN_BF16)
    hipLaunchKernelGGL(pack_gather_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, src, idx, (bf16_t*)dst, n, gb, sum_src,
                       sum ? sum_rows : 0, sum ? sum_cols : 0, sum_out);
  else
    return set_error(SRWN_E_DTYPE, "pack_gather: dtype %d", dtype);
  return check_launch("pack_gather");
}

extern "C" int srwn_pack_gather(const float* src, const int32_t* idx, void* dst, int64_t n, int32_t dtype,
                                void* stream) {
  return srwn_pack_gather_rowsum(src, idx, dst, n, dtype, nullptr, 0, 0, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// generic causal conv (ops.py:6-20), plain VALU: thread = (b, t, 4 output channels)
// ------------------------------------------------------------------------------------------
template <typename TOut>
__global__ void causal_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                   const float* __restrict__ bias, TOut* __restrict__ y, int B, int T, int Cin,
                                   int Cout, int K, int dilation, int shift) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int cq = (Cout + 3) / 4;
  int64_t total = (int64_t)B * T * cq;
  if (i >= total) return;
  int o0 = (int)(i % cq) * 4;
  int64_t bt = i / cq;
  int t = (int)(bt % T);
  int b = (int)(bt / T);
  float acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (bias && o0 + q < Cout) ? bias[o0 + q] : 0.0f;
  for (int k = 0; k < K; ++k) {
    int tk = t - (K - 1 - k) * dilation - shift;
    if (tk < 0 || tk >= T) continue;
    const float* xr = x + ((int64_t)b * T + tk) * Cin;
    const float* wk = w + (int64_t)k * Cin * Cout;
    for (int ci = 0; ci < Cin; ++ci) {
      float xv = xr[ci];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (o0 + q < Cout) acc[q] = fmaf(xv, wk[(int64_t)ci * Cout + o0 + q], acc[q]);
    }
  }
  TOut* yr = y + ((int64_t)b * T + t) * Cout + o0;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (o0 + q < Cout) yr[q] = (TOut)acc[q];
}

// Cin = 1 (the input conv of every stack, model.py:40,173,424): 8 output channels per thread, one 16-byte store (bf16).
// KT > 0: filter width known at compile time -- the thread keeps its 8 x (K + 1) weights and biases in registers and
// walks kCin1Rows rows (a block covers 32*kCin1Rows consecutive rows; at one row per thread the 24 weight loads per
// 16-byte store made the kernel instruction-bound: 15 us for 16 MB).
constexpr int kCin1Rows = 8;
template <typename TOut, int KT>
__global__ __launch_bounds__(256) void causal_conv_cin1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, TOut* __restrict__ y,
                                                               int B, int T, int Cout, int Krt, int dilation, int shift) {
  const int K = KT > 0 ? KT : Krt;
  const int lpr = Cout / 8;
  const int rpb = 256 / lpr;                               // rows a block covers per pass
  const int sub = threadIdx.x % lpr, rloc = threadIdx.x / lpr;
  const int64_t rows = (int64_t)B * T;
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bv[j] = bias ? bias[8 * sub + j] : 0.0f;
  constexpr int KR = KT > 0 ? KT : 1;
  float wv[KR][8];
  if (KT > 0) {
#pragma unroll
    for (int k = 0; k < KR; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[k][j] = w[k * Cout + 8 * sub + j];
  }
  const int64_t row0 = (int64_t)blockIdx.x * rpb * kCin1Rows + rloc;
#pragma unroll 2
  for (int it = 0; it < kCin1Rows; ++it) {
    const int64_t row = row0 + (int64_t)it * rpb;
    if (row >= rows) return;
    const int t = (int)(row % T);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bv[j];
    if (KT > 0) {
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const int tk = t - (K - 1 - k) * dilation - shift;
        const float xv = (tk >= 0 && tk < T) ? x[row - t + tk] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(xv, wv[k][j], v[j]);
      }
    } else {
      for (int k = 0; k < K; ++k) {
        const int tk = t - (K - 1 - k) * dilation - shift;
        if (tk < 0 || tk >= T) continue;
        const float xv = x[row - t + tk];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(xv, w[k * Cout + 8 * sub + j], v[j]);
      }
    }
    TOut* yr = y + row * Cout + 8 * sub;
    if (sizeof(TOut) == 2) {
      bf16x8 r;
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x8*>(yr) = r;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) yr[j] = (TOut)v[j];
    }
  }
}

extern "C" int srwn_causal_conv1d_fwd(const float* x, const float* w, const float* bias, void* y, int32_t B,
                                      int32_t T, int32_t Cin, int32_t Cout, int32_t K, int32_t dilation,
                                      int32_t shift, int32_t dtype_out, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!x || !w || !y) return set_error(SRWN_E_NULL, "causal_conv1d_fwd: null pointer");
  if (B < 0 || T < 0 || Cin < 1 || Cout < 1 || K < 1 || dilation < 1 || shift < -(1 << 30) || shift > (1 << 30))
    return set_error(SRWN_E_SHAPE, "causal_conv1d_fwd: B=%d T=%d Cin=%d Cout=%d K=%d d=%d shift=%d", B, T, Cin, Cout,
                     K, dilation, shift);
  if (Cin == 1 && Cout % 8 == 0 && 256 % (Cout / 8) == 0 && (dtype_out == SRWN_F32 || dtype_out == SRWN_BF16)) {
    const int64_t rows_per_block = (int64_t)(256 / (Cout / 8)) * kCin1Rows;
    dim3 g1((unsigned)(((int64_t)B * T + rows_per_block - 1) / rows_per_block)), b1(256);
#define SRWN_C1(TT, KT_)                                                                                      \
    hipLaunchKernelGGL((causal_conv_cin1_kernel<TT, KT_>), g1, b1, 0, (hipStream_t)stream, x, w, bias, (TT*)y, B, T, Cout, \
                       K, dilation, shift)
    if (dtype_out == SRWN_F32) { if (K == 2) SRWN_C1(float, 2); else SRWN_C1(float, 0); }
    else { if (K == 2) SRWN_C1(bf16_t, 2); else SRWN_C1(bf16_t, 0); }
#undef SRWN_C1
    return check_launch("causal_conv1d_fwd");
  }
  int64_t total = (int64_t)B * T * ((Cout + 3) / 4);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype_out == SRWN_F32)
    hipLaunchKernelGGL(causal_conv_kernel<float>, grid, block, 0, (hipStream_t)stream, x, w, bias, (float*)y, B, T,
                       Cin, Cout, K, dilation, shift);
  else if (dtype_out == SRWN_BF16)
    hipLaunchKernelGGL(causal_conv_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, x, w, bias, (bf16_t*)y, B,
                       T, Cin, Cout, K, dilation, shift);
  else
    return set_error(SRWN_E_DTYPE, "causal_conv1d_fwd: dtype %d", dtype_out);
  return check_launch("causal_conv1d_fwd");
}

// ------------------------------------------------------------------------------------------
// input conv (Cin = 1) weight/bias gradient: two-stage deterministic reduction.
// stage 1: block p sums rows [p*ROWS, (p+1)*ROWS) of the flattened [B*T] axis -> partials[p][(K+1)*R]
// stage 2: fixed-order sum over p.
// ------------------------------------------------------------------------------------------
constexpr int kIcRows = 256;

// KT > 0: the filter width as a compile-time constant (accumulators stay in registers; with a run-time K the
// acc[9][8] array lived in scratch memory: 68 MB of scratch traffic per launch at the benchmark size); KT = 0: any K <= 8
template <typename T, int KT>
__global__ __launch_bounds__(256) void init_conv_wgrad_stage1(const float* __restrict__ audio,
                                                              const T* __restrict__ g, float* __restrict__ partials,
                                                              int B, int Tlen, int R, int Krt, int shift) {
  const int K = KT > 0 ? KT : Krt;
  constexpr int KA = KT > 0 ? KT + 1 : 9;
  // thread = (8-channel group cg, row group rg): 16-byte loads of g, K+1 running sums per channel
  extern __shared__ float red[];  // [nrg][(K+1)*R]
  const int ncg = R / 8, nrg = 256 / ncg;
  const int cg = threadIdx.x % ncg, rg = threadIdx.x / ncg;
  const int64_t rows = (int64_t)B * Tlen;
  const int64_t r0 = (int64_t)blockIdx.x * kIcRows;
  float acc[KA][8];  // K <= 8
#pragma unroll
  for (int k = 0; k < KA; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[k][e] = 0.0f;
  const int t_block = (int)(r0 % Tlen);   // one 64-bit modulo per block; rows inside use 32-bit arithmetic
  for (int rr = rg; rr < kIcRows; rr += nrg) {
    const int64_t row = r0 + rr;
    if (row >= rows) break;
    const int t = (t_block + rr) % Tlen;
    float gv[8];
    const T* gp = g + row * R + cg * 8;
    if (sizeof(T) == 2) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(gp);
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = (float)v[e];
    } else {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(gp), v1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gv[e] = v0[e]; gv[4 + e] = v1[e]; }
    }
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      if (KT == 0 && k > K) break;
      float xv = 1.0f;                       // k == K: bias gradient (sum of g)
      if (k < K) {
        const int tk = t - (K - 1 - k) - shift;   // shift < 0: taps ahead of t (the non-causal encoder input conv)
        xv = (tk >= 0 && tk < Tlen) ? audio[row - t + tk] : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[k][e] = fmaf(xv, gv[e], acc[k][e]);
    }
  }
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    if (KT == 0 && k > K) break;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(rg * (K + 1) + k) * R + cg * 8 + e] = acc[k][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (K + 1) * R; i += 256) {
    float s = 0.0f;
    for (int q = 0; q < nrg; ++q) s += red[q * (K + 1) * R + i];
    partials[(int64_t)blockIdx.x * (K + 1) * R + i] = s;
  }
}

// one wave per output element: lanes stride over the partials (fixed order), then a shuffle tree
__global__ __launch_bounds__(256) void init_conv_wgrad_stage2(const float* __restrict__ partials, int64_t nparts,
                                                              float* __restrict__ gw, float* __restrict__ gb, int R,
                                                              int K) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= (K + 1) * R) return;
  float s = 0.0f;
  for (int64_t p = lane; p < nparts; p += 64) s += partials[p * (K + 1) * R + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) {
    if (i < K * R) gw[i] = s; else gb[i - K * R] = s;
  }
}

extern "C" int64_t srwn_init_conv_wgrad_partials(int32_t B, int32_t T, int32_t R, int32_t K) {
  int64_t rows = (int64_t)B * T;
  return ((rows + kIcRows - 1) / kIcRows) * (int64_t)(K + 1) * R;
}

extern "C" int srwn_init_conv_wgrad(const float* audio, const void* g, float* partials, float* gw, float* gb,
                                    int32_t B, int32_t T, int32_t R, int32_t K, int32_t shift, int32_t dtype,
                                    void* stream) {
  if (!audio || !g || !partials || ((gw == nullptr) != (gb == nullptr))) return set_error(SRWN_E_NULL, "init_conv_wgrad: null pointer");
  if (B < 1 || T < 1 || K < 1 || K > 8 || (R != 32 && R != 64 && R != 128))
    return set_error(SRWN_E_SHAPE, "init_conv_wgrad: B=%d T=%d R=%d K=%d", B, T, R, K);
  int64_t rows = (int64_t)B * T;
  int64_t nparts = (rows + kIcRows - 1) / kIcRows;
  size_t sh = (size_t)(256 / (R / 8)) * (K + 1) * R * sizeof(float);
  if (sh > 65536) return set_error(SRWN_E_UNSUPPORTED, "init_conv_wgrad: K=%d too large for the reduction buffer", K);
#define SRWN_IC(TT, KT_)                                                                                       \
  hipLaunchKernelGGL((init_conv_wgrad_stage1<TT, KT_>), dim3((unsigned)nparts), dim3(256), sh, (hipStream_t)stream, \
                     audio, (const TT*)g, partials, B, T, R, K, shift)
  if (dtype == SRWN_F32) {
    if (K == 2) SRWN_IC(float, 2); else SRWN_IC(float, 0);
  } else if (dtype == SRWN_BF16) {
    if (K == 2) SRWN_IC(bf16_t, 2); else SRWN_IC(bf16_t, 0);
  } else {
    return set_error(SRWN_E_DTYPE, "init_conv_wgrad: dtype %d", dtype);
  }
#undef SRWN_IC
  int rc = check_launch("init_conv_wgrad_stage1");
  if (rc) return rc;
  if (!gw) return 0;      // partials only: the caller sums them (e.g. as one more job of srwn_reduce_partials_multi)
  int n = (K + 1) * R;
  hipLaunchKernelGGL(init_conv_wgrad_stage2, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, partials, nparts,
                     gw, gb, R, K);
  return check_launch("init_conv_wgrad_stage2");
}

// ------------------------------------------------------------------------------------------
// loss partial reduction (fixed order, f64 accumulate)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_loss_kernel(const float* __restrict__ p, int64_t n, float scale,
                                                          float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)p[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)scale);
}

extern "C" int srwn_reduce_loss(const float* loss_partials, int64_t n, float scale, float* loss_out, void* stream) {
  if (!loss_partials || !loss_out) return set_error(SRWN_E_NULL, "reduce_loss: null pointer");
  hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_partials, n, scale,
                     loss_out);
  return check_launch("reduce_loss");
}
