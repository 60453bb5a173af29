// Auto-encoder pieces outside the MFMA GEMMs (model.py:136-156, ops.py:48-58, 178-201): the 1-channel non-causal
// input convolution of the encoder, small fp32 products on the [B*frames] axis, the mixture-of-logistics
// sampler.  gfx950 (MI355X) only.
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

namespace {

template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (bf16_t)v[j];
  *reinterpret_cast<bf16x8*>(p) = r;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

// ResidualDilationLayerNC on the raw clip (model.py:141-142; ops.py:49-52): x = relu(inputs) [B,T,1];
//   a[b,t,c] = relu( bias[c] + sum_k w[k][c] * x[b, t+k] ),  x beyond the clip = 0   (K taps, SAME padding for K=2)
// C/8 lanes per row, 8 channels each: the output write is the only traffic.
template <typename T>
__global__ __launch_bounds__(256) void nc_input_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ a, int B,
                                                       int Tlen, int C, int K) {
  const int lpr = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = idx / lpr;
  const int sub = (int)(idx % lpr);
  if (row >= (int64_t)B * Tlen) return;
  const int t = (int)(row % Tlen);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = bias[8 * sub + j];
  for (int k = 0; k < K; ++k) {
    if (t + k >= Tlen) break;
    const float xv = fmaxf(x[row + k], 0.0f);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaf(xv, w[k * C + 8 * sub + j], v[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.0f);
  store8<T>(a + row * C + 8 * sub, v);
}

// small fp32-accumulating product on few rows (the [B*frames] axis of the auto-encoder):
//   C[m][n] = (accumulate ? C : 0) + bias[n] + sum_k A(m,k) * B(k,n)
//   A(m,k) = a[(k/a_chunk)*a_chunk_stride + m*lda + k%a_chunk]   (dtype TA: bf16 or fp32)
//   B(k,n) = b[(k/b_chunk)*b_chunk_stride + (k%b_chunk)*ldb_k + n*ldb_n]   (fp32)
// 16 lanes share an output and stride over k (coalesced A reads), then a fixed xor-butterfly.
template <typename TA, typename TC>
__global__ __launch_bounds__(256) void small_gemm_kernel(const TA* __restrict__ a, int64_t lda, int a_chunk,
                                                         int64_t a_chunk_stride, const float* __restrict__ b,
                                                         int64_t ldb_k, int64_t ldb_n, int b_chunk,
                                                         int64_t b_chunk_stride, const float* __restrict__ bias,
                                                         TC* __restrict__ c, int64_t ldc, int M, int N, int Kd,
                                                         int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const bool live = i < (int64_t)M * N;
  const int n = live ? (int)(i % N) : 0, m = live ? (int)(i / N) : 0;
  float s = 0.0f;
  if (live) {
    for (int k = sub; k < Kd; k += 16) {
      const float av = (float)a[(int64_t)(k / a_chunk) * a_chunk_stride + (int64_t)m * lda + (k % a_chunk)];
      const float bv = b[(int64_t)(k / b_chunk) * b_chunk_stride + (int64_t)(k % b_chunk) * ldb_k + (int64_t)n * ldb_n];
      s = fmaf(av, bv, s);
    }
  }
#pragma unroll
  for (int w = 8; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
  if (live && sub == 0) {
    if (bias) s += bias[n];
    TC* d = c + (int64_t)m * ldc + n;
    *d = (TC)(accumulate ? (float)*d + s : s);
  }
}

// C[k][n] = scale * sum_m A[m][k] * D[m][n]  (+ bias_out[n] = scale * sum_m D[m][n]): weight gradients of the small
// products above, all fp32, M in the thousands.  16 lanes share an output and stride over m; fixed butterfly.
__global__ __launch_bounds__(256) void small_wgrad_kernel(const float* __restrict__ a, int64_t lda,
                                                          const float* __restrict__ d, int64_t ldd,
                                                          float* __restrict__ c, float* __restrict__ bias_out, int M,
                                                          int Kd, int N, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const int64_t total = (int64_t)(Kd + 1) * N;
  const bool live = i < total;
  const int n = live ? (int)(i % N) : 0, k = live ? (int)(i / N) : 0;
  double s = 0.0;
  if (live) {
    if (k < Kd) {
      for (int m = sub; m < M; m += 16) s += (double)a[(int64_t)m * lda + k] * (double)d[(int64_t)m * ldd + n];
    } else {
      for (int m = sub; m < M; m += 16) s += (double)d[(int64_t)m * ldd + n];
    }
  }
#pragma unroll
  for (int w = 8; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
  if (live && sub == 0) {
    if (k < Kd) c[(int64_t)k * N + n] = (float)(s * (double)scale);
    else if (bias_out) bias_out[n] = (float)(s * (double)scale);
  }
}

// sample_from_discretized_mix_logistic (ops.py:178-201) with the uniform draws supplied by the caller:
//   sel = argmax_m( logit_probs_m - log(-log(u1_m)) );  x = mean_sel + exp(max(log_scale_sel, -7)) * (log u2 - log(1-u2))
//   clipped to [-1, 1].  (the reference draws u1, u2 ~ U(1e-5, 1-1e-5); "coeffs" are computed there and never used)
__global__ __launch_bounds__(256) void mol_sample_kernel(const float* __restrict__ logits, int64_t ldl, int M,
                                                         const float* __restrict__ u1, const float* __restrict__ u2,
                                                         float* __restrict__ out, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float* l = logits + row * ldl;
  int sel = 0;
  float best = -INFINITY;
  for (int m = 0; m < M; ++m) {
    const float v = l[m] - logf(-logf(u1[row * M + m]));
    if (v > best) { best = v; sel = m; }   // first maximum wins (tf.argmax)
  }
  const float mean = l[M + sel];
  const float ls = fmaxf(l[2 * M + sel], -7.0f);
  const float u = u2[row];
  const float x = mean + expf(ls) * (logf(u) - logf(1.0f - u));
  out[row] = fminf(fmaxf(x, -1.0f), 1.0f);
}

}  // namespace

extern "C" int srwn_nc_input_fwd(const float* x, const float* w, const float* bias, void* a, int32_t B, int32_t T,
                                 int32_t C, int32_t K, int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!x || !w || !bias || !a) return set_error(SRWN_E_NULL, "nc_input_fwd: null pointer");
  if (B < 0 || T < 0 || C < 8 || C % 8 || K < 1) return set_error(SRWN_E_SHAPE, "nc_input_fwd: B=%d T=%d C=%d K=%d", B, T, C, K);
  const int64_t total = (int64_t)B * T * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SRWN_BF16) hipLaunchKernelGGL(nc_input_kernel<bf16_t>, grid, block, 0, st, x, w, bias, (bf16_t*)a, B, T, C, K);
  else if (dtype == SRWN_F32) hipLaunchKernelGGL(nc_input_kernel<float>, grid, block, 0, st, x, w, bias, (float*)a, B, T, C, K);
  else return set_error(SRWN_E_DTYPE, "nc_input_fwd: dtype %d", dtype);
  return check_launch("nc_input_fwd");
}

extern "C" int srwn_small_gemm(const void* a, int64_t lda, int32_t a_chunk, int64_t a_chunk_stride, int32_t a_dtype,
                               const float* b, int64_t ldb_k, int64_t ldb_n, int32_t b_chunk, int64_t b_chunk_stride,
                               const float* bias, void* c, int64_t ldc, int32_t c_dtype, int32_t M, int32_t N,
                               int32_t K, int32_t accumulate, void* stream) {
  if (M == 0 || N == 0) return 0;
  if (!a || !b || !c) return set_error(SRWN_E_NULL, "small_gemm: null pointer");
  if (M < 0 || N < 0 || K < 1 || a_chunk < 1 || b_chunk < 1 || K % a_chunk || K % b_chunk ||
      (a_chunk % b_chunk && b_chunk % a_chunk))
    return set_error(SRWN_E_SHAPE, "small_gemm: M=%d N=%d K=%d a_chunk=%d b_chunk=%d", M, N, K, a_chunk, b_chunk);
  const int64_t total = (int64_t)M * N;
  dim3 grid((unsigned)((total + 15) / 16)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define SRWN_SG(TA, TC)                                                                                          \
  hipLaunchKernelGGL((small_gemm_kernel<TA, TC>), grid, block, 0, st, (const TA*)a, lda, a_chunk, a_chunk_stride, \
                     b, ldb_k, ldb_n, b_chunk, b_chunk_stride, bias, (TC*)c, ldc, M, N, K, accumulate)
  if (a_dtype == SRWN_BF16 && c_dtype == SRWN_F32) SRWN_SG(bf16_t, float);
  else if (a_dtype == SRWN_F32 && c_dtype == SRWN_F32) SRWN_SG(float, float);
  else if (a_dtype == SRWN_F32 && c_dtype == SRWN_BF16) SRWN_SG(float, bf16_t);
  else if (a_dtype == SRWN_BF16 && c_dtype == SRWN_BF16) SRWN_SG(bf16_t, bf16_t);
  else return set_error(SRWN_E_DTYPE, "small_gemm: dtypes %d -> %d", a_dtype, c_dtype);
#undef SRWN_SG
  return check_launch("small_gemm");
}

extern "C" int srwn_small_wgrad(const float* a, int64_t lda, const float* d, int64_t ldd, float* c, float* bias_out,
                                int32_t M, int32_t K, int32_t N, float scale, void* stream) {
  if (K == 0 || N == 0) return 0;
  if (!a || !d || !c) return set_error(SRWN_E_NULL, "small_wgrad: null pointer");
  if (M < 0 || K < 0 || N < 0) return set_error(SRWN_E_SHAPE, "small_wgrad: M=%d K=%d N=%d", M, K, N);
  const int64_t total = (int64_t)(K + 1) * N;
  hipLaunchKernelGGL(small_wgrad_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, (hipStream_t)stream, a,
                     lda, d, ldd, c, bias_out, M, K, N, scale);
  return check_launch("small_wgrad");
}

extern "C" int srwn_mol_sample(const float* logits, int64_t ldl, int32_t M, const float* u1, const float* u2,
                               float* out, int64_t rows, void* stream) {
  if (rows == 0) return 0;
  if (!logits || !u1 || !u2 || !out) return set_error(SRWN_E_NULL, "mol_sample: null pointer");
  if (rows < 0 || M < 1 || M > 16 || ldl < 3 * M) return set_error(SRWN_E_SHAPE, "mol_sample: rows=%lld M=%d ldl=%lld", (long long)rows, M, (long long)ldl);
  hipLaunchKernelGGL(mol_sample_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, ldl, M, u1, u2, out, rows);
  return check_launch("mol_sample");
}
