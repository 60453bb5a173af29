// The remaining free functions of the reference's ops.py as device kernels (plain VALU: none of these is on the timed
// path; they exist so that `from ops import *` finds every name of ops.py backed by HIP, not by a CPU fallback):
//   log_prob_from_logits / log_sum_exp (ops.py:111-122), categorical_sample (ops.py:106-109), probs_logistic
//   (ops.py:203-214), per-position discretized_mix_logistic_loss (ops.py:124-175, sum_all=False), and the elementwise
//   pieces of the generic-shape ResidualDilationLayer / ResidualDilationLayerNC (ops.py:23-58).
// gfx950 (MI355X) only.
#include <cmath>
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

namespace {

// one wave per row: max, sum-exp over the last axis in fp32, fixed butterfly order (deterministic)
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          float* __restrict__ lse, int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.0f;
  for (int c = lane; c < C; c += 64) s += expf(xr[c] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float l = logf(s);
  if (lse && lane == 0) lse[row] = m + l;                       // ops.py:122
  if (y)
    for (int c = lane; c < C; c += 64) y[row * C + c] = xr[c] - m - l;   // ops.py:115
}

// counter-based uniform in (0,1): splitmix64 of (seed, index)
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);
}

// tf.multinomial(logits - max, 1) (ops.py:107): one draw per row from softmax(logits), as Gumbel-max
__global__ __launch_bounds__(256) void categorical_sample_kernel(const float* __restrict__ logits,
                                                                 int32_t* __restrict__ out, int64_t rows, int C,
                                                                 uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float best = -INFINITY;
  int arg = 0;
  for (int c = lane; c < C; c += 64) {
    const float u = uniform01(seed, (uint64_t)row * C + c);
    const float v = logits[row * C + c] - logf(-logf(u));
    if (v > best) { best = v; arg = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  if (lane == 0) out[row] = arg;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float softplusf_(float v) { return v > 0.0f ? v + log1pf(expf(-v)) : log1pf(expf(v)); }

// ops.py:203-214
__global__ void probs_logistic_kernel(const float* __restrict__ scale, const float* __restrict__ mu,
                                      const float* __restrict__ y, float* __restrict__ out, int64_t n, float half_bin,
                                      float scale_min) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sc = fmaxf(scale[i], scale_min);
  const float inv = 1.0f / sc, cy = y[i] - mu[i];
  out[i] = sigmoidf_(inv * (cy + half_bin)) - sigmoidf_(inv * (cy - half_bin));
}

// z = tanh(f), c = z * sigmoid(z)   (ops.py:28,33,36)
__global__ void tanh_gate_kernel(const float* __restrict__ f, float* __restrict__ z, float* __restrict__ c, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float zz = tanhf(f[i]);
  z[i] = zz;
  c[i] = zz * sigmoidf_(zz);
}

// gate_mode of SURVEY 8(b): SRWN_GATE_REFERENCE = the graph the reference RUNS (ops.py:33 overwrites the gate conv's
// result: c = z * sigmoid(z), z = tanh(f); g is not read) / SRWN_GATE_WAVENET = the canonical unit ops.py:31-32 builds and
// discards: c = tanh(f) * sigmoid(g), g = the gate conv's output
__global__ void gated_activation_kernel(const float* __restrict__ f, const float* __restrict__ g, float* __restrict__ z,
                                        float* __restrict__ c, int64_t n, int wavenet) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float zz = tanhf(f[i]);
  z[i] = zz;
  c[i] = zz * sigmoidf_(wavenet ? g[i] : zz);
}

// dense = (inputs + residual) * sqrt(.5) (ops.py:40); inputs [rows, cin] broadcasts over channels when cin == 1
__global__ void residual_combine_kernel(const float* __restrict__ x, int cin, const float* __restrict__ res, int R,
                                        float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = i / R;
  const int ch = (int)(i - row * R);
  const float xv = cin == 1 ? x[row] : x[row * cin + ch];
  out[i] = (xv + res[i]) * kSqrtHalf;
}

__global__ void relu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fmaxf(x[i], 0.0f);
}

// per-position  -log_sum_exp_m( log p_m(x) + log_softmax(logit_probs)_m )   (ops.py:124-175 with sum_all=False)
__global__ void mol_nll_rows_kernel(const float* __restrict__ l, int64_t ldl, const float* __restrict__ x, int M,
                                    float* __restrict__ out, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float* lr = l + row * ldl;
  const float xv = x[row];
  float mp = -INFINITY;
  for (int m = 0; m < M; ++m) mp = fmaxf(mp, lr[m]);
  float sp = 0.0f;
  for (int m = 0; m < M; ++m) sp += expf(lr[m] - mp);
  const float lsp = mp + logf(sp);
  float best = -INFINITY;
  float lp[64];
  for (int m = 0; m < M; ++m) {
    const float mean = lr[M + m], ls = fmaxf(lr[2 * M + m], -7.0f);
    const float cx = xv - mean, inv = expf(-ls);
    const float plus_in = inv * (cx + 1.0f / 255.0f), min_in = inv * (cx - 1.0f / 255.0f), mid_in = inv * cx;
    const float cdf_delta = sigmoidf_(plus_in) - sigmoidf_(min_in);
    float v;
    if (xv < -0.999f) v = plus_in - softplusf_(plus_in);
    else if (xv > 0.999f) v = -softplusf_(min_in);
    else if (cdf_delta > 1e-5f) v = logf(fmaxf(cdf_delta, 1e-12f));
    else v = mid_in - ls - 2.0f * softplusf_(mid_in) - logf(127.5f);
    v += lr[m] - lsp;
    lp[m] = v;
    best = fmaxf(best, v);
  }
  float s = 0.0f;
  for (int m = 0; m < M; ++m) s += expf(lp[m] - best);
  out[row] = -(best + logf(s));
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int srwn_log_softmax(const float* x, float* y, float* lse, int64_t rows, int32_t C, void* stream) {
  if (rows == 0) return 0;
  if (!x || (!y && !lse)) return set_error(SRWN_E_NULL, "log_softmax: null pointer");
  if (rows < 0 || C < 1 || (rows + 3) / 4 > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "log_softmax: rows=%lld C=%d", (long long)rows, C);
  hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, lse, rows, C);
  return check_launch("log_softmax");
}

extern "C" int srwn_categorical_sample(const float* logits, int32_t* out, int64_t rows, int32_t C, uint64_t seed,
                                       void* stream) {
  if (rows == 0) return 0;
  if (!logits || !out) return set_error(SRWN_E_NULL, "categorical_sample: null pointer");
  if (rows < 0 || C < 1 || (rows + 3) / 4 > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "categorical_sample: rows=%lld C=%d", (long long)rows, C);
  hipLaunchKernelGGL(categorical_sample_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     logits, out, rows, C, seed);
  return check_launch("categorical_sample");
}

extern "C" int srwn_probs_logistic(const float* scale, const float* mu, const float* y, float* out, int64_t n,
                                   int32_t num_classes, float log_scale_min, void* stream) {
  if (n == 0) return 0;
  if (!scale || !mu || !y || !out) return set_error(SRWN_E_NULL, "probs_logistic: null pointer");
  if (n < 0 || num_classes < 2) return set_error(SRWN_E_SHAPE, "probs_logistic: n=%lld num_classes=%d", (long long)n, num_classes);
  hipLaunchKernelGGL(probs_logistic_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, scale, mu, y, out, n,
                     1.0f / (float)(num_classes - 1), expf(log_scale_min));
  return check_launch("probs_logistic");
}

extern "C" int srwn_gated_activation(const float* f, const float* g, float* z, float* c, int64_t n, int32_t gate_mode,
                                     void* stream) {
  if (n == 0) return 0;
  if (gate_mode != SRWN_GATE_REFERENCE && gate_mode != SRWN_GATE_WAVENET)
    return set_error(SRWN_E_UNSUPPORTED, "gated_activation: gate_mode %d", gate_mode);
  if (!f || !z || !c || (gate_mode == SRWN_GATE_WAVENET && !g)) return set_error(SRWN_E_NULL, "gated_activation: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "gated_activation: n=%lld", (long long)n);
  hipLaunchKernelGGL(gated_activation_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, f, g, z, c, n,
                     gate_mode == SRWN_GATE_WAVENET ? 1 : 0);
  return check_launch("gated_activation");
}

extern "C" int srwn_tanh_gate(const float* f, float* z, float* c, int64_t n, void* stream) {
  if (n == 0) return 0;
  if (!f || !z || !c) return set_error(SRWN_E_NULL, "tanh_gate: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "tanh_gate: n=%lld", (long long)n);
  hipLaunchKernelGGL(tanh_gate_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, f, z, c, n);
  return check_launch("tanh_gate");
}

extern "C" int srwn_residual_combine(const float* x, int32_t cin, const float* res, int32_t R, float* out,
                                     int64_t rows, void* stream) {
  if (rows == 0) return 0;
  if (!x || !res || !out) return set_error(SRWN_E_NULL, "residual_combine: null pointer");
  if (rows < 0 || R < 1 || (cin != 1 && cin != R))
    return set_error(SRWN_E_SHAPE, "residual_combine: inputs have %d channels, residual %d (must match, or 1: broadcast)", cin, R);
  hipLaunchKernelGGL(residual_combine_kernel, grid1(rows * R), dim3(256), 0, (hipStream_t)stream, x, cin, res, R, out, rows * R);
  return check_launch("residual_combine");
}

extern "C" int srwn_relu(const float* x, float* y, int64_t n, void* stream) {
  if (n == 0) return 0;
  if (!x || !y) return set_error(SRWN_E_NULL, "relu: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "relu: n=%lld", (long long)n);
  hipLaunchKernelGGL(relu_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, x, y, n);
  return check_launch("relu");
}

extern "C" int srwn_mol_nll_rows(const float* logits, int64_t ldl, const float* x, int32_t M, float* out, int64_t rows,
                                 void* stream) {
  if (rows == 0) return 0;
  if (!logits || !x || !out) return set_error(SRWN_E_NULL, "mol_nll_rows: null pointer");
  if (rows < 0 || M < 1 || M > 64 || ldl < 3 * (int64_t)M) return set_error(SRWN_E_SHAPE, "mol_nll_rows: rows=%lld M=%d ldl=%lld", (long long)rows, M, (long long)ldl);
  hipLaunchKernelGGL(mol_nll_rows_kernel, grid1(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, x, M, out, rows);
  return check_launch("mol_nll_rows");
}
