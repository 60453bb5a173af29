// Optimizer: tf.train.AdamOptimizer semantics (model.py:31,117,382) on the flat fp32 parameter buffer.
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

// step counter lives on the device so that a captured hipGraph replays with the right bias correction
__global__ void adam_tick_kernel(int64_t* step) { step[0] += 1; }

__global__ void adam_step_kernel(float* __restrict__ theta, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, const int64_t* __restrict__ step, float lr,
                                 float b1, float b2, float eps, float grad_scale,
                                 const float* __restrict__ scale_dev) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (scale_dev) grad_scale *= scale_dev[0];
  const double t = (double)step[0];
  // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); epsilon is added to the UNcorrected sqrt(v) (TF formula)
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  const float gi = g[i] * grad_scale;
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

extern "C" int srwn_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step,
                              float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  if (n == 0) return 0;
  if (!params || !grads || !m || !v || !step) return set_error(SRWN_E_NULL, "adam_step: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "adam_step: n=%lld", (long long)n);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step);
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, params, grads, m, v, n,
                     step, lr, beta1, beta2, eps, grad_scale, (const float*)nullptr);
  return check_launch("adam_step");
}

// Same update with the gradient scale read from device memory (the clip factor of srwn_clip_scale);
// `tick` = 0 lets several parameter buffers share one step counter (tick it on the first call only).
extern "C" int srwn_adam_step_scaled(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step,
                                     float lr, float beta1, float beta2, float eps, const float* grad_scale_dev,
                                     int32_t tick, void* stream) {
  if (n == 0) return 0;
  if (!params || !grads || !m || !v || !step || !grad_scale_dev)
    return set_error(SRWN_E_NULL, "adam_step_scaled: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "adam_step_scaled: n=%lld", (long long)n);
  hipStream_t st = (hipStream_t)stream;
  if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step);
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, params, grads, m, v, n,
                     step, lr, beta1, beta2, eps, 1.0f, grad_scale_dev);
  return check_launch("adam_step_scaled");
}
