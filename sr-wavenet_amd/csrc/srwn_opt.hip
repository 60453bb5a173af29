// Optimizer: tf.train.AdamOptimizer semantics (model.py:31,117,382) on the flat fp32 parameter buffer.
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

// The step counter lives on the device so that a captured hipGraph replays with the right bias correction.  It is ticked
// INSIDE the update launch (it used to be a one-thread launch of its own in front of it: ~5 us + a launch boundary per
// step): every block reads t = count + 1 when it starts; each block adds one to an arrival counter when it is done, and
// the block whose add comes last -- every other block has read the count by then -- stores the new count.  No payload
// crosses workgroups: the arrival counter is the upper half of the 64-bit step word, zero again when the launch ends
// (the final store writes the whole word), so the word reads as a plain int64 step count between launches.
constexpr int kAdamBlocks = 256;      // one block per CU
static unsigned adam_grid(int64_t n) {
  const int64_t b = (n + 1023) / 1024;
  return (unsigned)(b < kAdamBlocks ? b : kAdamBlocks);
}
static int adam_vec(const void* a, const void* b, const void* c, const void* d) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

// Four parameters per thread; the bias-corrected step size (two fp64 pow per call of the TF formula) is computed once
// per block and broadcast -- every thread used to evaluate it for itself, which made a 28-MB streaming kernel
// instruction-bound (18 us for 1 M parameters).  Same expression, same bits.
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ theta, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        int64_t* step, int tick, float lr, float b1, float b2,
                                                        float eps, float grad_scale,
                                                        const float* __restrict__ scale_dev, int vec) {
  __shared__ float s_lr;
  __shared__ int s_t;
  if (threadIdx.x == 0) {
    // (an sc1 load: the low word only -- the upper one is the arrival counter of this very launch)
    const int count = __hip_atomic_load(reinterpret_cast<int*>(step), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_t = count + (tick ? 1 : 0);
    const double t = (double)s_t;
    // lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); epsilon is added to the UNcorrected sqrt(v) (TF formula)
    s_lr = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const float lr_t = s_lr;
  if (scale_dev) grad_scale *= scale_dev[0];
  auto one = [&](float gi, float& mi, float& vi, float& th) {
    gi *= grad_scale;
    mi = b1 * mi + (1.0f - b1) * gi;
    vi = b2 * vi + (1.0f - b2) * gi * gi;
    th -= lr_t * mi / (sqrtf(vi) + eps);
  };
  // at most kAdamBlocks blocks walk the buffer (grid stride): every block is one arrival on ONE counter, and arrivals on
  // one address are served one after the other at the memory side (~12 ns each: 980 blocks -- one per 1024 parameters --
  // made this a 18-us kernel; 256 arrivals cost ~3 us)
  for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x * 4) {
    if (vec && i0 + 4 <= n) {      // vec: all four buffers 16-byte aligned (host check)
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 gg = *reinterpret_cast<const f4*>(g + i0), mm = *reinterpret_cast<f4*>(m + i0), vv = *reinterpret_cast<f4*>(v + i0),
         tt = *reinterpret_cast<f4*>(theta + i0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { float a = mm[j], b = vv[j], c = tt[j]; one(gg[j], a, b, c); mm[j] = a; vv[j] = b; tt[j] = c; }
      *reinterpret_cast<f4*>(m + i0) = mm;
      *reinterpret_cast<f4*>(v + i0) = vv;
      *reinterpret_cast<f4*>(theta + i0) = tt;
    } else {
      for (int64_t i = i0; i < n && i < i0 + 4; ++i) one(g[i], m[i], v[i], theta[i]);
    }
  }
  if (tick && threadIdx.x == 0) {      // (thread 0 read the count before this point: program order)
    unsigned* arrived = reinterpret_cast<unsigned*>(step) + 1;
    const unsigned prev = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == gridDim.x - 1)      // the last block: the new count, the arrival counter back to zero, in one 8-byte store
      __hip_atomic_store(step, (int64_t)s_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" int srwn_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step,
                              float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  if (n == 0) return 0;
  if (!params || !grads || !m || !v || !step) return set_error(SRWN_E_NULL, "adam_step: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "adam_step: n=%lld", (long long)n);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_step_kernel, dim3(adam_grid(n)), dim3(256), 0, st, params, grads, m, v, n,
                     step, 1, lr, beta1, beta2, eps, grad_scale, (const float*)nullptr, adam_vec(params, grads, m, v));
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
  hipLaunchKernelGGL(adam_step_kernel, dim3(adam_grid(n)), dim3(256), 0, st, params, grads, m, v, n,
                     step, tick ? 1 : 0, lr, beta1, beta2, eps, 1.0f, grad_scale_dev, adam_vec(params, grads, m, v));
  return check_launch("adam_step_scaled");
}
