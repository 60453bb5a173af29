// Does the end-of-kernel write-back of dirty L2 lines cost the layer kernels their last microseconds, and does a store
// cache policy avoid it?  layer_fwd's traffic shape (1 streamed input, 2 streamed outputs, 16.4 MB each, 30-layer
// rotation) as a pure copy with plain / nontemporal / sc1 / sc0 sc1 stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ void st(f32x4* p, f32x4 v) {
  if (POL == 0) *p = v;
  else if (POL == 1) __builtin_nontemporal_store(v, p);
  else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

template <int POL>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ x, f32x4* __restrict__ y1, f32x4* __restrict__ y2, int nvec) {
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    f32x4 a = x[i];
    st<POL>(y1 + i, a * 2.0f);
    st<POL>(y2 + i, a + 1.0f);
  }
}

int main() {
  const int L = 30;
  const size_t nvec = 128000ull * 128 / 16, bytes = nvec * 16;
  std::vector<f32x4*> xs(L + 1), zs(L);
  for (auto& p : xs) { (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); }
  for (auto& p : zs) { (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); }
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  auto run = [&](const char* name, auto kern, int blocks) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(s);
      for (int l = 0; l < L; ++l) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, xs[l], xs[l + 1], zs[l], (int)nvec);
      (void)hipEventRecord(e); (void)hipEventSynchronize(e);
      float ms; (void)hipEventElapsedTime(&ms, s, e);
      if (rep && ms < best) best = ms;
    }
    printf("%-10s blocks=%5d  %.2f us/launch  (%.0f GB/s of 49.2 MB)\n", name, blocks, best * 1e3 / L, 3 * bytes / (best * 1e-3 / L) / 1e9);
  };
  for (int blocks : {512, 2048}) {
    run("plain", k<0>, blocks);
    run("nt", k<1>, blocks);
    run("sc1", k<2>, blocks);
    run("sc0 sc1", k<3>, blocks);
    run("sc0", k<4>, blocks);
  }
  return 0;
}
