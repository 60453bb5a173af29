// Issue cost of the VALU instructions the layer kernels lean on, one wave per SIMD (no partner): N back-to-back
// independent instructions of one kind, timed with s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 valubench.hip -o valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int KIND>
__global__ void bench(unsigned long long* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7};
  _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1, h2_ = (_Float16)a2, h3 = (_Float16)a3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 16; ++it) {
    if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (KIND == 1) { REP64(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (KIND == 2) { REP64(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (KIND == 3) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
    if (KIND == 4) { REP64(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(h0), "+v"(h1), "+v"(h2_), "+v"(h3));) }
    if (KIND == 5) { REP64(asm volatile("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3" : "+v"(h0), "+v"(h1), "+v"(h2_), "+v"(h3));) }
    if (KIND == 6) { REP64(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (KIND == 7) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));) }
    if (KIND == 8) { REP64(asm volatile("v_pk_fma_f16 %0, %0, %0, %0\n v_pk_fma_f16 %1, %1, %1, %1\n v_pk_fma_f16 %2, %2, %2, %2\n v_pk_fma_f16 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (KIND == 9) { REP64(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (a0 + a1 + a2 + a3 + p0[0] + p1[0] + p2[0] + p3[0] + (float)h0 + (float)h1 + (float)h2_ + (float)h3 == 12345.678f) out[0] = 0;
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 4096 * sizeof(unsigned long long));
  const char* names[] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_pk_fma_f32", "v_exp_f16", "v_rcp_f16", "v_cvt_pk_bf16_f32", "v_pk_mul_f32", "v_pk_fma_f16", "v_sqrt_f32"};
  for (int waves = 1; waves <= 4; waves *= 2) {        // waves per SIMD (block of 256, 512 or 1024 threads, one block per CU)
    for (int k = 0; k < 10; ++k) {
      std::vector<unsigned long long> h(1024);
      for (int rep = 0; rep < 2; ++rep) {
        dim3 g(256), b(256 * waves);
        switch (k) {
#define C(K) case K: hipLaunchKernelGGL(bench<K>, g, b, 0, 0, d, 1.0f); break;
          C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9)
        }
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), d, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < 256; ++i) s += (double)h[i];
      const double n = 16.0 * 64 * 4;
      printf("%-20s %d wave(s)/SIMD: %.2f s_memtime ticks per instruction per wave\n", names[k], waves, s / 256 / n);
    }
  }
  // chip-wide throughput: 2048 blocks of 1024 threads (two resident per CU = 8 waves per SIMD), timed with events
  for (int k : {0, 3, 1}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      dim3 g(2048), b(1024);
      switch (k) {
        C(0) C(1) C(3)
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = 2048.0 * 16 * 16.0 * 64 * 4;      // wave-instructions
    printf("%-20s 8 waves/SIMD: %.3f ms -> %.2f wave-instructions per ns chip-wide = %.3f per SIMD per ns\n", names[k], ms,
           winstr / (ms * 1e6), winstr / (ms * 1e6) / 1024);
  }
  return 0;
}
