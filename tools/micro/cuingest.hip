// Micro-benchmark: how many bytes per second enter a CU, by source.  One 512-thread workgroup per CU (256 workgroups x
// NROUND rounds); per step every wave takes in
//   H KB from a stream no workgroup shares (HBM: 16-byte-per-lane global loads into registers), and
//   W KB from a 1-MB image that EVERY workgroup walks (L2 after the first touch; LDS-DMA, 1 KiB per wave-instruction,
//   the way rowgemm / colgemm / the head chain stage their weight images),
// with nothing else to do (a xor of the loaded registers keeps the loads alive; the LDS bytes are never read).
// Prints GB/s per CU and chip-wide for (H, W) in {(4,0), (0,4), (4,4), (2,4), (4,2)} KB per wave and step.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/cuingest.hip -o tools/micro/cuingest && tools/micro/cuingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

template <int HK, int WK>
__global__ __launch_bounds__(512) void k(const char* __restrict__ hbm, const char* __restrict__ img, size_t img_bytes,
                                         unsigned* out, int steps, size_t hbm_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 8192;
  const char* hp = hbm + (size_t)blockIdx.x * hbm_per_wg + (size_t)wave * (HK > 0 ? HK : 1) * 1024 + lane * 16;
  u32x4 acc = {0, 0, 0, 0};
  size_t wo = ((size_t)wave * 4096) % img_bytes;
  for (int s = 0; s < steps; ++s) {
    u32x4 v[HK > 0 ? HK : 1];
#pragma unroll
    for (int i = 0; i < HK; ++i) v[i] = *reinterpret_cast<const u32x4*>(hp + (size_t)s * 8 * HK * 1024 + i * 1024);
#pragma unroll
    for (int i = 0; i < WK; ++i) {
      glds16(img + wo + lane * 16, __builtin_amdgcn_readfirstlane(lds + (i & 7) * 1024));
      wo += 8 * 1024; if (wo >= img_bytes) wo -= img_bytes;
    }
#pragma unroll
    for (int i = 0; i < HK; ++i) acc ^= v[i];
    if ((s & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[threadIdx.x] = acc[0];
}

// the same stream written instead of read (NT: non-temporal stores), optionally beside a read stream of RK KB per wave and step
template <int SK, int RK, bool NT>
__global__ __launch_bounds__(512) void kw(char* __restrict__ dst, const char* __restrict__ src, unsigned* out, int steps, size_t per_wg) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* dp = dst + (size_t)blockIdx.x * per_wg + (size_t)wave * SK * 1024 + lane * 16;
  const char* sp = src + (size_t)blockIdx.x * per_wg + (size_t)wave * (RK > 0 ? RK : 1) * 1024 + lane * 16;
  u32x4 acc = {1u + threadIdx.x, 2, 3, 4};
  for (int s = 0; s < steps; ++s) {
    u32x4 v[RK > 0 ? RK : 1];
#pragma unroll
    for (int i = 0; i < RK; ++i) v[i] = *reinterpret_cast<const u32x4*>(sp + (size_t)s * 8 * RK * 1024 + i * 1024);
#pragma unroll
    for (int i = 0; i < SK; ++i) {
      u32x4* q = reinterpret_cast<u32x4*>(dp + (size_t)s * 8 * SK * 1024 + i * 1024);
      if (NT) __builtin_nontemporal_store(acc, q); else *q = acc;
    }
#pragma unroll
    for (int i = 0; i < RK; ++i) acc ^= v[i];
    if ((s & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if ((acc[0] ^ acc[1]) == 0x12345678u) out[threadIdx.x] = acc[0];
}

template <int SK, int RK, bool NT> void runw(char* dst, const char* src, unsigned* out, int cus) {
  const int steps = 1024;
  const size_t per_wg = (size_t)steps * 8 * 4 * 1024;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kw<SK, RK, NT>), dim3(cus), dim3(512), 0, 0, dst, src, out, steps, per_wg);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
  }
  const double wb = (double)cus * steps * 8 * SK * 1024, rb = (double)cus * steps * 8 * RK * 1024;
  printf("per wave and step: %d KB written (%s stores) + %d KB read from another stream: %7.3f ms  -> %5.2f TB/s chip (%5.2f written + %5.2f read)\n",
         SK, NT ? "non-temporal" : "plain", RK, best, (wb + rb) / best / 1e9, wb / best / 1e9, rb / best / 1e9);
}

template <int HK, int WK> void run(const char* hbm, const char* img, size_t img_bytes, unsigned* out, int cus) {
  const int steps = 2048;
  const size_t per_wg = (size_t)steps * 8 * (HK > 0 ? HK : 1) * 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<HK, WK>), dim3(cus), dim3(512), 65536, 0, hbm, img, img_bytes, out, steps, per_wg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
  }
  const double hb = (double)cus * steps * 8 * HK * 1024, wb = (double)cus * steps * 8 * WK * 1024;
  printf("per wave and step: %d KB from HBM (loads) + %d KB from a shared 1-MB image (LDS-DMA): %7.3f ms  -> %6.1f GB/s per CU  (%5.2f TB/s chip: %5.2f HBM + %5.2f L2)\n",
         HK, WK, best, (hb + wb) / cus / best / 1e6, (hb + wb) / best / 1e9, hb / best / 1e9, wb / best / 1e9);
}

int main() {
  int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t img_bytes = 1 << 20;
  const size_t hbm_bytes = (size_t)cus * 2048 * 8 * 4 * 1024;      // 16 GiB at 256 CUs: every byte read once
  char *hbm, *img; unsigned* out;
  if (hipMalloc(&hbm, hbm_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&img, img_bytes); hipMalloc(&out, 4096);
  hipMemset(hbm, 1, hbm_bytes); hipMemset(img, 2, img_bytes);
  hipFuncSetAttribute((const void*)k<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("%d CUs\n", cus);
  run<4, 0>(hbm, img, img_bytes, out, cus);
  run<0, 4>(hbm, img, img_bytes, out, cus);
  run<4, 4>(hbm, img, img_bytes, out, cus);
  run<2, 4>(hbm, img, img_bytes, out, cus);
  run<4, 2>(hbm, img, img_bytes, out, cus);
  run<0, 8>(hbm, img, img_bytes, out, cus);
  char* dst = hbm + hbm_bytes / 2;      // (the write tests use the two halves of the buffer: 8 GiB each)
  runw<4, 0, false>(dst, hbm, out, cus);
  runw<4, 0, true>(dst, hbm, out, cus);
  runw<4, 1, true>(dst, hbm, out, cus);
  runw<2, 2, true>(dst, hbm, out, cus);
  runw<1, 4, true>(dst, hbm, out, cus);
  return 0;
}
