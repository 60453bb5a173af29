// Micro-benchmark: does the fragment-shaped access pattern of the layer kernels limit them?
//  A: per 32-row tile, loads as MFMA B fragments (16 B and 2x8 B per lane at row stride 128 B) and
//     stores in accumulator layout (8 B per lane, 32 rows per instruction)
//  B: same bytes, fully coalesced (64 lanes x 16 B contiguous)
//  C: fragment loads, coalesced stores;  D: coalesced loads, fragment stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int LOADFRAG, int STOREFRAG>
__global__ __launch_bounds__(256) void k(const char* __restrict__ x, char* __restrict__ y1, char* __restrict__ y2,
                                         int ntiles, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int stride = gridDim.x * 4;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += stride) {
    const size_t row0 = (size_t)tile * 32;
    f32x4 acc = {0, 0, 0, 0};
    if (LOADFRAG) {
      const char* r1 = x + (row0 + col) * 128;
      const char* r0 = x + (row0 + col >= (size_t)d ? row0 + col - d : 0) * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc += *reinterpret_cast<const f32x4*>(r0 + 32 * ks + 16 * half);
        f32x2 a = *reinterpret_cast<const f32x2*>(r1 + 32 * ks + 8 * half);
        f32x2 b = *reinterpret_cast<const f32x2*>(r1 + 32 * ks + 16 + 8 * half);
        acc[0] += a[0] + b[0]; acc[1] += a[1] + b[1];
      }
    } else {
      const char* r1 = x + row0 * 128 + lane * 16;
      const char* r0 = x + (row0 >= (size_t)d ? row0 - d : 0) * 128 + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc += *reinterpret_cast<const f32x4*>(r0 + 1024 * i);
        acc += *reinterpret_cast<const f32x4*>(r1 + 1024 * i);
      }
    }
    if (STOREFRAG) {
      char* o1 = y1 + (row0 + col) * 128;
      char* o2 = y2 + (row0 + col) * 128;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        *reinterpret_cast<f32x2*>(o1 + 16 * g + 8 * half) = f32x2{acc[0] + g, acc[1]};
        *reinterpret_cast<f32x2*>(o2 + 16 * g + 8 * half) = f32x2{acc[2] + g, acc[3]};
      }
    } else {
      char* o1 = y1 + row0 * 128 + lane * 16;
      char* o2 = y2 + row0 * 128 + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(o1 + 1024 * i) = acc + (float)i;
        *reinterpret_cast<f32x4*>(o2 + 1024 * i) = acc - (float)i;
      }
    }
  }
}

int main() {
  const int rows = 128000, ntiles = rows / 32;
  char *x, *y1, *y2;
  hipMalloc(&x, (size_t)rows * 128); hipMalloc(&y1, (size_t)rows * 128); hipMalloc(&y2, (size_t)rows * 128);
  hipMemset(x, 1, (size_t)rows * 128);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  auto run = [&](const char* name, auto kern, int blocks) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, y1, y2, ntiles, 512);
    hipEventRecord(s);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, y1, y2, ntiles, 512);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("%-28s blocks=%4d  %.2f us  (%.0f GB/s of 49 MB)\n", name, blocks, ms * 20, 49.152e6 / (ms * 20e-6) / 1e9);
  };
  for (int blocks : {256, 512, 1000}) {
    run("A frag-load  frag-store", k<1, 1>, blocks);
    run("B coal-load  coal-store", k<0, 0>, blocks);
    run("C frag-load  coal-store", k<1, 0>, blocks);
    run("D coal-load  frag-store", k<0, 1>, blocks);
  }
  return 0;
}
