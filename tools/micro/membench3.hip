// layer_bwd traffic (5 input streams, 2 output streams over a 30-layer rotation) with the kernel's actual
// per-lane load shapes: accumulator-layout 8-B loads (g_in, dcs, z) and B-fragment 16-B loads (two df taps),
// against the same bytes loaded as whole rows.  Stores are whole rows in both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int SHAPED>
__global__ __launch_bounds__(256) void k(const char* __restrict__ g_in, const char* __restrict__ df_up,
                                         const char* __restrict__ dcs, const char* __restrict__ z,
                                         char* __restrict__ g_out, char* __restrict__ df_out, int ntiles, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int stride = gridDim.x * 4;
  const int nrows = ntiles * 32;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += stride) {
    const size_t row0 = (size_t)tile * 32;
    f32x4 acc = {0, 0, 0, 0};
    if (SHAPED) {
      const size_t r = row0 + col, r2 = r + d < (size_t)nrows ? r + d : r;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        f32x2 a = *reinterpret_cast<const f32x2*>(g_in + r * 128 + 16 * g + 8 * half);
        f32x2 b = *reinterpret_cast<const f32x2*>(dcs + r * 128 + 16 * g + 8 * half);
        f32x2 c = *reinterpret_cast<const f32x2*>(z + r * 128 + 16 * g + 8 * half);
        acc[0] += a[0] + b[0] * c[0]; acc[1] += a[1] + b[1] * c[1];
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc += *reinterpret_cast<const f32x4*>(df_up + r * 128 + 32 * ks + 16 * half);
        acc += *reinterpret_cast<const f32x4*>(df_up + r2 * 128 + 32 * ks + 16 * half);
      }
    } else {
      const size_t o = row0 * 128 + lane * 16;
      const size_t o2 = (row0 + d < (size_t)nrows ? row0 + d : row0) * 128 + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc += *reinterpret_cast<const f32x4*>(g_in + o + 1024 * i);
        acc += *reinterpret_cast<const f32x4*>(dcs + o + 1024 * i) * *reinterpret_cast<const f32x4*>(z + o + 1024 * i);
        acc += *reinterpret_cast<const f32x4*>(df_up + o + 1024 * i);
        acc += *reinterpret_cast<const f32x4*>(df_up + o2 + 1024 * i);
      }
    }
    char* o1 = g_out + row0 * 128 + lane * 16;
    char* o2 = df_out + row0 * 128 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(o1 + 1024 * i) = acc + (float)i;
      *reinterpret_cast<f32x4*>(o2 + 1024 * i) = acc - (float)i;
    }
  }
}

int main() {
  const int L = 30, rows = 128000, ntiles = rows / 32;
  const size_t bytes = (size_t)rows * 128;
  std::vector<char*> gs(L + 2), dfs(L + 1), dcs(L), zs(L);
  for (auto* v : {&gs, &dfs, &dcs, &zs})
    for (auto& p : *v) { (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); }
  hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  auto run = [&](const char* name, auto kern, int blocks) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(s);
      for (int l = L - 1; l >= 0; --l)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, gs[l + 2], dfs[l + 1], dcs[l], zs[l], gs[l + 1], dfs[l],
                           ntiles, 512);
      (void)hipEventRecord(e); (void)hipEventSynchronize(e);
      float ms; (void)hipEventElapsedTime(&ms, s, e);
      if (rep && ms < best) best = ms;
    }
    printf("%-10s blocks=%5d  %.2f us/launch  (%.0f GB/s of 98.3 MB)\n", name, blocks, best * 1e3 / L,
           6 * bytes / (best * 1e-3 / L) / 1e9);
  };
  for (int blocks : {256, 512, 1000}) {
    run("rows", k<0>, blocks);
    run("shaped", k<1>, blocks);
  }
  return 0;
}
