// What does the layer_bwd traffic shape (5 streamed inputs, 2 streamed outputs, 16.4 MB each, one launch per
// layer over a 30-layer rotation so nothing but the previous launch's outputs is cache resident) cost as a
// pure copy?  Variants: grid size, loads in flight per wave, nontemporal hints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ g_in, const f32x4* __restrict__ df_up,
                                         const f32x4* __restrict__ dcs, const f32x4* __restrict__ z,
                                         f32x4* __restrict__ g_out, f32x4* __restrict__ df_out, int nvec, int dvec) {
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    const int j = i + dvec < nvec ? i + dvec : i;
    f32x4 a = g_in[i], b = df_up[i], c = df_up[j];
    f32x4 d = NT ? __builtin_nontemporal_load(dcs + i) : dcs[i];
    f32x4 e = NT ? __builtin_nontemporal_load(z + i) : z[i];
    f32x4 g = a + b + c;
    g_out[i] = g;
    df_out[i] = (g + d) * e;
  }
}

int main() {
  const int L = 30;
  const size_t nvec = 128000ull * 128 / 16;
  const size_t bytes = nvec * 16;
  std::vector<f32x4*> gs(L + 2), dfs(L + 1), dcs(L), zs(L);
  for (auto& p : gs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
  for (auto& p : dfs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
  for (auto& p : dcs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
  for (auto& p : zs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  auto run = [&](const char* name, auto kern, int blocks) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(s);
      for (int l = L - 1; l >= 0; --l)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, gs[l + 2], dfs[l + 1], dcs[l], zs[l], gs[l + 1], dfs[l],
                           (int)nvec, 512 * 8);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      if (rep && ms < best) best = ms;
    }
    printf("%-10s blocks=%5d  %.2f us/launch  (%.0f GB/s of 98.3 MB)\n", name, blocks, best * 1e3 / L,
           6 * bytes / (best * 1e-3 / L) / 1e9);
  };
  for (int blocks : {256, 512, 1024, 2048, 4096, 32000}) {
    run("plain", k<0>, blocks);
    run("nt", k<1>, blocks);
  }
  return 0;
}
