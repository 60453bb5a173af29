// The chunk loop of the row-streaming GEMMs (rowgemm / colgemm / head chain) with nothing else in it: one workgroup of
// eight waves per CU; per chunk every wave runs 32 v_mfma_f32_32x32x16_bf16 (8 row tiles x 4 k-steps: 1 024 cycles of the
// matrix pipe) whose A fragments come from a 32-KB weight chunk in LDS (ds_read_b128, one per MFMA, a k-step ahead),
// the next chunk streams into the other LDS buffer by LDS-DMA from an L2-resident image, one barrier per chunk.
// Variants switch single ingredients off, to see which one sets the time per chunk:
//   bit 0: no LDS-DMA (the two buffers are filled once)      bit 1: no barrier / wait per chunk
//   bit 2: A fragments from registers (no ds_read)           bit 3: the same FLOPs as 64 v_mfma_f32_16x16x32_bf16
//   (same LDS bytes, same registers: 16 row blocks x 2 k-steps of A fragments, each used for two 16-column blocks)
// Build: hipcc --offload-arch=gfx950 -O3 gemmloop.hip -o gemmloop ; prints microseconds per chunk and the matrix-pipe share.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int MT = 8, KSC = 4, FB = 1024, CHUNK_B = MT * KSC * FB;   // 32 KB

__device__ __forceinline__ void glds16(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

template <int V>
__global__ __launch_bounds__(512) void loop_kernel(const char* wimg, int nimg_chunks, int nchunks, float* out) {
  constexpr bool NODMA = V & 1, NOBAR = V & 2, NOLDS = V & 4, M16 = V & 8, PAIR = V & 16;   // PAIR: one barrier per TWO chunks (four buffers)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x / 64;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto stage = [&](int c, int buf) {
    const int per = (CHUNK_B / 1024) / nw;
    for (int i = 0; i < per; ++i) {
      const int p = wave * per + i;
      glds16(wimg + (size_t)(c % nimg_chunks) * CHUNK_B + p * 1024 + lane * 16,
             __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * CHUNK_B + p * 1024)));
    }
  };
  stage(0, 0);
  stage(1, 1);
  if (PAIR) { stage(2, 2); stage(3, 3); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 acc[MT];
  for (int m = 0; m < MT; ++m)
    for (int q = 0; q < 16; ++q) acc[m][q] = 0.0f;
  bf16x8 b[KSC];
  for (int k = 0; k < KSC; ++k)
    for (int j = 0; j < 8; ++j) b[k][j] = (__bf16)(0.001f * (lane + j + k));
  bf16x8 areg[MT];
  for (int m = 0; m < MT; ++m)
    for (int j = 0; j < 8; ++j) areg[m][j] = (__bf16)(0.002f * (lane + j + m));
  for (int c = 0; c < nchunks; ++c) {
    if (PAIR) {
      // (chunks c, c+1 were published by the last barrier; c+2, c+3 are in flight into the other two buffers)
    } else if (!NODMA && c + 1 < nchunks) stage(c + 1, (c + 1) & 1);
    const bf16x8* lw = reinterpret_cast<const bf16x8*>(smem + (PAIR ? (c & 3) : (c & 1)) * CHUNK_B) + lane;
    if (M16) {
      // 16 row blocks x 2 k-steps (32 deep) = the same 32 fragments; acc16[rb][cb] = 16x16 tiles
      f32x4 (&a16)[32] = reinterpret_cast<f32x4 (&)[32]>(acc);
      // four groups of eight fragments (k-step ks = g >> 1, row blocks 8 (g & 1) ..), the next group read ahead
      bf16x8 af[2][8];
      if (!NOLDS) {
#pragma unroll
        for (int m = 0; m < 8; ++m) af[0][m] = lw[(m * 2) * 64];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (!NOLDS && g + 1 < 4) {
#pragma unroll
          for (int m = 0; m < 8; ++m) af[(g + 1) & 1][m] = lw[((8 * ((g + 1) & 1) + m) * 2 + ((g + 1) >> 1)) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const int rb = 8 * (g & 1) + m;
            a16[2 * rb + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(NOLDS ? areg[m] : af[g & 1][m], b[2 * (g >> 1) + cb], a16[2 * rb + cb], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (NOLDS) {
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[m], b[ks], acc[m], 0, 0, 0);
    } else {
      bf16x8 af[2][MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[0][m] = lw[(m * KSC) * 64];
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        if (ks + 1 < KSC) {
#pragma unroll
          for (int m = 0; m < MT; ++m) af[(ks + 1) & 1][m] = lw[(m * KSC + ks + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][m], b[ks], acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (PAIR) {
      if (c & 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!NODMA) { stage(c + 3, (c + 3) & 3); stage(c + 4, (c + 4) & 3); }   // into the buffers just read
      }
    } else if (!NOBAR) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  float s = 0.0f;
  for (int m = 0; m < MT; ++m)
    for (int q = 0; q < 16; ++q) s += acc[m][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
static void run(const char* name, const char* wimg, int nimg, float* out, int waves) {
  const int nchunks = 120, grid = 256;
  hipFuncSetAttribute((const void*)loop_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * CHUNK_B);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(loop_kernel<V>, dim3(grid), dim3(64 * waves), 4 * CHUNK_B, 0, wimg, nimg, nchunks, out);
  hipEventRecord(e0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(loop_kernel<V>, dim3(grid), dim3(64 * waves), 4 * CHUNK_B, 0, wimg, nimg, nchunks, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us_chunk = ms * 1e3 / reps / nchunks;
  // matrix-pipe time of one chunk: waves/4 per SIMD x 32 MFMAs x 32 cycles at 2.4 GHz
  const double mfma_us = (waves / 4.0) * 32 * 32 / 2400.0;
  printf("%-58s %d waves: %.3f us per chunk (MFMA alone %.3f us at 2.4 GHz: %.0f %%), %.0f TFLOP/s\n", name, waves, us_chunk, mfma_us,
         100.0 * mfma_us / us_chunk, grid * waves * 32.0 * 32768 / us_chunk * 1e-6);
}

int main() {
  const int nimg = 30;
  char* wimg; float* out;
  hipMalloc(&wimg, (size_t)nimg * CHUNK_B);
  hipMemset(wimg, 0x11, (size_t)nimg * CHUNK_B);
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run<0>("as the GEMMs run it (DMA + LDS reads + barrier)", wimg, nimg, out, 8);
  run<1>("no LDS-DMA", wimg, nimg, out, 8);
  run<2>("no barrier / wait (DMA still issued)", wimg, nimg, out, 8);
  run<3>("no DMA, no barrier", wimg, nimg, out, 8);
  run<4>("A fragments from registers (DMA + barrier kept)", wimg, nimg, out, 8);
  run<7>("MFMAs only", wimg, nimg, out, 8);
  run<0>("as the GEMMs run it", wimg, nimg, out, 4);
  run<3>("no DMA, no barrier", wimg, nimg, out, 4);
  run<7>("MFMAs only", wimg, nimg, out, 4);
  run<8>("16x16x32: as the GEMMs run it", wimg, nimg, out, 8);
  run<11>("16x16x32: no DMA, no barrier", wimg, nimg, out, 8);
  run<15>("16x16x32: MFMAs only", wimg, nimg, out, 8);
  run<16>("one barrier per two chunks (four buffers)", wimg, nimg, out, 8);
  run<24>("16x16x32 + one barrier per two chunks", wimg, nimg, out, 8);
  return 0;
}
