// Building blocks shared by the multi-layer ("group") kernels of the residual stack: srwn_group.hip (forward, and the
// backward data-gradient chain) and srwn_groupw.hip (backward chain + layer weight gradients in one launch).
// gfx950 (MI355X) only.  Segment geometry and time decomposition: header of srwn_group.hip.
#pragma once
#include <cstdint>
#include "srwn_common.h"

namespace srwn {
namespace grp {

constexpr int kMaxGroup = 8;
constexpr int kLdsBudget = 160 * 1024;

template <typename T> struct Raw4g;
template <> struct Raw4g<bf16_t> {
  typedef bf16x4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
  static __device__ __forceinline__ float get(const type& v, int e) { return (float)v[e]; }
  static __device__ __forceinline__ type zero() { return type{(bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f, (bf16_t)0.0f}; }
  static __device__ __forceinline__ type pack(float a, float b, float c, float d) {
    return type{(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
  }
};
template <> struct Raw4g<float> {
  typedef f32x4 type;
  static __device__ __forceinline__ type load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ float get(const type& v, int e) { return v[e]; }
  static __device__ __forceinline__ type zero() { return type{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ type pack(float a, float b, float c, float d) { return type{a, b, c, d}; }
};

// LDS-DMA the compiler does not see (a visible one makes it answer every later wait with vmcnt(0)); retired by
// dma_wait() before the barrier that publishes the buffer.  M0 (LDS base of the DMA) is saved and restored.
__device__ __forceinline__ void glds16_untracked(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// a lane-linear byte image (packed MFMA fragments) global -> LDS, 1 KiB per wave-instruction
__device__ __forceinline__ void dma_image(const void* gsrc, unsigned lds_dst, int nbytes, int wave, int lane,
                                          int nwaves = 8) {
  const char* g = reinterpret_cast<const char*>(gsrc) + lane * 16;
  for (int p = wave; p < nbytes / 1024; p += nwaves) glds16_untracked(g + (size_t)p * 1024, __builtin_amdgcn_readfirstlane(lds_dst + p * 1024));
}

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

inline int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

// common stride, sub-dilations, halo of a run of layers
inline int group_geometry(const int32_t* dil, int nl, int* st_out, int* sub, int* H_out) {
  auto gcd = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
  int st = 0;
  for (int i = 0; i < nl; ++i) {
    if (dil[i] < 1) return -1;
    st = gcd(st, dil[i]);
  }
  int H = 0;
  for (int i = 0; i < nl; ++i) { sub[i] = dil[i] / st; H += sub[i]; }
  *st_out = st; *H_out = H;
  return 0;
}

// positions per segment and tiles per image for a residue class of J positions
// A residue class that fits one segment needs no halo at all (the positions before / beyond it are the conv's zero
// padding): *H becomes 0 then, which for T = 16000 turns the 32..512 groups' 17 tiles into 16 = two per wave.
inline void choose_segments(int J, int* Hp, int B, int st, int nt_max, int seg_rows, int* W, int* NT, int* nsub) {
  const int H = *Hp;
  const int wmax = nt_max * 32 - H;
  int w;
  if (seg_rows > 0) {
    w = seg_rows < wmax ? seg_rows : wmax;
  } else {
    int ns = (J + wmax - 1) / wmax;                       // fewest segments the image size allows
    const long long streams = (long long)B * st;
    long long want = (num_cus() + streams - 1) / streams; // enough segments for one per CU ...
    const int floor_w = 4 * H > 128 ? 4 * H : 128;        // ... while the halo stays <= 25 % of a segment
    long long cap = J / floor_w; if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want > ns) ns = (int)want;
    w = (J + ns - 1) / ns;
  }
  if (w < 1) w = 1;
  *W = w;
  *nsub = (J + w - 1) / w;
  if (*nsub == 1) *Hp = 0;
  *NT = (*Hp + w + 31) / 32;
}

}  // namespace grp
}  // namespace srwn
