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

// ----------------------------------------------------------------------------------------------
// "Weight-gradient tiles": 32 consecutive positions of a segment x R channels, stored TRANSPOSED (fragment by fragment of
// 16 channels, see wt_load) so that a lane's eight consecutive elements are eight time steps of one channel -- directly the A fragment of a
// v_mfma_f32_16x16x32_bf16 that contracts over time (lane l: row l & 15, k = 8 (l >> 4) + j).  The forward group kernel
// writes them (the layer input x and the gate output c = z sigmoid z, from the LDS image with transposing reads), the
// backward group kernel loads them as fragments and contracts them with df / G from its own LDS image:
//   dWf[k] = x^T . df(shifted),  dWr = c^T . G       (tf.gradients of ops.py:27,39)
// Element (c, kg, j) of a tile is time step kordW(kg, j) of the tile: the order in which a
// transposing LDS read (ds_read_b64_tr_b16: four rows per read) is free of bank conflicts at the image's row stride of
// 36 dwords (rows 4 apart), and -- the only requirement -- the same for both operands of a product.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ constexpr int kordW(int kg, int j) { return 16 * (kg >> 1) + 2 * (kg & 1) + (j >> 2) + 4 * (j & 3); }

// 32 rows (time) x 16 columns (channels col0..col0+15) of a row-major LDS tile as a 16x16x32 fragment: lane l holds
// column col0 + (l & 15), rows row0 + kordW(l >> 4, j)
template <typename T> struct LdT16;
template <> struct LdT16<bf16_t> {
  static __device__ __forceinline__ Frag<bf16_t> load(const bf16_t* tile, int stride, int row0, int col0, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int kg = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16_t* base = tile + (size_t)(row0 + 16 * (kg >> 1) + 2 * (kg & 1) + 4 * q) * stride + col0 + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + stride));
    Frag<bf16_t> f;
    f.v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    return f;
  }
};
template <> struct LdT16<float> {
  static __device__ __forceinline__ Frag<float> load(const float* tile, int stride, int row0, int col0, int lane) {
    const int kg = lane >> 4;
    const float* base = tile + (size_t)row0 * stride + col0 + (lane & 15);
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.set(j, base[(size_t)kordW(kg, j) * stride]);
    return f;
  }
};

// The same fragment with the lane-dependent part of the address taken out of the loop over tiles: base() once per loop,
// then one add of a wave-uniform row offset per tile and immediate offsets for the column block (the transposing read
// itself is the only other instruction: the contraction loops are issue-bound, not LDS- or MFMA-bound).
template <typename T> struct LdT16p;
template <> struct LdT16p<bf16_t> {
  static __device__ __forceinline__ const bf16_t* base(const bf16_t* tile, int stride, int lane) {
    const int kg = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    return tile + (size_t)(16 * (kg >> 1) + 2 * (kg & 1) + 4 * q) * stride + 4 * p;
  }
  template <int STRIDE> static __device__ __forceinline__ Frag<bf16_t> load(const bf16_t* b, int col0) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + col0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + col0 + STRIDE));
    Frag<bf16_t> f;
    f.v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    return f;
  }
};
template <> struct LdT16p<float> {
  static __device__ __forceinline__ const float* base(const float* tile, int stride, int lane) {
    const int kg = lane >> 4;
    return tile + (size_t)(16 * (kg >> 1) + 2 * (kg & 1)) * stride + (lane & 15);
  }
  template <int STRIDE> static __device__ __forceinline__ Frag<float> load(const float* b, int col0) {
    Frag<float> f;   // kordW(kg, j) - kordW(kg, 0) = (j >> 2) + 4 (j & 3)
#pragma unroll
    for (int j = 0; j < 8; ++j) f.set(j, b[col0 + ((j >> 2) + 4 * (j & 3)) * STRIDE]);
    return f;
  }
};

// one 16x16 output tile, 32-deep contraction (lane l: D row 4 (l >> 4) + r in register r, column l & 15)
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.get(j), b.get(j), acc, 0, 0, 0);
}

// acc + sum_j f[j] * m[j]  (column sums over time for the bias gradients; m = 0/1 row mask in fragment form)
__device__ __forceinline__ float frag_dot(float acc, const Frag<bf16_t>& f, const Frag<bf16_t>& m) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f.v, f.v, 0, 1), __builtin_shufflevector(m.v, m.v, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f.v, f.v, 2, 3), __builtin_shufflevector(m.v, m.v, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f.v, f.v, 4, 5), __builtin_shufflevector(m.v, m.v, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f.v, f.v, 6, 7), __builtin_shufflevector(m.v, m.v, 6, 7), acc, false);
  return acc;
}
__device__ __forceinline__ float frag_dot(float acc, const Frag<float>& f, const Frag<float>& m) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = fmaf(f.get(j), m.get(j), acc);
  return acc;
}

// the fragment of channels c0..c0+15 of a weight-gradient tile in HBM, and its store.  A tile is R/16 fragments of 64
// lanes x 8 elements IN LANE ORDER (lane l = 16 kg + channel: element (c, kg, j) of block c0 at (c0/16)*512 + (16 kg + c)*8
// + j): a wave's load or store of a fragment is one contiguous KiB.  (Round 3 first stored [channel][kg][j]: the same
// KiB, but consecutive lanes 64 bytes apart -- four address-coalescer cycles per quad of lanes instead of one, on every
// fragment of the forward kernel's stores and of the backward kernels' contraction loops.)
template <typename T> __device__ __forceinline__ Frag<T> wt_load(const T* tile, int c0, int lane) {
  // (plain loads: two waves of a workgroup read every fragment -- the output blocks of a 16-channel slice are split over
  // a wave pair -- and the second read has to find the line; non-temporal here: backward group kernels +44 us per step)
  return load_nat(tile + (size_t)(c0 >> 4) * 512 + lane * 8);
}
// (non-temporal, like the z rows and the head's outputs: round 4 A/B on one box, -52 us per step for the three together --
// tiles -16, head outputs -12, z rows -5 alone -- the write-allocated lines of streams nobody reads for a long while had been
// pushing the operands of the kernels that DO run next out of the L2 / Infinity Cache: the skip sum alone -17 us)
__device__ __forceinline__ void wt_store(bf16_t* tile, int c0, int lane, const Frag<bf16_t>& f) {
  __builtin_nontemporal_store(f.v, reinterpret_cast<bf16x8*>(tile + (size_t)(c0 >> 4) * 512 + lane * 8));
}
__device__ __forceinline__ void wt_store(float* tile, int c0, int lane, const Frag<float>& f) {
  float* p = tile + (size_t)(c0 >> 4) * 512 + lane * 8;
  *reinterpret_cast<f32x4*>(p) = f.lo;
  *reinterpret_cast<f32x4*>(p + 4) = f.hi;
}
// rows [0, 32) of an LDS tile (R channels, row stride `stride`) -> a weight-gradient tile; time steps >= hi zeroed
// (they belong to the neighbouring segment, or lie beyond the clip)
template <typename T, int R>
__device__ __forceinline__ void wt_store_tile(T* gtile, const T* rows, int stride, int hi, int lane) {
#pragma unroll
  for (int cb = 0; cb < R / 16; ++cb) {
    Frag<T> f = LdT16<T>::load(rows, stride, 0, 16 * cb, lane);
    if (hi < 32) {
      const int kg = lane >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (kordW(kg, j) >= hi) f.set(j, 0.0f);
    }
    wt_store(gtile, 16 * cb, lane, f);
  }
}

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
