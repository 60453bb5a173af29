// Shared device-side building blocks for the gfx950 (MI355X / CDNA4) WaveNet kernels.
//
// Orientation used by every MFMA kernel in this library ("channels on rows, time on lanes"):
//   D[n][t] = sum_k A[n][k] * B[k][t]      A = weights (W^T), B = activations^T, D = output^T
// with v_mfma_f32_32x32x16_bf16 (bf16 mode) or 8 x v_mfma_f32_32x32x2_f32 (exact fp32 mode).
// * B fragments are 8 consecutive channels of one time row of a channels-last [rows][C] tensor:
//   one 16-byte (bf16) load per lane, no LDS, no transpose.
// * The accumulator tile (lane = time column, registers = channels) is directly the B operand of
//   the next product that contracts over channels (gate -> 1x1 residual, dgrad chains), so the
//   fused residual layer needs no LDS round trip for activations.
// Lane maps (MI355X guide, "Fragment layout"): lane l: r = l&31, h = l>>5.
//   A frag element j  = A[row r][k = kord(h,j)],  B frag element j = B[k = kord(h,j)][col r]
//   C/D register  q   = D[row (q&3) + 8*(q>>2) + 4*h][col r]
// "natural" k order kord = 8h + j; "permuted" k order kord = 8*(j>>2) + 4h + (j&3), which is the
// order in which accumulator registers 8s..8s+7 present their rows when reused as a B operand.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srwn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr float kSqrtHalf = 0.7071067811865476f;  // reference literal, ops.py:40

// ----------------------------------------------------------------------------------------------
// fragments: 8 elements of T per lane
// ----------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ float get(int j) const { return (float)v[j]; }
  __device__ __forceinline__ void set(int j, float x) { v[j] = (bf16_t)x; }
};
template <> struct Frag<float> {
  f32x4 lo, hi;
  __device__ __forceinline__ float get(int j) const { return j < 4 ? lo[j] : hi[j - 4]; }
  __device__ __forceinline__ void set(int j, float x) {
    if (j < 4) lo[j] = x; else hi[j - 4] = x;
  }
};

template <typename T> __device__ __forceinline__ Frag<T> zero_frag();
template <> __device__ __forceinline__ Frag<bf16_t> zero_frag<bf16_t>() {
  Frag<bf16_t> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (bf16_t)0.0f;
  return f;
}
template <> __device__ __forceinline__ Frag<float> zero_frag<float>() {
  Frag<float> f;
  f.lo = f32x4{0.f, 0.f, 0.f, 0.f};
  f.hi = f32x4{0.f, 0.f, 0.f, 0.f};
  return f;
}

// natural order: p points at the lane's first element (caller adds +8*h)
__device__ __forceinline__ Frag<bf16_t> load_nat(const bf16_t* p) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(p);
  return f;
}
__device__ __forceinline__ Frag<float> load_nat(const float* p) {
  Frag<float> f;
  f.lo = *reinterpret_cast<const f32x4*>(p);
  f.hi = *reinterpret_cast<const f32x4*>(p + 4);
  return f;
}
// permuted order: elements 0-3 at p[0..3], 4-7 at p[8..11] (caller adds +4*h)
__device__ __forceinline__ Frag<bf16_t> load_perm(const bf16_t* p) {
  bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
  bf16x4 b = *reinterpret_cast<const bf16x4*>(p + 8);
  Frag<bf16_t> f;
  f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
__device__ __forceinline__ Frag<float> load_perm(const float* p) {
  Frag<float> f;
  f.lo = *reinterpret_cast<const f32x4*>(p);
  f.hi = *reinterpret_cast<const f32x4*>(p + 8);
  return f;
}

// one 32x32 output tile, 16-deep contraction
__device__ __forceinline__ void mma(f32x16& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.lo[j], b.lo[j], acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.hi[j], b.hi[j], acc, 0, 0, 0);
}

// row (channel) held by accumulator register q in lane half h, within its 32-row tile
__device__ __forceinline__ constexpr int crow(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// 4 consecutive channels of one time row, from/to accumulator group g (registers 4g..4g+3)
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (bf16_t)a; v[1] = (bf16_t)b; v[2] = (bf16_t)c; v[3] = (bf16_t)d;
  *reinterpret_cast<bf16x4*>(p) = v;
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// ----------------------------------------------------------------------------------------------
// weight images -> LDS with LDS-DMA (global_load_lds_dwordx4): 1-KiB lane-linear pieces, no VGPR round
// trip and -- unlike a load/ds_write loop -- no dependent wait per piece: every piece of every wave is in
// flight at once; the caller's __syncthreads() (vmcnt(0) + barrier) retires them.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_dma_copy(const void* gsrc, void* lds_dst, int nbytes, int wave, int lane,
                                             int nwaves) {
  const char* g = reinterpret_cast<const char*>(gsrc) + lane * 16;
  char* l = reinterpret_cast<char*>(lds_dst);
  for (int p = wave; p < nbytes / 1024; p += nwaves)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)p * 1024),
                                     (__attribute__((address_space(3))) void*)(l + (size_t)p * 1024), 16, 0, 0);
}

// ----------------------------------------------------------------------------------------------
// accumulator-layout tile -> HBM as whole rows.  In accumulator layout a lane owns ONE time row, so a
// direct store writes 8 bytes into each of 32 different cache lines per instruction; measured on
// MI355X (tools/micro/membench.hip) that partial-line pattern caps a layer kernel at ~2.8 TB/s versus
// ~4.9 TB/s for whole-row stores.  So tiles are transposed through a wave-private LDS buffer
// ([32 rows][W + pad] elements) and leave as 16 bytes per lane, consecutive lanes = consecutive bytes.
// The buffer is private to the wave: LDS executes one wave's instructions in order, so only the
// compiler has to be kept from reordering (the asm memory clobbers); no barrier is involved.
// ----------------------------------------------------------------------------------------------
template <typename T> struct RowStage {
  static constexpr int VEC = 16 / (int)sizeof(T);   // elements per 16-byte piece
  static __host__ __device__ __forceinline__ constexpr int stride(int W) { return W + VEC; }   // padded row, 16-B aligned
};

__device__ __forceinline__ void wave_lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// vals[mt][q]: accumulator-layout values of MTN row tiles (W = 32*MTN channels) for this lane's time row
// `col`; gtile points at channel 0 of the tile's first row in HBM, rows `grow_stride` elements apart.
// DUP = true (needs rows_valid >= 1): rows beyond rows_valid re-store the last valid row instead of being skipped, so
// every store instruction is issued unconditionally -- with the skip, each one sits behind an execz branch, the
// compiler can no longer count the stores that follow a prefetch, and its s_waitcnt for the prefetched registers
// becomes vmcnt(0): every tile then waits for the previous tile's stores to be acknowledged.
template <typename T, int MTN, bool DUP = false>
__device__ __forceinline__ void store_rows_via_lds(T* stage, T* gtile, int64_t grow_stride, const float (&vals)[MTN][16],
                                                   int rows_valid, int lane) {
  constexpr int W = 32 * MTN, LS = RowStage<T>::stride(W), VEC = RowStage<T>::VEC;
  constexpr int LPR = W / VEC;          // lanes per row when reading back
  constexpr int RPI = 64 / LPR;         // rows per store instruction
  const int col = lane & 31, half = lane >> 5;
  wave_lds_order();                     // earlier reads of the buffer are done
#pragma unroll
  for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      store4(stage + col * LS + 32 * mt + 8 * g + 4 * half, vals[mt][4 * g], vals[mt][4 * g + 1], vals[mt][4 * g + 2],
             vals[mt][4 * g + 3]);
  wave_lds_order();
  const int rsub = lane / LPR, piece = lane % LPR;
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) {
    int r = i * RPI + rsub;
    if (DUP) r = r < rows_valid ? r : rows_valid - 1;
    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * LS + piece * VEC);
    // (non-temporal: every caller streams rows that a LATER launch reads -- the skip sum's r0, the skip data gradient's
    // dcs ...; round 4 A/B with the reduction's non-temporal loads: -22 us per step, most of it in the kernels that read
    // those rows next)
    if (DUP || r < rows_valid) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(gtile + (int64_t)r * grow_stride + piece * VEC));
  }
}

// ----------------------------------------------------------------------------------------------
// nonlinearities.  fp32 mode = libm-accurate (parity <= 1e-3 vs the oracle); bf16 mode = hardware
// exp2/rcp (their ~1 ulp error is far below the bf16 storage rounding).
// ----------------------------------------------------------------------------------------------
template <typename T> struct Math;
template <> struct Math<float> {
  static __device__ __forceinline__ float tanh_(float x) { return tanhf(x); }
  static __device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }
};
template <> struct Math<bf16_t> {
  static __device__ __forceinline__ float tanh_(float x) {
    float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // exp(2x)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  }
  // sigmoid on [-1, 1] only: its sole argument on this path is z = tanh(.) (ops.py:28,33).  Odd near-minimax
  // polynomial of sigmoid(x) - 1/2, max error 2.7e-6 -- 4 full-rate FMAs instead of exp + rcp (two quarter-rate
  // transcendentals): the gate recomputation was the VALU bottleneck of the skip-sum and skip-wgrad GEMMs.
  static __device__ __forceinline__ float sigmoid_(float x) {
    const float u = x * x;
    float t = fmaf(u, 0.00175846f, -0.02067844f);
    t = fmaf(u, t, 0.24998121f);
    return fmaf(x, t, 0.5f);
  }
};

// the reference's gate (ops.py:28-36, incl. the discarded-gate-conv behaviour of line 33):
//   z = tanh(f); c = z * sigmoid(z)
template <typename T> __device__ __forceinline__ float gate_of_z(float z) { return z * Math<T>::sigmoid_(z); }
// The gate of a whole fragment.  bf16: written on register pairs (v_pk_mul_f32 / v_pk_fma_f32: two lanes of fp32 per
// instruction, the same IEEE operations in the same order as gate_of_z<bf16_t>, so the results are bit-identical) -- 33
// VALU instructions per fragment.  Inside the skip sum's scheduled chunk loop hipcc's SLP pass gave up on the per-element
// form: 76 per fragment, half of them unpaired, and the VALU port -- not the matrix pipe -- bounded the loop.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void gate_frag(Frag<T>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.set(j, gate_of_z<T>(f.get(j)));
}
__device__ __forceinline__ uint32_t gate_pair(uint32_t zz) {      // two bf16 z in one register -> their two gated values
  const f32x2 z = {__builtin_bit_cast(float, zz << 16), __builtin_bit_cast(float, zz & 0xffff0000u)};
  const f32x2 u = z * z;
  f32x2 t = __builtin_elementwise_fma(u, f32x2{0.00175846f, 0.00175846f}, f32x2{-0.02067844f, -0.02067844f});
  t = __builtin_elementwise_fma(u, t, f32x2{0.24998121f, 0.24998121f});
  const f32x2 s = __builtin_elementwise_fma(z, t, f32x2{0.5f, 0.5f});
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(z * s, bf16x2));
}
template <> __device__ __forceinline__ void gate_frag<bf16_t>(Frag<bf16_t>& f) {
  u32x4 r = __builtin_bit_cast(u32x4, f.v);
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = gate_pair(r[i]);
  f.v = __builtin_bit_cast(bf16x8, r);
}
// d c / d f  = (s + z s (1-s)) * (1 - z^2)
template <typename T> __device__ __forceinline__ float dgate_df(float z) {
  float s = Math<T>::sigmoid_(z);
  return (s + z * s * (1.0f - s)) * (1.0f - z * z);
}
// bf16 mode: the same function as ONE polynomial on z in [-1, 1] (z is a tanh output).  With u = z^2 the even part of
// d c / d f is exactly (1 - u)/2 (sigmoid(-z) = 1 - sigmoid(z)) and the odd part is z r(u); r fitted to degree 4:
// max error 9e-7 (the sigmoid-polynomial form above: 2.5e-6), 7 FMAs instead of 10 operations -- this derivative is a
// quarter of the backward kernels' VALU work.
template <> __device__ __forceinline__ float dgate_df<bf16_t>(float z) {
  const float u = z * z;
  float r = fmaf(u, 0.00142598f, -0.01381972f);
  r = fmaf(u, r, 0.0957148f);
  r = fmaf(u, r, -0.5833199f);
  r = fmaf(u, r, 0.49999976f);
  return fmaf(z, r, fmaf(u, -0.5f, 0.5f));
}

// tanh of four values at once (bf16 mode): Math<bf16_t>::tanh_'s operations with the three that are not transcendental
// written on the vector, so that they become packed instructions (between two scalar-only transcendentals the compiler
// leaves them single: 3 of the 5 instructions per value, in a kernel that issues one instruction per ~5 cycles and wave)
__device__ __forceinline__ f32x4 tanh4_bf16(f32x4 x) {
  const f32x4 t = x * f32x4{2.8853900817779268f, 2.8853900817779268f, 2.8853900817779268f, 2.8853900817779268f};
  f32x4 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
  e = e + f32x4{1.0f, 1.0f, 1.0f, 1.0f};
  const f32x4 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1]), __builtin_amdgcn_rcpf(e[2]), __builtin_amdgcn_rcpf(e[3])};
  // 1 - 2 r, in the rounding of tanh_ (2.0f * r is exact, so the subtraction is the one rounding either way)
  return f32x4{1.0f, 1.0f, 1.0f, 1.0f} - f32x4{2.0f, 2.0f, 2.0f, 2.0f} * r;
}

// dgate_df of four values at once (bf16 mode): the same operations on two register pairs side by side.  Written per value
// the compiler pairs the lanes but schedules ONE Horner chain at a time (live ranges first: the kernels that use it run at
// 256 registers), and every link of a chain of packed fp32 needs a wait state behind the one before: 6 s_nop per pair,
// a third of the instruction slots of the chain.  Two chains interleaved need none.
__device__ __forceinline__ f32x4 dgate_df4(f32x4 z) {
  const f32x4 u = z * z;
  f32x4 r = __builtin_elementwise_fma(u, f32x4{0.00142598f, 0.00142598f, 0.00142598f, 0.00142598f}, f32x4{-0.01381972f, -0.01381972f, -0.01381972f, -0.01381972f});
  r = __builtin_elementwise_fma(u, r, f32x4{0.0957148f, 0.0957148f, 0.0957148f, 0.0957148f});
  r = __builtin_elementwise_fma(u, r, f32x4{-0.5833199f, -0.5833199f, -0.5833199f, -0.5833199f});
  r = __builtin_elementwise_fma(u, r, f32x4{0.49999976f, 0.49999976f, 0.49999976f, 0.49999976f});
  const f32x4 ev = __builtin_elementwise_fma(u, f32x4{-0.5f, -0.5f, -0.5f, -0.5f}, f32x4{0.5f, 0.5f, 0.5f, 0.5f});
  return __builtin_elementwise_fma(z, r, ev);
}

}  // namespace srwn
