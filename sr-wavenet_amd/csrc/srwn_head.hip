// The head of the softmax teacher, forward AND backward, in one launch (bf16, 256 skip channels, <= 256 classes):
//   r1      = relu(r0 . W1 + b1)                                   model.py:53-54
//   logits  = r1 . W2 + b2 ; loss_row = logsumexp(logits) - logits[target]        model.py:56 + softmax-CE (model.py:100-112)
//   dlogits = (softmax(logits) - onehot(target)) * grad_scale
//   da1     = (dlogits . W2^T) * (r1 > 0)                          autodiff of model.py:56, 54
//   dtotal  = (da1 . W1^T) * (r0 > 0)                              autodiff of model.py:53, 51
// Every product is row-local, so a wave takes 32 rows through all four of them: the accumulator tile of one product
// (channels on registers, time on lanes) is the B operand of the next (srwn_common.h), and only what the weight-gradient
// passes need later (r1, dlogits, da1, dtotal) is written.  The four 256x256 weight images (the last three packed in
// the accumulator's k order) stream through LDS as ONE pipeline of sixteen 32-KB chunks, LDS-DMA double-buffered, shared
// by the eight waves of the workgroup.  Replaces four launches (head 1x1, softmax head, two head data gradients:
// 232 us, 666 MB) -- same values up to the summation order inside an MFMA k-step.
// gfx950 (MI355X) only.
#include <cstdlib>
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

namespace {

__device__ __forceinline__ void glds16_untracked(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {       // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

typedef short i16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
// on packed bf16 pairs: relu (a negative bf16 is a negative int16; -0 -> +0), "is nonzero" as 0/1 per half (the pair is
// +0 or positive), and 0/1 -> 0/0xffff per half
__device__ __forceinline__ unsigned pk_relu(unsigned w) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), i16x2_t{0, 0}));
}
// (asm: the builtin forms are canonicalised into compare / select / permute chains, 4x the instructions)
__device__ __forceinline__ unsigned pk_nonzero(unsigned w) {
  unsigned d;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(w), "s"(0x00010001u));
  return d;
}
__device__ __forceinline__ unsigned pk_ones(unsigned b) {
  unsigned d;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(b), "s"(0xffffffffu));
  return d;
}

struct HcArgs {
  const void* r0;            // [rows, 256] relu'd skip sum
  const void* w[4];          // packed [8][16] images: W1 (natural k), W2, W2^T, W1^T (permuted k)
  const float* b1; const float* b2;
  const int32_t* targets; float* loss_partials; float grad_scale; int cout_valid;
  void* r1; void* dlogits; void* da1; void* dtotal;   // [rows, 256] each
  int64_t rows;
  unsigned long long* stamps;   // diagnostic builds only (srwn_debug_stamp_buffer)
  int safe_wait;                // SRWN_SAFE_WAIT: vmcnt(0) instead of the counted wait
  int stagger;                  // first-round workgroups start (blockIdx / 8) % 8 x stagger x ~1k cycles late
  int first_round;              // workgroups that start at once (= CUs)
};

constexpr int kHcWaves = 8;
constexpr int kHcBufs = 3;       // weight chunks in flight: the one being read + two on their way (an L2 round trip is longer than a step)
template <int V> struct IC { static constexpr int value = V; };

template <bool STAMP> struct HcStamper {
  unsigned long long* p; int n;
  __device__ __forceinline__ void operator()(int tag) {
    if (STAMP && p && n < 512) { p[n] = ((unsigned long long)tag << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull); ++n; }
  }
};

template <typename T, bool STAMP = false>
__global__ __launch_bounds__(64 * kHcWaves) void headchain_kernel(HcArgs a) {
  constexpr int MT = 8, KS = 16, KSC = 4, NCH = KS / KSC, C = 256;
  constexpr int FB = (int)sizeof(Frag<T>) * 64, CHUNK_B = MT * KSC * FB, PIECES = CHUNK_B / 1024;
  static_assert(PIECES % kHcWaves == 0, "chunk must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][CHUNK_B] weights | 8 row stages
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kHcWaves + wave;
  const int64_t r0row = tile * 32;
  const int64_t row = r0row + col;
  const bool valid = row < a.rows;
  const int64_t rowc = valid ? row : (a.rows - 1);
  const int rows_valid = (a.rows - r0row) < 32 ? (int)(a.rows - r0row) : 32;   // <= 0: idle wave (still in the barriers)
  T* rstage = reinterpret_cast<T*>(smem + kHcBufs * CHUNK_B) + wave * (32 * RowStage<T>::stride(64));
  float* lbias = reinterpret_cast<float*>(smem + kHcBufs * CHUNK_B + kHcWaves * 32 * RowStage<T>::stride(64) * sizeof(T));   // b1 | b2
  unsigned char* gbits = reinterpret_cast<unsigned char*>(lbias + 2 * C) + wave * (32 * 33);   // r0 > 0: [row][channel / 8], rows 33 B apart
  for (int i = threadIdx.x; i < 2 * C; i += 64 * kHcWaves) lbias[i] = i < C ? a.b1[i] : (i - C < a.cout_valid ? a.b2[i - C] : -1e30f);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  HcStamper<STAMP> stamp{nullptr, 0};
  if (STAMP && blockIdx.x == 0 && lane == 0 && wave < 2) stamp.p = a.stamps + wave * 512;
  stamp(1);
  // Every workgroup is load burst -> 16 chunk steps -> four store bursts, and the first round of them starts in the same
  // cycle on every CU: HBM and the matrix pipes took turns chip-wide (profiles/r03_w: the prologue's 128-KB tile 24.6 k
  // cycles, an epilogue's stores up to 12.5 k, of a workgroup's 98 k).  A start offset per CU smears the bursts.
  if (a.stagger > 0 && (int)blockIdx.x < a.first_round) {
    const int d = (int)((blockIdx.x >> 3) & 7) * a.stagger;
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(16);      // ~1k cycles each
  }

  auto stage = [&](int g, int buf) __attribute__((always_inline)) {      // chunk c = g % 4 of image g / 4 -> weight buffer `buf`
    const char* wbase = reinterpret_cast<const char*>(a.w[g / NCH]);
    const int c = g % NCH;
#pragma unroll
    for (int i = 0; i < PIECES / kHcWaves; ++i) {
      const int p = wave * (PIECES / kHcWaves) + i;
      const int mt = p / (KSC * FB / 1024), within = (p % (KSC * FB / 1024)) * 1024;
      const char* gsrc = wbase + ((size_t)mt * KS + (size_t)c * KSC) * FB + within + lane * 16;
      glds16_untracked(gsrc, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * CHUNK_B + mt * (KSC * FB) + within)));
    }
  };

  const int tgt = valid ? a.targets[row] : -1;
  // B operand of the running product: sixteen k-steps of 16 channels for this lane's row
  Frag<T> bfr[KS];
  stage(0, 0);
  if (kHcBufs > 2) stage(1, 1);
  // (whole 128-byte row pieces through the wave's row stage instead -- 8 lines per load instruction, not 32 -- measured the
  // same 97.5 / 97.9 us: the prologue is the chip-wide burst and its latency, not the address path)
  {
    const T* xr = reinterpret_cast<const T*>(a.r0) + rowc * C + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bfr[ks] = load_nat(xr + 16 * ks);
  }
  // r0 > 0 (the gate of the last product's output) as one bit per channel in LDS: the lane holds channels
  // 16ks + 8*half .. +8 of its row = byte 2ks + half; the last epilogue reads bytes in its own (row-piece) layout.
  // (Re-reading r0 there instead costs an HBM round trip per 64-channel pass: the rows have left L2 by then.)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const u32x4 wd = __builtin_bit_cast(u32x4, bfr[ks].v);
    unsigned b = 0u;
#pragma unroll
    for (int e = 3; e >= 0; --e) b = (b << 2) | pk_nonzero(pk_relu(wd[e]));     // bits 2e and 2e + 16
    gbits[col * 33 + 2 * ks + half] = (unsigned char)(b | (b >> 15));
  }
  f32x16 acc[MT];
  unsigned m1[4] = {0u, 0u, 0u, 0u};     // r1 > 0: word j = tiles 2j, 2j+1; packed pair idx = 8*(mt%2) + i -> bits 15-idx, 31-idx
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stamp(2);
  // channel of accumulator register q of tile mt = cn(mt, q) + 4*half: the lane half goes into bases and bounds once
  auto cn = [](int mt, int q) { return 32 * mt + crow(q, 0); };
  // accumulators start at the bias (as the separate launches do): 32 floats per 64 channels, fetched when the tile is free
  auto seed = [&](int mt, const float* nb) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = nb ? *reinterpret_cast<const f32x4*>(nb + 4 * half + 32 * mt + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][4 * g + r] = b[r];
    }
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) seed(mt, lbias);

  // Stage epilogue, 64 channels (two accumulator tiles) at a time: packed bf16 pairs = fn(j, idx, a0, a1) -> rows in
  // HBM (through `rd`, which sees each 16-byte row piece on its way out) and the next product's B fragments (k-steps
  // 4j..4j+3 come from tiles 2j, 2j+1); then the two tiles restart at the next product's bias.
  // (Rewriting all 128 accumulators in place and re-reading them needs a second set of 16-register tuples at the
  // stage boundary: ~150 spills in a 256-register kernel.)
  auto finish = [&](void* dst, auto fn, bool gated, const float* nextbias, bool frags) __attribute__((always_inline)) {
    constexpr int LS = RowStage<T>::stride(64);
    T* ytile = reinterpret_cast<T*>(dst) + (rows_valid > 0 ? r0row : 0) * C;
    const int rsub = lane >> 3, piece = lane & 7;             // 8 lanes x 16 B per 64-channel row piece
    const int rv = __builtin_amdgcn_readfirstlane(rows_valid);
    // gated: dtotal = (.) * (r0 > 0), the bits from the prologue meet the 16-byte row pieces on their way out
#pragma unroll
    for (int j = 0; j < MT / 2; ++j) {
      unsigned pk[2][8];
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          pk[m2][i] = fn(j, 8 * m2 + i, acc[2 * j + m2][2 * i], acc[2 * j + m2][2 * i + 1]);
      wave_lds_order();
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<u32x2*>(rstage + col * LS + 32 * m2 + 8 * g + 4 * half) = u32x2{pk[m2][2 * g], pk[m2][2 * g + 1]};
      wave_lds_order();
#pragma unroll
      for (int i = 0; i < 4; ++i) {                            // rows past the end repeat the last one (same bytes, same address)
        const int r = min(8 * i + rsub, rv - 1);
        u32x4 v = *reinterpret_cast<const u32x4*>(rstage + r * LS + piece * 8);
        if (gated) {
          const int b = gbits[max(r, 0) * 33 + 8 * j + piece];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] &= __builtin_amdgcn_perm((unsigned)__builtin_amdgcn_sbfe(b, 2 * e + 1, 1), (unsigned)__builtin_amdgcn_sbfe(b, 2 * e, 1), 0x05040100u);
        }
        if (rv > 0) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(ytile + r * C + 64 * j + piece * 8));
      }
      if (frags) {
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const u32x4 w{pk[m2][4 * h2], pk[m2][4 * h2 + 1], pk[m2][4 * h2 + 2], pk[m2][4 * h2 + 3]};
            bfr[2 * (2 * j + m2) + h2].v = __builtin_bit_cast(bf16x8, w);
          }
      }
      seed(2 * j, nextbias);
      seed(2 * j + 1, nextbias);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto step = [&](auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value;
    constexpr int s = g / NCH, c = g % NCH;
    constexpr int AHEAD = kHcBufs - 1;
    if (g + AHEAD < 4 * NCH) stage(g + AHEAD, (g + AHEAD) % kHcBufs);
    stamp(10);
    const Frag<T>* lw = reinterpret_cast<const Frag<T>*>(smem + (g % kHcBufs) * CHUNK_B) + lane;
    {
      // weight fragments two MFMAs ahead of their use (a ring of three fragments: 12 registers; a whole k-step ahead
      // would need 64 beside the 128 accumulators and the 64 registers of the running B operand)
      constexpr int NF = KSC * MT;
      Frag<T> af[3];
      af[0] = lw[0];                                  // fragment f = ks*MT + mt lives at (mt*KSC + ks)*64
      af[1] = lw[(1 * KSC) * 64];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int ks = f / MT, mt = f % MT;
        if (f + 2 < NF) af[(f + 2) % 3] = lw[(((f + 2) % MT) * KSC + (f + 2) / MT) * 64];
        __builtin_amdgcn_sched_barrier(0);
        mma(acc[mt], af[f % 3], bfr[c * KSC + ks]);
      }
    }
    stamp(11);
    // chunk g+1 has landed (and this wave's older stores are out); the pieces of chunk g+2 are the youngest in flight
    if constexpr (AHEAD > 1 && g + AHEAD < 4 * NCH) {
      if (a.safe_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES / kHcWaves) : "memory");
    }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(12);
    if constexpr (c == NCH - 1) {
      if constexpr (s == 0) {            // r1 = relu(.): on the packed pair (a negative bf16 is a negative int16), its > 0 bits
        finish(a.r1, [&](int j, int idx, float a0, float a1) {
          const unsigned w = pk_relu(pack2(a0, a1));
          m1[j] = (m1[j] << 1) | pk_nonzero(w);
          asm volatile("" : "+v"(m1[j]));     // built now: left to the optimiser the 64 terms are spilled and summed at the use
          return w; }, false, lbias + C, true);
      } else if constexpr (s == 1) {     // softmax cross-entropy over the class axis: registers x two lane halves
        auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };   // the 128 class compares are redone per pass,
        const int tgh = opaque(tgt - 4 * half);                               // not kept as 128 lane masks
        float m = -INFINITY, vt = 0.0f;  // (classes past cout_valid carry a bias of -1e30: they never win, exp() = 0)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 16; q += 2) {
            m = fmaxf(fmaxf(m, acc[mt][q]), acc[mt][q + 1]);
            vt = cn(mt, q) == tgh ? acc[mt][q] : vt;
            vt = cn(mt, q + 1) == tgh ? acc[mt][q + 1] : vt;
          }
        m = fmaxf(m, __shfl_xor(m, 32));
        vt += __shfl_xor(vt, 32);
        const float ml = -m * 1.4426950408889634f;
        float sum = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            acc[mt][q] = __builtin_amdgcn_exp2f(fmaf(acc[mt][q], 1.4426950408889634f, ml));
            sum += acc[mt][q];
          }
        sum += __shfl_xor(sum, 32);
        const float lse = m + __logf(sum);
        float loss = (valid && half == 0) ? (lse - vt) : 0.0f;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) loss += __shfl_xor(loss, off);
        if (lane == 0 && rows_valid > 0) a.loss_partials[tile] = loss;
        const float ps = a.grad_scale / sum, ngs = -a.grad_scale;
        const int tgc = opaque(tgt - 4 * half);
        finish(a.dlogits, [&](int j, int idx, float a0, float a1) {
          const int n = cn(2 * j + (idx >> 3), 2 * (idx & 7));
          return pack2(fmaf(a0, ps, n == tgc ? ngs : 0.0f), fmaf(a1, ps, n + 1 == tgc ? ngs : 0.0f)); }, false, nullptr, true);
      } else if constexpr (s == 2) {     // da1 = (.) * (r1 > 0)
        finish(a.da1, [&](int j, int idx, float a0, float a1) {
          const unsigned k0 = (unsigned)__builtin_amdgcn_sbfe((int)m1[j], 15 - idx, 1);     // 0 or ~0
          const unsigned k1 = (unsigned)__builtin_amdgcn_sbfe((int)m1[j], 31 - idx, 1);
          return pack2(__uint_as_float(__float_as_uint(a0) & k0), __uint_as_float(__float_as_uint(a1) & k1)); },
               false, nullptr, true);
      } else {                           // dtotal = (.) * (r0 > 0)
        finish(a.dtotal, [&](int j, int idx, float a0, float a1) { return pack2(a0, a1); }, true, nullptr, false);
      }
    }
    if constexpr (c == NCH - 1) stamp(13);
    __syncthreads();
    stamp(14);
  };
  step(IC<0>{});  step(IC<1>{});  step(IC<2>{});  step(IC<3>{});
  step(IC<4>{});  step(IC<5>{});  step(IC<6>{});  step(IC<7>{});
  step(IC<8>{});  step(IC<9>{});  step(IC<10>{}); step(IC<11>{});
  step(IC<12>{}); step(IC<13>{}); step(IC<14>{}); step(IC<15>{});
}

}  // namespace

static int num_cus_hc() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus;
}

extern "C" int srwn_head_chain(const void* r0, const void* w1, const void* w2_perm, const void* w2T_perm,
                               const void* w1T_perm, const float* b1, const float* b2, const int32_t* targets,
                               float* loss_partials, void* r1, void* dlogits, void* da1, void* dtotal, int32_t S,
                               int32_t cout_pad, int32_t cout_valid, int64_t rows, float grad_scale, int32_t dtype,
                               void* stream) {
  if (rows == 0) return 0;
  if (!r0 || !w1 || !w2_perm || !w2T_perm || !w1T_perm || !b1 || !b2 || !targets || !loss_partials || !r1 ||
      !dlogits || !da1 || !dtotal)
    return set_error(SRWN_E_NULL, "head_chain: null pointer");
  if (dtype != SRWN_BF16) return set_error(SRWN_E_UNSUPPORTED, "head_chain: built for bf16 (fp32 keeps the four launches)");
  if (S != 256 || cout_pad != 256) return set_error(SRWN_E_UNSUPPORTED, "head_chain: built for 256 skip channels and <= 256 classes (S=%d, cout_pad=%d)", S, cout_pad);
  if (rows < 0 || cout_valid < 1 || cout_valid > 256 || (rows + 31) / 32 / kHcWaves + 1 > 0x7fffffffLL)
    return set_error(SRWN_E_SHAPE, "head_chain: rows=%lld cout_valid=%d", (long long)rows, cout_valid);
  HcArgs a{r0, {w1, w2_perm, w2T_perm, w1T_perm}, b1, b2, targets, loss_partials, grad_scale, cout_valid,
           r1, dlogits, da1, dtotal, rows, debug_stamps(), safe_wait(), 0, 0};
  {
    static const int stg = [] { const char* e = getenv("SRWN_HC_STAGGER"); return e ? atoi(e) : 2; }();
    a.stagger = stg;
    a.first_round = num_cus_hc();
  }
  const int64_t tiles = (rows + 31) / 32;
  const size_t sh = kHcBufs * (size_t)(8 * 4 * 1024) + (size_t)kHcWaves * 32 * RowStage<bf16_t>::stride(64) * sizeof(bf16_t) + 2 * 256 * sizeof(float) + (size_t)kHcWaves * 32 * 33;
  auto kfn = headchain_kernel<bf16_t, false>;
  SRWN_DIAG_ONLY(if (a.stamps) kfn = headchain_kernel<bf16_t, true>;)
  hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  if (e != hipSuccess) return set_error((int)e, "head_chain: LDS %zu: %s", sh, hipGetErrorString(e));
  hipLaunchKernelGGL(kfn, dim3((unsigned)((tiles + kHcWaves - 1) / kHcWaves)), dim3(64 * kHcWaves), sh,
                     (hipStream_t)stream, a);
  return check_launch("head_chain");
}
