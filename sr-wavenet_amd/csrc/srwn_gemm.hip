// Row-streaming channels GEMM for 256-wide outputs (skip sum, head 1x1s, head data gradients, fused
// softmax head):  y^T[n][row] = epi( bias[n] + sum_k W[k][n] * pro(x[row][k]) ),  n < 256.
//
// Structure (one workgroup = 4 waves = 128 rows, all 256 output channels):
//  * each wave owns ONE 32-row column tile and ALL 8 row tiles of outputs (128 accumulator registers),
//    so every activation fragment is loaded (and its gate recomputed) exactly once chip-wide;
//  * the packed weights stream through LDS in chunks of KSC k-steps, double-buffered, filled with
//    global_load_lds_dwordx4 (lane-linear 1 KiB pieces, no VGPR round trip) while the previous chunk
//    is being consumed; one barrier per chunk (guide: "minimum 2-phase" schedule);
//  * activation fragments for chunk c+1 are fetched straight from HBM into registers while chunk c's
//    MFMAs run.
#include <cstdlib>
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

#define SRWN_EPI_SOFTMAX_CE 3

struct RgArgs {
  const void* x; int64_t x_row_stride; int64_t x_chunk_stride; int chunk_len; int ks_total;
  const void* wpack; const float* bias; void* y; int64_t y_row_stride; int cout_valid; int64_t rows;
  const void* aux; int64_t aux_row_stride;
  // softmax-CE epilogue
  const int32_t* targets; float* loss_partials; float* logits_out; float grad_scale;
  // TAPS: k-chunk c is time tap c -> row + c*tap_step inside the same clip of tap_T rows (zero outside);
  // frame_add (fp32 [clips*frames, frame_add_ld]) * frame_add_scale is added per row before the epilogue
  int tap_T; int tap_step; const float* frame_add; int64_t frame_add_ld; int frames; int pool; float frame_add_scale;
  int safe_wait;   // SRWN_SAFE_WAIT: vmcnt(0) instead of the counted wait
  unsigned long long* stamps;   // diagnostic instantiation only (srwn_debug_stamp_buffer): waves 0 and 4 of workgroup 0
};

// In-kernel time stamps (as in srwn_group.hip / srwn_head.hip): lane 0 of waves 0 and 4 of workgroup 0 -- the two waves of
// SIMD 0 -- 512 entries each (tools/rg_stamps.py).
#ifndef SRWN_RG_PIPE
#define SRWN_RG_PIPE 1
#endif
#ifndef SRWN_RG_PIPE_VALU
#define SRWN_RG_PIPE_VALU 6
#endif
constexpr bool kRgPipe = SRWN_RG_PIPE != 0;        // the gated skip sum gates chunk c+1 between the MFMAs of chunk c
constexpr int kRgPipeValu = SRWN_RG_PIPE_VALU;    // VALU instructions scheduled behind each of those MFMAs
template <bool STAMP> struct RgStamper {
  unsigned long long* p; int n;
  __device__ __forceinline__ void operator()(int tag) {
    if (STAMP && p && n < 512) { p[n] = ((unsigned long long)tag << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull); ++n; }
  }
};

template <typename T> __device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// LDS-DMA the compiler does not see (it treats a visible one as a pending write to any LDS address and answers with
// s_waitcnt vmcnt(0) -- which would also drain activation loads issued chunks ahead).  The kernel retires these pieces
// itself with a counted s_waitcnt before the chunk barrier.  M0 (LDS base of the DMA) is saved and restored.
__device__ __forceinline__ void glds16_untracked(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

// Waves per workgroup.  Every workgroup streams the WHOLE weight image through LDS, so the LDS-DMA volume is
// (rows / rows per workgroup) x image: with 4 waves (128 rows) the skip sum moved 1 GB of weights for 0.49 GB of
// activations and ran at the chip's LDS-DMA rate (~6.4 TB/s, MI355X_MICROARCH.md), not at its MFMA or HBM rate.
// Eight waves share each chunk (the same 8 waves per CU as two 4-wave workgroups: the registers allow no more).
// The softmax head and the NT = 2 experiment need more than 256 registers (one wave per SIMD): four waves there.
constexpr int rg_waves(int epi, int nt) { return (epi == SRWN_EPI_SOFTMAX_CE || nt == 2) ? 4 : 8; }

template <typename T, int MT, int KSC, int PRO, int EPI, int NT, bool TAPS = false, bool STAMP = false>
__global__ __launch_bounds__(64 * rg_waves(EPI, NT), (EPI == SRWN_EPI_SOFTMAX_CE || NT == 2) ? 1 : 2) void rowgemm_kernel(RgArgs a) {
  constexpr int kRgWaves = rg_waves(EPI, NT);
  RgStamper<STAMP> stamp{nullptr, 0};
  if (STAMP && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 3) == 0) stamp.p = a.stamps + (threadIdx.x >> 8) * 512;
  stamp(1);
  static_assert(MT % 2 == 0, "outputs are emitted in 64-channel groups");
  static_assert(EPI != SRWN_EPI_SOFTMAX_CE || NT == 1, "softmax epilogue holds one column tile");
  constexpr int FB = sizeof(Frag<T>) * 64;          // bytes of one fragment image (1 KiB bf16, 2 KiB f32)
  constexpr int CHUNK_B = MT * KSC * FB;            // bytes of one weight chunk in LDS
  constexpr int PIECES = CHUNK_B / 1024;            // 1-KiB glds pieces per chunk
  static_assert(PIECES % kRgWaves == 0, "chunk must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][CHUNK_B]; reused as row stages at the end

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int64_t tile0 = ((int64_t)blockIdx.x * kRgWaves + wave) * NT;   // first 32-row tile of this wave
  int64_t rowv[NT];
  bool valid[NT];
  int tclip[NT];   // (TAPS) time index of the row inside its clip
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    rowv[nt] = (tile0 + nt) * 32 + col;
    valid[nt] = rowv[nt] < a.rows;
    tclip[nt] = TAPS ? (int)((valid[nt] ? rowv[nt] : (a.rows - 1)) % a.tap_T) : 0;
  }
  const int nchunks = a.ks_total / KSC;
  const char* wbase = reinterpret_cast<const char*>(a.wpack);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // piece p of chunk c: fragment run (mt, byte offset) -> global address; LDS image is [mt][KSC][64] frags
  auto stage = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < PIECES / kRgWaves; ++i) {
      const int p = wave * (PIECES / kRgWaves) + i;
      const int mt = p / (KSC * FB / 1024), within = (p % (KSC * FB / 1024)) * 1024;
      const char* g = wbase + ((size_t)mt * a.ks_total + (size_t)c * KSC) * FB + within + lane * 16;
      const unsigned l = lds_base + (unsigned)(buf * CHUNK_B + mt * (KSC * FB) + within);
      glds16_untracked(g, __builtin_amdgcn_readfirstlane(l));
    }
  };
  // The load cursor: (source chunk, offset inside it, element offset) of the k-steps of the next chunk to request.
  // Chunks are requested in order, the last one again once the end is reached (a re-fetch that keeps the counted waits
  // fixed), so the cursor only ever steps by 16.  The division kg / chunk_len it replaces was 4 x 30 scalar instructions
  // at the head of every chunk, all waves at once, in a loop that is bound by the instructions a wave can issue (one per
  // ~5 cycles: in-kernel stamps of a one-wave-per-SIMD variant, profiles/r04_j_skipsum_stamps.txt): skip sum -5.6 us.
  // (chunk_len % 16 == 0: host check.)
  int cur_chunk = 0, cur_within = 0, cur_next = 0;
  int64_t cur_off = 0;      // = cur_chunk * x_chunk_stride + cur_within, in elements
  const int64_t chunk_step = a.x_chunk_stride - a.chunk_len + 16;
  auto cursor = [&](int (&chunk)[KSC], int (&within)[KSC], int64_t (&off)[KSC]) {
    int cc = cur_chunk, ww = cur_within;
    int64_t oo = cur_off;
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      chunk[ks] = cc; within[ks] = ww; off[ks] = oo;
      const bool wrap = ww + 16 >= a.chunk_len;
      ww = wrap ? 0 : ww + 16;
      oo += wrap ? chunk_step : (int64_t)16;
      cc += wrap ? 1 : 0;
    }
    if (cur_next + 1 < nchunks) { cur_chunk = cc; cur_within = ww; cur_off = oo; ++cur_next; }
  };
  // unconditional (clamped) activation loads; rows beyond the end produce values that are never stored
  auto load_b = [&](Frag<T> (&dst)[NT][KSC], bool (&ok)[NT][KSC]) {
    int chunks[KSC], withins[KSC];
    int64_t offs[KSC];
    cursor(chunks, withins, offs);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        const int chunk = chunks[ks], within = withins[ks];
        const int64_t rbase = valid[nt] ? rowv[nt] : (a.rows - 1);
        if (TAPS) {
          const int tp = tclip[nt] + chunk * a.tap_step;
          ok[nt][ks] = tp >= 0 && tp < a.tap_T;
          const T* p = reinterpret_cast<const T*>(a.x) + (rbase + (ok[nt][ks] ? chunk * a.tap_step : 0)) * a.x_row_stride +
                       within + 8 * half;
          dst[nt][ks] = load_nat(p);
        } else {
          ok[nt][ks] = true;
          const T* p = reinterpret_cast<const T*>(a.x) + offs[ks] + rbase * a.x_row_stride + 8 * half;
          dst[nt][ks] = load_nat(p);
        }
      }
  };

  f32x16 acc[MT][NT];
  const bool bias_al = (reinterpret_cast<size_t>(a.bias) & 15) == 0;
  auto init_acc = [&]() {       // accumulators start at the bias (four consecutive channels per register group)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n0 = 32 * mt + 8 * gq + 4 * half;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
          if (n0 + 3 < a.cout_valid && bias_al) bv = *reinterpret_cast<const f32x4*>(a.bias + n0);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (n0 + e < a.cout_valid) ? a.bias[n0 + e] : 0.0f;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt][4 * gq + e] = bv[e];
      }
  };
  constexpr bool PIPE = kRgPipe && PRO == SRWN_PRO_GATE && sizeof(T) == 2 && !TAPS && NT == 1;
  if (!PIPE) init_acc();

  // ---- the gated skip sum (bf16): the gate of chunk c+1 runs BETWEEN the MFMAs of chunk c.  In the loop below a chunk's
  // gate (VALU, ~250 instructions per wave) and its 32 MFMAs are a dependent chain inside a wave, so the two pipes only
  // overlap across the two waves of a SIMD -- and there the older wave takes the issue slots: profiles/r03_o (stamps):
  // wave 0 is through a chunk after 3 050 cycles and waits 1 900 at the barrier for wave 4, the matrix pipe busy 2 048 of
  // 5 300.  Three register sets: the one being multiplied (gated a chunk ago), the one being gated (arrived), the one in
  // flight.
  if constexpr (PIPE) {
    Frag<T> bX[KSC], bY[KSC], bZ[KSC];
    const int64_t rbase = valid[0] ? rowv[0] : (a.rows - 1);
    auto loadp = [&](Frag<T> (&dst)[KSC]) {
      int chunks[KSC], withins[KSC];
      int64_t offs[KSC];
      cursor(chunks, withins, offs);
      const T* rp = reinterpret_cast<const T*>(a.x) + rbase * a.x_row_stride + 8 * half;
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) dst[ks] = load_nat(rp + offs[ks]);
    };
    auto gate = [&](Frag<T>& f) { gate_frag<T>(f); };
    stage(0, 0);
    loadp(bX);
    loadp(bY);
    init_acc();                 // (behind the first requests: the bias comes from the L2 while they cross the HBM)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) gate(bX[ks]);
    auto chunkp = [&](int c, Frag<T> (&cur)[KSC], Frag<T> (&nxt)[KSC], Frag<T> (&ld)[KSC]) {
      stamp(10);
      // nxt (requested a chunk ago) has to be in before anything younger is issued: the compiler's counted wait for it
      // would otherwise cover this chunk's DMA pieces as well (it does not count them)
      if constexpr (KSC == 4) asm volatile("" : "+v"(nxt[0].v), "+v"(nxt[1].v), "+v"(nxt[2].v), "+v"(nxt[3].v));   // (one wait)
      else {
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) asm volatile("" : "+v"(nxt[ks].v));
      }
      stamp(11);
      if (c + 1 < nchunks) stage(c + 1, (c + 1) & 1);
      loadp(ld);
      stamp(12);
      const Frag<T>* lw = reinterpret_cast<const Frag<T>*>(smem + (c & 1) * CHUNK_B) + lane;
      Frag<T> af[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[0][mt] = lw[(mt * KSC) * 64];
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        if (ks + 1 < KSC) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) af[(ks + 1) & 1][mt] = lw[(mt * KSC + ks + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mma(acc[mt][0], af[ks & 1][mt], cur[ks]);
        gate(nxt[ks]);
        asm volatile("" : "+v"(nxt[ks].v));    // (pins the gate HERE: left alone it sinks to its use, behind the barrier)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {      // one MFMA, then a slice of the next chunk's gate in its shadow
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, kRgPipeValu, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (STAMP) { asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[MT - 1][0][15])); stamp(13); }
      if (a.safe_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KSC) : "memory");
      stamp(14);
      __syncthreads();
      stamp(15);
    };
    for (int c = 0; c < nchunks; c += 3) {
      chunkp(c, bX, bY, bZ);
      if (c + 1 < nchunks) chunkp(c + 1, bY, bZ, bX);
      if (c + 2 < nchunks) chunkp(c + 2, bZ, bX, bY);
    }
  } else {
  // Pipeline: weight chunk c+1 streams into the other LDS buffer (LDS-DMA from L2) while chunk c is consumed;
  // activation fragments are fetched from HBM TWO chunks ahead into a ring of two register sets -- one chunk of
  // MFMAs (~0.5 us) does not cover an HBM round trip under load, and with a one-chunk distance every chunk barrier
  // exposed the remainder.  Per chunk: exactly NT*KSC activation loads are issued after the DMA pieces, so
  // s_waitcnt vmcnt(NT*KSC) before the barrier retires the DMA (and everything older) and nothing newer.
  Frag<T> bcur[NT][KSC], bA[NT][KSC], bB[NT][KSC];
  bool okA[NT][KSC], okB[NT][KSC];
  stage(0, 0);
  load_b(bA, okA);
  load_b(bB, okB);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto chunk = [&](int c, Frag<T> (&bthis)[NT][KSC], bool (&okthis)[NT][KSC]) {
    stamp(10);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        bcur[nt][ks] = bthis[nt][ks];   // rows past the end are clamped re-reads: computed, never stored
        if (TAPS) bcur[nt][ks] = okthis[nt][ks] ? bcur[nt][ks] : zero_frag<T>();   // taps outside the clip
        if (PRO == SRWN_PRO_GATE) gate_frag<T>(bcur[nt][ks]);
      }
    if (STAMP) { asm volatile("" :: "v"(bcur[0][0].get(0)), "v"(bcur[0][KSC - 1].get(7))); stamp(11); }   // activations arrived, first gate values
    if (c + 1 < nchunks) stage(c + 1, (c + 1) & 1);
    load_b(bthis, okthis);   // (past the end: a re-fetch that keeps the count fixed)
    stamp(12);
    const Frag<T>* lw = reinterpret_cast<const Frag<T>*>(smem + (c & 1) * CHUNK_B) + lane;
    // weight fragments one k-step ahead in registers: left to itself hipcc emits ds_read -> s_waitcnt lgkmcnt(0) ->
    // v_mfma per fragment, i.e. one exposed LDS round trip per 32-cycle MFMA; the sched barriers keep the next
    // k-step's reads ahead of this k-step's MFMAs
    // (bf16 only: an fp32 fragment is 8 registers and the second set does not fit beside 128 accumulators)
    constexpr bool APF = sizeof(T) == 2 && !TAPS && NT == 1;
    if constexpr (APF) {
      Frag<T> af[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[0][mt] = lw[(mt * KSC) * 64];
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        if (ks + 1 < KSC) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) af[(ks + 1) & 1][mt] = lw[(mt * KSC + ks + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) mma(acc[mt][nt], af[ks & 1][mt], bcur[nt][ks]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const Frag<T> af = lw[(mt * KSC + ks) * 64];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) mma(acc[mt][nt], af, bcur[nt][ks]);
        }
    }
    if (STAMP) { asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[MT - 1][NT - 1][15])); stamp(13); }             // the chunk's MFMAs retired
    if (a.safe_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT * KSC) : "memory");
    stamp(14);
    __syncthreads();
    stamp(15);
  };
  {
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
      chunk(c, bA, okA);
      chunk(c + 1, bB, okB);
    }
    if (c < nchunks) chunk(c, bA, okA);
  }
  }   // !PIPE
  // the weight buffers are free now: each wave takes a private row stage ([32][64 + pad]) from them
  T* rstage = reinterpret_cast<T*>(smem) + wave * (32 * RowStage<T>::stride(64));

  // ------------------------------------------------------------------------------------ epilogues
  if (EPI == SRWN_EPI_SOFTMAX_CE) {
    const int64_t row = rowv[0];
    const bool vld = valid[0];
    const int tgt = vld ? a.targets[row] : -1;
    // The range and target compares (n < cout_valid, n == tgt) are loop-invariant across the three passes; hipcc
    // hoists all 256 of them into SGPR pairs, runs out and spills several hundred through v_writelane.  `opaque`
    // makes each pass recompute its own (2 VALU ops per element) and consume them at once.  One exp per logit: the
    // accumulators are overwritten by exp(logit - max) and scaled by grad_scale/sum for the gradient.
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    float m = -INFINITY, vt = 0.0f;
    {
      const int cv = opaque(a.cout_valid), tg = opaque(tgt);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = 32 * mt + crow(q, half);
          if (n < cv) m = fmaxf(m, acc[mt][0][q]);
          if (n == tg) vt = acc[mt][0][q];
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    vt += __shfl_xor(vt, 32);
    if (a.logits_out && vld) {
      float* lr = a.logits_out + row * (int64_t)a.cout_valid;
      const int cv = opaque(a.cout_valid);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = 32 * mt + crow(q, half);
          if (n < cv) lr[n] = acc[mt][0][q];
        }
    }
    float s = 0.0f;
    {
      const int cv = opaque(a.cout_valid);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = 32 * mt + crow(q, half);
          const float e = (n < cv) ? __expf(acc[mt][0][q] - m) : 0.0f;
          acc[mt][0][q] = e;
          s += e;
        }
    }
    s += __shfl_xor(s, 32);
    const float lse = m + __logf(s);
    float loss = (vld && half == 0) ? (lse - vt) : 0.0f;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) loss += __shfl_xor(loss, off);
    if (lane == 0 && tile0 * 32 < a.rows) a.loss_partials[tile0] = loss;
    if (a.y) {
      const float ps = a.grad_scale / s;   // probabilities * grad_scale
      const int64_t r0 = tile0 * 32;
      const int rows_valid = (a.rows - r0) < 32 ? (int)(a.rows - r0) : 32;
      T* ytile = reinterpret_cast<T*>(a.y) + (rows_valid > 0 ? r0 : 0) * a.y_row_stride;
#pragma unroll
      for (int j = 0; j < MT / 2; ++j) {
        const int tg = opaque(tgt);
        float v[2][16];
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int n = 32 * (2 * j + m2) + crow(q, half);
            v[m2][q] = fmaf(acc[2 * j + m2][0][q], ps, n == tg ? -a.grad_scale : 0.0f);
          }
        store_rows_via_lds<T, 2>(rstage, ytile + 64 * j, a.y_row_stride, v, rows_valid, lane);
      }
    }
    return;
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int64_t r0 = (tile0 + nt) * 32;
    const int rows_valid = (a.rows - r0) < 32 ? (int)(a.rows - r0) : 32;   // <= 0 for idle tiles
    T* ytile = reinterpret_cast<T*>(a.y) + (rows_valid > 0 ? r0 : 0) * a.y_row_stride;
    const T* arow = (EPI == SRWN_EPI_MASK)
                        ? reinterpret_cast<const T*>(a.aux) + (valid[nt] ? rowv[nt] : 0) * a.aux_row_stride
                        : nullptr;
    const float* frow = nullptr;
    if (TAPS && a.frame_add) {
      const int64_t rr = valid[nt] ? rowv[nt] : 0;
      frow = a.frame_add + ((rr / a.tap_T) * a.frames + (tclip[nt] / a.pool)) * a.frame_add_ld;
    }
#pragma unroll
    for (int j = 0; j < MT / 2; ++j) {
      if (64 * j >= a.cout_valid) continue;   // cout_valid is a multiple of 64 on this path (host check)
      float v[2][16];
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 mk = {1.f, 1.f, 1.f, 1.f};
          if (EPI == SRWN_EPI_MASK) mk = load4(arow + 32 * (2 * j + m2) + 8 * g + 4 * half);
          f32x4 fa = {0.f, 0.f, 0.f, 0.f};
          if (TAPS && frow) fa = *reinterpret_cast<const f32x4*>(frow + 32 * (2 * j + m2) + 8 * g + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[2 * j + m2][nt][4 * g + e];
            if (TAPS) x = fmaf(fa[e], a.frame_add_scale, x);
            if (EPI == SRWN_EPI_RELU) x = fmaxf(x, 0.0f);
            if (EPI == SRWN_EPI_MASK) x = (mk[e] > 0.0f) ? x : 0.0f;
            v[m2][4 * g + e] = x;
          }
        }
      store_rows_via_lds<T, 2>(rstage, ytile + 64 * j, a.y_row_stride, v, rows_valid, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Output-streaming GEMM: the skip-path data gradient of EVERY layer in one launch
//   dcs[l][row][n] = sum_s dtotal[row][s] * Ws_l[n][s]          (autodiff of ops.py:44 for all l)
// Each wave keeps its 32 rows of dtotal as B fragments in registers for the whole kernel (loaded
// once), the per-layer weight images [R/32][S/16] stream through LDS (LDS-DMA, double-buffered in
// bf16 mode), and every layer's [32 x R] result leaves as whole rows.  This moves 2*R*S flop/sample
// per layer out of the latency-bound per-layer backward kernels into one MFMA-dense pass.
// ------------------------------------------------------------------------------------------
struct CgArgs {
  const void* x; int64_t x_row_stride; const void* wpack; void* y; int64_t y_layer_stride; int nlayers; int64_t rows;
  int safe_wait;
};

// NW waves per workgroup share each layer's weight image (every workgroup streams all of them: 8 waves halve that
// LDS-DMA volume, which at 4 waves was twice the size of the output).
template <typename T, int RT, int KSS, int NBUF, int NW>
__global__ __launch_bounds__(64 * NW) void colgemm_kernel(CgArgs a) {
  constexpr int R = 32 * RT;
  constexpr int FB = sizeof(Frag<T>) * 64;
  constexpr int CHUNK_B = RT * KSS * FB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][CHUNK_B] weights | NW row stages
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  T* stage = reinterpret_cast<T*>(smem + NBUF * CHUNK_B) + wave * (32 * RowStage<T>::stride(R));
  const int64_t row0 = ((int64_t)blockIdx.x * NW + wave) * 32;
  const int64_t row = row0 + col;
  const bool valid = row < a.rows;
  const int rows_valid = (a.rows - row0) < 32 ? (int)(a.rows - row0) : 32;   // may be <= 0 for idle waves
  const char* wbase = reinterpret_cast<const char*>(a.wpack);

  // bf16: the weight DMA is issued from inline asm and retired by a counted wait that leaves the layer's four row
  // stores in flight.  (__syncthreads() carries s_waitcnt vmcnt(0): every layer then waited for its 4 KB of stores to
  // be acknowledged before the next layer's MFMAs could start -- 30 store round trips per wave.)
  constexpr bool COUNTED = sizeof(T) == 2 && NBUF == 2;
  constexpr int PIECES = CHUNK_B / 1024;
  static_assert(!COUNTED || PIECES % NW == 0, "weight image must split evenly over the waves");
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](int l, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PIECES / NW; ++i) {
      const int p = wave * (PIECES / NW) + i;
      glds16_untracked(wbase + (size_t)l * CHUNK_B + (size_t)p * 1024 + lane * 16,
                       __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * CHUNK_B + p * 1024)));
    }
  };
  if constexpr (COUNTED) dma(0, 0); else lds_dma_copy(wbase, smem, CHUNK_B, wave, lane, NW);
  Frag<T> bf[KSS];
  {
    const T* p = reinterpret_cast<const T*>(a.x) + (valid ? row : 0) * a.x_row_stride + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks) bf[ks] = load_nat(p + 16 * ks);
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks) bf[ks] = valid ? bf[ks] : zero_frag<T>();
  }
  if constexpr (COUNTED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int l = 0; l < a.nlayers; ++l) {
    const int buf = (NBUF == 2) ? (l & 1) : 0;
    if (NBUF == 2 && l + 1 < a.nlayers) {
      if constexpr (COUNTED) dma(l + 1, (l + 1) & 1);
      else lds_dma_copy(wbase + (size_t)(l + 1) * CHUNK_B, smem + ((l + 1) & 1) * CHUNK_B, CHUNK_B, wave, lane, NW);
    }
    const Frag<T>* lw = reinterpret_cast<const Frag<T>*>(smem + buf * CHUNK_B) + lane;
    f32x16 acc[RT];
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][q] = 0.0f;
    // weight fragments four k-steps ahead of their MFMAs (see rowgemm_kernel)
    constexpr int G = 4;
    Frag<T> af[2][G][RT];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int mt = 0; mt < RT; ++mt) af[0][j][mt] = lw[(mt * KSS + j) * 64];
#pragma unroll
    for (int k0 = 0; k0 < KSS; k0 += G) {
      if (k0 + G < KSS) {
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) af[((k0 / G) + 1) & 1][j][mt] = lw[(mt * KSS + k0 + G + j) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int mt = 0; mt < RT; ++mt) mma(acc[mt], af[(k0 / G) & 1][j][mt], bf[k0 + j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    float v[RT][16];
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) v[mt][q] = acc[mt][q];
    T* ytile = reinterpret_cast<T*>(a.y) + (int64_t)l * a.y_layer_stride + (rows_valid > 0 ? row0 : 0) * R;
    if constexpr (COUNTED) {
      constexpr int STORES = 32 / (64 / (R / RowStage<T>::VEC));    // store instructions of one tile (whole rows, 16 B per lane)
      if (rows_valid > 0) {                                         // (wave-uniform)
        store_rows_via_lds<T, RT, true>(stage, ytile, R, v, rows_valid, lane);
        if (a.safe_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STORES) : "memory");   // next image landed, older stores out; these fly on
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      store_rows_via_lds<T, RT>(stage, ytile, R, v, rows_valid, lane);
      __syncthreads();
    }
    if (NBUF == 1 && l + 1 < a.nlayers) {
      lds_dma_copy(wbase + (size_t)(l + 1) * CHUNK_B, smem, CHUNK_B, wave, lane, NW);
      __syncthreads();
    }
  }
}

template <typename T, int RT, int KSS, int NBUF>
static int launch_cg(const CgArgs& a, hipStream_t st) {
  constexpr int R = 32 * RT;
  constexpr int NW = (sizeof(T) == 2) ? 8 : 4;
  const size_t sh = (size_t)NBUF * RT * KSS * sizeof(Frag<T>) * 64 + (size_t)NW * 32 * RowStage<T>::stride(R) * sizeof(T);
  auto kfn = colgemm_kernel<T, RT, KSS, NBUF, NW>;
  hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  if (e != hipSuccess) return set_error((int)e, "skip_dgrad_all: LDS %zu: %s", sh, hipGetErrorString(e));
  hipLaunchKernelGGL(kfn, dim3((unsigned)((a.rows + 32 * NW - 1) / (32 * NW))), dim3(64 * NW), sh, st, a);
  return check_launch("skip_dgrad_all");
}

extern "C" int srwn_skip_dgrad_all(const void* dtotal, const void* wskipT_all, void* dcs, int64_t dcs_layer_stride,
                                   int32_t nlayers, int64_t rows, int32_t R, int32_t S, int32_t dtype, void* stream) {
  if (rows == 0 || nlayers == 0) return 0;
  if (!dtotal || !wskipT_all || !dcs) return set_error(SRWN_E_NULL, "skip_dgrad_all: null pointer");
  if (rows < 0 || nlayers < 0) return set_error(SRWN_E_SHAPE, "skip_dgrad_all: rows=%lld layers=%d", (long long)rows, nlayers);
  CgArgs a{dtotal, S, wskipT_all, dcs, dcs_layer_stride, nlayers, rows, safe_wait()};
  hipStream_t st = (hipStream_t)stream;
  if (R == 64 && S == 256) {
    if (dtype == SRWN_BF16) return launch_cg<bf16_t, 2, 16, 2>(a, st);
    if (dtype == SRWN_F32) return launch_cg<float, 2, 16, 1>(a, st);
    return set_error(SRWN_E_DTYPE, "skip_dgrad_all: dtype %d", dtype);
  }
  if (R == 32 && S == 128) {
    if (dtype == SRWN_BF16) return launch_cg<bf16_t, 1, 8, 2>(a, st);
    if (dtype == SRWN_F32) return launch_cg<float, 1, 8, 2>(a, st);
    return set_error(SRWN_E_DTYPE, "skip_dgrad_all: dtype %d", dtype);
  }
  return set_error(SRWN_E_UNSUPPORTED, "skip_dgrad_all: built for (R,S) = (64,256) and (32,128), got (%d,%d)", R, S);
}

namespace srwn {

template <typename T, int MT, int KSC, int NT>
static int launch_rg(const RgArgs& a, int pro, int epi, hipStream_t st) {
  constexpr int CHUNK_B = MT * KSC * (int)sizeof(Frag<T>) * 64;
  // the weight double buffer is reused as the waves' private row stages in the epilogue: take the larger of the two
  constexpr int NTS = 1;   // softmax epilogue always one column tile per wave
  const int nw = rg_waves(epi, (epi == SRWN_EPI_SOFTMAX_CE) ? NTS : NT);
  const size_t stage_b = (size_t)nw * 32 * RowStage<T>::stride(64) * sizeof(T);
  const size_t sh = 2 * (size_t)CHUNK_B > stage_b ? 2 * (size_t)CHUNK_B : stage_b;
  const int rows_per_block = 32 * nw * ((epi == SRWN_EPI_SOFTMAX_CE) ? NTS : NT);
  dim3 grid((unsigned)((a.rows + rows_per_block - 1) / rows_per_block)), block(64 * nw);
#define SRWN_RG(P, E)                                                                                        \
  if (pro == P && epi == E) {                                                                                \
    auto kfn = rowgemm_kernel<T, MT, KSC, P, E, (E == SRWN_EPI_SOFTMAX_CE) ? 1 : NT>;                        \
    if constexpr (sizeof(T) == 2 && P == SRWN_PRO_GATE && E == SRWN_EPI_RELU && NT == 1 && MT == 8) {         \
      SRWN_DIAG_ONLY(if (a.stamps) kfn = rowgemm_kernel<T, MT, KSC, P, E, NT, false, true>;)   /* diagnostic build: the skip sum */ \
    }                                                                                                        \
    if (sh > 32768) {                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
      if (e != hipSuccess) return set_error((int)e, "rowgemm: LDS %zu: %s", sh, hipGetErrorString(e));       \
    }                                                                                                        \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, a);                                                         \
    return check_launch("rowgemm");                                                                          \
  }
  SRWN_RG(SRWN_PRO_NONE, SRWN_EPI_NONE)
  SRWN_RG(SRWN_PRO_NONE, SRWN_EPI_RELU)
  SRWN_RG(SRWN_PRO_NONE, SRWN_EPI_MASK)
  SRWN_RG(SRWN_PRO_GATE, SRWN_EPI_NONE)
  SRWN_RG(SRWN_PRO_GATE, SRWN_EPI_RELU)
  SRWN_RG(SRWN_PRO_NONE, SRWN_EPI_SOFTMAX_CE)
#undef SRWN_RG
  return set_error(SRWN_E_UNSUPPORTED, "rowgemm: pro %d / epi %d combination not built", pro, epi);
}

template <typename T, int MT, int KSC>
static int launch_rg_taps(const RgArgs& a, int epi, hipStream_t st) {
  constexpr int CHUNK_B = MT * KSC * (int)sizeof(Frag<T>) * 64;
  // the weight double buffer is reused as the waves' private row stages in the epilogue: take the larger of the two
  constexpr int kRgWaves = rg_waves(SRWN_EPI_NONE, 1);
  const size_t stage_b = (size_t)kRgWaves * 32 * RowStage<T>::stride(64) * sizeof(T);
  const size_t sh = 2 * (size_t)CHUNK_B > stage_b ? 2 * (size_t)CHUNK_B : stage_b;
  dim3 grid((unsigned)((a.rows + 32 * kRgWaves - 1) / (32 * kRgWaves))), block(64 * kRgWaves);
#define SRWN_RGT(E)                                                                                          \
  if (epi == E) {                                                                                            \
    auto kfn = rowgemm_kernel<T, MT, KSC, SRWN_PRO_NONE, E, 1, true>;                                        \
    if (sh > 32768) {                                                                                        \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
      if (e != hipSuccess) return set_error((int)e, "tap_linear: LDS %zu: %s", sh, hipGetErrorString(e));    \
    }                                                                                                        \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, a);                                                         \
    return check_launch("tap_linear");                                                                       \
  }
  SRWN_RGT(SRWN_EPI_NONE)
  SRWN_RGT(SRWN_EPI_RELU)
  SRWN_RGT(SRWN_EPI_MASK)
#undef SRWN_RGT
  return set_error(SRWN_E_UNSUPPORTED, "tap_linear: epilogue %d not built", epi);
}

// Returns 1 if the shape is served by the row-streaming kernel (then *rc holds the launch result).
int rowgemm_dispatch(const void* x, int64_t x_row_stride, int64_t x_chunk_stride, int chunk_len, int Cin,
                     const void* wpack, const float* bias, void* y, int64_t y_row_stride, int cout_pad,
                     int cout_valid, int64_t rows, const void* aux, int64_t aux_row_stride, const int32_t* targets,
                     float* loss_partials, float* logits_out, float grad_scale, int pro, int epi, int dtype,
                     hipStream_t st, int* rc) {
  if ((cout_pad != 256 && cout_pad != 128) || (Cin % 64) != 0 || rows < 1 || chunk_len < 16 || (chunk_len % 16) != 0) return 0;
  if (epi != SRWN_EPI_SOFTMAX_CE && (cout_valid % 64) != 0) return 0;
  if (cout_pad == 128) {   // 128-wide products (the reference scripts' skip_channels=128): 4 row tiles per wave
    if (epi == SRWN_EPI_SOFTMAX_CE) return 0;
    RgArgs a4{x, x_row_stride, x_chunk_stride, chunk_len, Cin / 16, wpack, bias, y, y_row_stride, cout_valid, rows,
              aux, aux_row_stride, targets, loss_partials, logits_out, grad_scale, 0, 0, nullptr, 0, 0, 1, 0.0f, safe_wait(), nullptr};
    if (dtype == SRWN_BF16) *rc = launch_rg<bf16_t, 4, 4, 1>(a4, pro, epi, st);
    else if (dtype == SRWN_F32) *rc = launch_rg<float, 4, 2, 1>(a4, pro, epi, st);
    else return 0;
    return 1;
  }
  RgArgs a{x, x_row_stride, x_chunk_stride, chunk_len, Cin / 16, wpack, bias, y, y_row_stride, cout_valid, rows,
           aux, aux_row_stride, targets, loss_partials, logits_out, grad_scale, 0, 0, nullptr, 0, 0, 1, 0.0f, safe_wait(), debug_stamps()};
  if (dtype == SRWN_BF16) *rc = launch_rg<bf16_t, 8, 4, 1>(a, pro, epi, st);   // (two row tiles per wave measured slower: DESIGN.md 5)
  else if (dtype == SRWN_F32) *rc = launch_rg<float, 8, 2, 1>(a, pro, epi, st);
  else return 0;
  return 1;
}

}  // namespace srwn

// ------------------------------------------------------------------------------------------
// time-tap GEMM (the non-causal K=2 convolutions of ResidualDilationLayerNC, ops.py:48-58, their data
// gradients, and the 1x1s around them), 128 or 256 output channels:
//   y[row][n] = epi( bias[n] + frame_add[clip*frames + t/pool][n]*scale
//                    + sum_tap sum_i x[row + tap*tap_step][i] * W[tap*Cin + i][n] ),   taps outside the clip = 0
// ------------------------------------------------------------------------------------------
extern "C" int srwn_tap_linear(const void* x, int64_t x_row_stride, int32_t ntaps, int32_t tap_step, int32_t T,
                               int32_t Cin, const void* wpack, const float* bias, void* y, int64_t y_row_stride,
                               int32_t cout, int64_t rows, const void* aux, int64_t aux_row_stride,
                               const float* frame_add, int64_t frame_add_ld, int32_t frames, int32_t pool_stride,
                               float frame_add_scale, int32_t epi, int32_t dtype, void* stream) {
  if (rows == 0) return 0;
  if (!x || !wpack || !y) return set_error(SRWN_E_NULL, "tap_linear: null pointer");
  if (epi == SRWN_EPI_MASK && !aux) return set_error(SRWN_E_NULL, "tap_linear: EPI_MASK needs aux");
  if (rows < 0 || T < 1 || rows % T || ntaps < 1 || Cin < 64 || Cin % 64 || (cout != 128 && cout != 256) ||
      (frame_add && (frames < 1 || pool_stride < 1 || (int64_t)frames * pool_stride < T || frame_add_ld < cout)))
    return set_error(SRWN_E_SHAPE, "tap_linear: rows=%lld T=%d taps=%d Cin=%d cout=%d frames=%d pool=%d",
                     (long long)rows, T, ntaps, Cin, cout, frames, pool_stride);
  RgArgs a{x, x_row_stride, 0, Cin, ntaps * Cin / 16, wpack, bias, y, y_row_stride, cout, rows, aux, aux_row_stride,
           nullptr, nullptr, nullptr, 0.0f, T, tap_step, frame_add, frame_add_ld, frames, pool_stride, frame_add_scale,
           safe_wait(), nullptr};
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SRWN_BF16) return cout == 128 ? launch_rg_taps<bf16_t, 4, 4>(a, epi, st) : launch_rg_taps<bf16_t, 8, 4>(a, epi, st);
  if (dtype == SRWN_F32) return cout == 128 ? launch_rg_taps<float, 4, 2>(a, epi, st) : launch_rg_taps<float, 8, 2>(a, epi, st);
  return set_error(SRWN_E_DTYPE, "tap_linear: dtype %d", dtype);
}

