// Multi-layer ("group") kernels of the residual stack: several consecutive ResidualDilationLayers (ops.py:23-46,
// stacked by model.py:42-47 / 176-189 / 428-453) per launch, activations carried between layers in LDS instead of HBM.
// gfx950 (MI355X) only; MFMA orientation and lane maps: srwn_common.h.
//
// Time decomposition.  A group is a run of layers whose dilations are all multiples of a common stride `st`
// (st = gcd; d_g = st * sub_g).  Positions t = j*st + r of one residue class r form a sub-sequence on which the
// group's layers act with the small dilations sub_g, independently of every other residue class.  So
//   {1,2,4,8,16}      -> st = 1,  sub = 1,2,4,8,16   (32 segments of 500 steps per 16000-step clip)
//   {32,...,512}      -> st = 32, sub = 1,2,4,8,16   (32 residue classes of 500 steps per clip)
// are the same program: a workgroup owns one SEGMENT = (clip b, residue r, positions [j0, j0+W)) plus a halo of
// H = sum(sub_g) <= 31 positions on the causal side, which it recomputes instead of exchanging with a neighbour.
// The only difference is the HBM row stride (a row = R channels of one time step = one or two whole cache lines).
//
// Forward (group_fwd_kernel): 8 waves; the segment's rows live in one LDS image [rows][R] (padded rows); per layer
//   tap(t - sub) fragments of every owned 32-row tile are read -> barrier -> conv MFMAs, tanh, gate, 1x1 residual
//   exactly as layer_fwd_kernel does them (same MFMA order: results are bit-identical to the per-layer path) ->
//   z and the new x go through the tile's own image rows to HBM as whole rows -> barrier.  Weights of layer g+1
//   stream into the second weight buffer while layer g computes.
#include <cmath>
#include <cstdlib>
#include <vector>
#include "srwn_common.h"
#include "srwn_group.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;
using namespace srwn::grp;

namespace {

struct GroupFwdArgs {
  int safe_wait;        // SRWN_SAFE_WAIT: vmcnt(0) instead of the counted waits
  const void* x0;                 // input of the group's first layer [B,T,R]
  void* x_out;                    // layer g's output at x_out + g*layer_stride (elements)
  void* z_out;                    // layer g's z at z_out + g*layer_stride
  int64_t layer_stride;
  const void* wconv[kMaxGroup];   // packed conv images (natural k order), as srwn_residual_layer_fwd takes them
  const void* wres[kMaxGroup];    // packed 1x1 residual images (permuted k order)
  const float* bias_f[kMaxGroup];
  const float* bias_r[kMaxGroup];
  const void* cond[kMaxGroup];    // conditioning bias of the layer ABOVE layer g (added to what layer g stores), or null
  int cond_frames, pool, cond_stride;
  int sub[kMaxGroup];             // dilation / st
  int nl, st, Tlen, B;
  int W, H, NT;                   // positions per segment, halo positions, 32-row tiles per segment image
  int nsub;                       // segments per sub-sequence
  int nseg;                       // B * st * nsub
  unsigned long long* stamps;     // diagnostic builds only (srwn_debug_stamp_buffer): s_memtime stamps of workgroup 0
  // weight-gradient tiles (srwn_group.h), WT instantiations only: the transposed layer input x_g and gate output c_g of
  // the positions each segment owns, layer g at xT / cT + g*wt_stride, tile (seg, k) at ((seg*KT)+k)*R*32
  void* xT; void* cT; int64_t wt_stride; int KT;
  int store_inner_x;              // WT: 0 = only the group's top layer stores its output rows (nothing reads the inner ones)
  // input conv fused in (srwn_residual_group_fwd_ic; x0 == null): the group's input rows are computed where they would be
  // loaded -- x0[b,t,c] = ic_b[c] + ic_w[0][c] audio[b, t-1-shift] + ic_w[1][c] audio[b, t-shift] (model.py:40 / 172-173,
  // RightShift folded into the taps, zeros before the clip) -- instead of a launch that writes them and a read that
  // fetches them back (16.4 MB each way for config 2)
  const float* ic_audio; const float* ic_w; const float* ic_b; int ic_shift;
};

// In-kernel time stamps (MI355X guide, "In-kernel stamps"): lane 0 of waves 0 and 1 of workgroup 0 append the shader
// clock to a buffer no other code reads.  Compiled in only when a buffer was registered (STAMP instantiation).
template <bool STAMP> struct Stamper {
  unsigned long long* p; int n;
  __device__ __forceinline__ void operator()(int) {}
};

// Which segment a workgroup takes.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, and its
// L2), neighbouring segments of a clip share their halo rows: with the identity map a halo row is fetched into two L2s.
// XCD k takes a contiguous run of segments (speed only -- any bijection is correct; round 4 A/B: -3..-5 us per step).
__device__ __forceinline__ int xcd_segment(int blk, int nseg) {
  if ((nseg & 7) == 0) return (blk & 7) * (nseg >> 3) + (blk >> 3);
  return blk;
}
template <> struct Stamper<true> {
  unsigned long long* p; int n;
  __device__ __forceinline__ void operator()(int tag) {
    if (p && n < 512) { p[n] = ((unsigned long long)tag << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull); ++n; }   // 512 per wave: the registered buffer holds 2 x 512
  }
};

template <typename T, int RT, bool COND, int MAXT, int NWB, int NWV = 8, bool WDMA = true, bool STAMP = false, bool WT = false, bool IC = false>
__global__ __launch_bounds__(64 * NWV) void group_fwd_kernel(GroupFwdArgs a) {
  static_assert(!IC || (WT && !COND), "input conv fused in: the unconditioned weight-gradient-tile kernels only");
  constexpr int R = 32 * RT, K = 2, KS = R / 16;
  constexpr int NCONV = RT * K * KS, NRES = RT * KS, NW = NCONV + NRES;   // weight fragments per layer
  constexpr int LS = RowStage<T>::stride(R), VEC = RowStage<T>::VEC;
  constexpr int LPR = R / VEC, RPI = 64 / LPR, NI = 32 / RPI;             // whole-row access: lanes per row, rows per instr
  constexpr int WBYTES = NW * 64 * (int)sizeof(Frag<T>);
  constexpr int WPIECES = WBYTES / 16, CPIECES = NCONV * 64 * (int)sizeof(Frag<T>) / 16;
  typedef typename Raw4g<T>::type raw4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<T>* wbuf = reinterpret_cast<Frag<T>*>(smem);                       // [NWB][NW*64]
  float* bbuf = reinterpret_cast<float*>(smem + (size_t)NWB * WBYTES);    // [NWB][2R]: bias_f | bias_r
  T* img = reinterpret_cast<T*>(smem + (size_t)NWB * WBYTES + (size_t)NWB * 2 * R * 4);   // [NT*32][LS]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int rsub = lane / LPR, piece = lane % LPR;
  Stamper<STAMP> stamp{nullptr, 0};
  if (STAMP && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && wave < 8) stamp.p = a.stamps + (wave >> 2) * 512;   // waves 0 and 4: the two waves of SIMD 0
  stamp(1);

  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  f32x4 breg;
  constexpr int WPT = WDMA ? 1 : (WPIECES + 64 * NWV - 1) / (64 * NWV);
  f32x4 wreg[WPT];
  auto wload = [&](int g, int buf) {   // layer g's [conv | res] images -> weight buffer `buf` by LDS-DMA; biases -> registers
    if (WDMA) {
      dma_image(a.wconv[g], lds_base + buf * WBYTES, CPIECES * 16, wave, lane, NWV);
      dma_image(a.wres[g], lds_base + buf * WBYTES + CPIECES * 16, (WPIECES - CPIECES) * 16, wave, lane, NWV);
    } else {   // through registers: the compiler tracks these loads itself
      const f32x4* pc = reinterpret_cast<const f32x4*>(a.wconv[g]);
      const f32x4* pr = reinterpret_cast<const f32x4*>(a.wres[g]);
#pragma unroll
      for (int v = 0; v < WPT; ++v) {
        int p = tid + v * 64 * NWV;
        p = p < WPIECES ? p : WPIECES - 1;
        wreg[v] = p < CPIECES ? pc[p] : pr[p - CPIECES];
      }
    }
    if (tid < 2 * R / 4) {
      const int c = 4 * tid;
      breg = c < R ? *reinterpret_cast<const f32x4*>(a.bias_f[g] + c) : *reinterpret_cast<const f32x4*>(a.bias_r[g] + c - R);
    }
  };
  // retire the DMA, park the biases.  `younger` = tiles whose 2*NI row stores this wave has issued SINCE the DMA:
  // vmcnt counts in issue order, so leaving that many operations outstanding retires the DMA without waiting for
  // the stores to be acknowledged.
  auto wstore = [&](int buf, int younger, bool xrows) {
    constexpr int STP = 2 * NI + (WT ? R / 16 : 0);   // row stores of z and x (+ the c tile's pieces) per stored tile
    constexpr int STZ = NI + (WT ? R / 16 : 0);       // ... of a layer that does not store its x rows (WT, inner layers)
    static_assert(MAXT <= 3 && 3 * STP < 64, "vmcnt immediates");
    if (!WDMA) {
      f32x4* dst = reinterpret_cast<f32x4*>(smem + (size_t)buf * WBYTES);
#pragma unroll
      for (int v = 0; v < WPT; ++v) {
        const int p = tid + v * 64 * NWV;
        if (p < WPIECES) dst[p] = wreg[v];
      }
    } else if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (xrows) {
      if (a.safe_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STP) : "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * STP) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * STP) : "memory");
    } else {
      if (a.safe_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STZ) : "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * STZ) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * STZ) : "memory");
    }
    if (tid < 2 * R / 4) *reinterpret_cast<f32x4*>(bbuf + buf * 2 * R + 4 * tid) = breg;
  };

  for (int sblk = blockIdx.x; sblk < a.nseg; sblk += gridDim.x) {
    const int seg = xcd_segment(sblk, a.nseg);
    // segment -> (clip b, residue r, first position j0); consecutive ids = neighbouring memory
    const int per_clip = a.st * a.nsub;
    const int b = seg / per_clip;
    const int rem = seg - b * per_clip;
    int r, j0;
    if (a.nsub == 1) { r = rem; j0 = 0; }
    else { r = rem / a.nsub; j0 = (rem - r * a.nsub) * a.W; }
    const int Jr = (a.Tlen - r + a.st - 1) / a.st;             // positions of this residue class (may be 0)
    const int Wseg = (Jr - j0) < a.W ? (Jr - j0) : a.W;        // positions this segment owns (<= 0: nothing to do)
    const int jbase = j0 - a.H;                                // position of image row 0
    const size_t clip = (size_t)b * a.Tlen;
    // global row (element offset / R) of position j, clamped into the residue class
    auto grow = [&](int j) -> size_t {
      int jj = j < 0 ? 0 : j;
      jj = jj < Jr ? jj : Jr - 1;
      jj = jj < 0 ? 0 : jj;
      size_t t = (size_t)jj * a.st + r;
      t = t < (size_t)a.Tlen ? t : (size_t)a.Tlen - 1;
      return clip + t;
    };

    // ---- segment image + the first layer's weights
    wload(0, 0);
    if constexpr (IC) {    // the input conv of the stack, in the arithmetic of causal_conv_cin1_kernel:
      const int nrows = a.NT * 32;      // v = bias; v = fma(x[t-1-shift], w0, v); v = fma(x[t-shift], w1, v); one rounding to T
      constexpr int RPP = 64 * NWV / LPR;
      const int c0 = (tid % LPR) * VEC;
      float w0[VEC], w1[VEC], bb[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) { w0[e] = a.ic_w[c0 + e]; w1[e] = a.ic_w[R + c0 + e]; bb[e] = a.ic_b[c0 + e]; }
      const float* au = a.ic_audio + clip;
      constexpr int UN = (MAXT * NWV * 32 + RPP - 1) / RPP;   // all of the image's rows in one batch of loads (one round trip)
      float x1[UN], x0v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        int i = u * RPP + tid / LPR;
        i = i < nrows ? i : nrows - 1;
        const int t = (int)(grow(jbase + i) - clip);        // (rows beyond the clip are clamped re-computations, as the loads were)
        const int t1 = t - a.ic_shift, t0 = t1 - 1;
        const bool ok1 = t1 >= 0 && t1 < a.Tlen, ok0 = t0 >= 0 && t0 < a.Tlen;
        x1[u] = au[ok1 ? t1 : 0];
        x0v[u] = au[ok0 ? t0 : 0];
        x1[u] = ok1 ? x1[u] : 0.0f;
        x0v[u] = ok0 ? x0v[u] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = u * RPP + tid / LPR;
        if (i < nrows) {
          float v[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] = fmaf(x1[u], w1[e], fmaf(x0v[u], w0[e], bb[e]));
          T* dst = img + (size_t)i * LS + c0;
#pragma unroll
          for (int e = 0; e < VEC; e += 4) store4(dst + e, v[e], v[e + 1], v[e + 2], v[e + 3]);
        }
      }
    } else {
      const T* x0 = reinterpret_cast<const T*>(a.x0);
      const int nrows = a.NT * 32;
      constexpr int RPP = 64 * NWV / LPR;            // rows per pass of the whole workgroup
      constexpr int UN = (MAXT * NWV * 32 + RPP - 1) / RPP;   // the whole image in one batch of loads: one HBM round trip
      for (int i0 = 0; i0 < nrows; i0 += RPP * UN) {
        f32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          int i = i0 + u * RPP + tid / LPR;
          i = i < nrows ? i : nrows - 1;
          // (non-temporal, like the backward kernel's row loads: read once, the halo rows twice)
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x0 + grow(jbase + i) * R + (tid % LPR) * VEC));
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = i0 + u * RPP + tid / LPR;
          if (i < nrows) *reinterpret_cast<f32x4*>(img + (size_t)i * LS + (tid % LPR) * VEC) = v[u];
        }
      }
    }
    wstore(0, 0, true);
    stamp(2);
    wg_barrier();
    stamp(3);
    int nstored = 0;                 // owned tiles that store rows (wave-uniform)
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      const int q = wave + NWV * m;
      int lo = a.H - 32 * q;
      lo = lo < 0 ? 0 : lo;
      int hi = a.H + Wseg - 32 * q;
      hi = hi > 32 ? 32 : hi;
      if (q < a.NT && hi > lo) ++nstored;
    }

    for (int g = 0; g < a.nl; ++g) {
      const int d = a.sub[g];
      const int wb = (NWB == 2) ? (g & 1) : 0;
      if (NWB == 1 && g > 0) wload(g, 0);
      // ---- tap (t - d) fragments of every owned tile, before anyone overwrites the rows they come from
      Frag<T> tap0[MAXT][KS];
#pragma unroll
      for (int m = 0; m < MAXT; ++m) {
        const int q = wave + NWV * m;
        if (q < a.NT) {
          int src = 32 * q + col - d;
          src = src < 0 ? 0 : src;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) tap0[m][ks] = load_nat(img + (size_t)src * LS + 16 * ks + 8 * half);
          if (WT) {
            // the layer's input of the tile's owned positions, transposed, for the backward pass's dWf (the image holds
            // x_g of every row here; the halo is a whole number of tiles in this mode, so tile q is the segment's
            // weight-gradient tile q - H/32)
            const int k = q - a.H / 32;
            int hi = Wseg - 32 * k;
            hi = hi > 32 ? 32 : hi;
            if (k >= 0 && hi > 0) {
              int lw = lane;       // (opaque copy: the tile store's addresses are not worth a register across the layer)
              asm volatile("" : "+v"(lw));
              wt_store_tile<T, R>(reinterpret_cast<T*>(a.xT) + (size_t)g * a.wt_stride + ((size_t)seg * a.KT + k) * (R * 32),
                                  img + (size_t)(32 * q) * LS, LS, hi, lw);
            }
          }
        }
      }
      if (NWB == 1 && g > 0) wstore(0, 0, true);
      stamp(10);
      wg_barrier();
      stamp(11);
      const bool more = g + 1 < a.nl;
      const bool xrows = !WT || a.store_inner_x || !more;   // the layer's output rows go to HBM (WT: the group's top layer only)
      if (NWB == 2 && more) wload(g + 1, wb ^ 1);

      const Frag<T>* lds_conv = wbuf + (size_t)wb * NW * 64;
      const Frag<T>* lds_res = lds_conv + NCONV * 64;
      const float* bl = bbuf + wb * 2 * R;
      T* zg = reinterpret_cast<T*>(a.z_out) + (size_t)g * a.layer_stride;
      T* xg = reinterpret_cast<T*>(a.x_out) + (size_t)g * a.layer_stride;
      const T* cg = COND ? reinterpret_cast<const T*>(a.cond[g]) : nullptr;

#pragma unroll
      for (int m = 0; m < MAXT; ++m) {
        const int q = wave + NWV * m;
        if (q >= a.NT) continue;
        // (WT: the halo is whole tiles, and the group's TOP layer feeds no tap inside the launch: its halo tiles would be
        // computed for nobody -- for 500 + 32 rows that is the 17th tile, a third round of eight waves for one of them)
        if (WT && !more && q < a.H / 32) continue;
        T* trow = img + (size_t)(32 * q) * LS;           // the tile's own rows
        const int j = jbase + 32 * q + col;              // this lane's position
        const bool ok0 = (j - d) >= 0;                   // causal zero padding of the delayed tap (ops.py:9)
        // rows of this tile that exist and belong to the segment: [lo, hi)
        int lo = a.H - 32 * q;
        lo = lo < 0 ? 0 : lo;
        int hi = a.H + Wseg - 32 * q;
        hi = hi > 32 ? 32 : hi;
        const bool st_ok = hi > lo;

        Frag<T> cur1[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) cur1[ks] = load_nat(trow + (size_t)col * LS + 16 * ks + 8 * half);
        raw4 xres[RT][4];
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) xres[mt][gq] = Raw4g<T>::load(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half);

        // ---- dilated causal conv as one (K*R)-deep contraction; accumulators start at the bias
        f32x16 accF[RT];
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + 32 * mt + 8 * gq + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) accF[mt][4 * gq + e] = bv[e];
          }
        // (the causal zero padding only exists in a clip's first tiles: the select on every fragment register only there)
        if (__builtin_amdgcn_ballot_w64(!ok0) != 0ull) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) tap0[m][ks] = ok0 ? tap0[m][ks] : zero_frag<T>();
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) mma(accF[mt], lds_conv[(mt * (K * KS) + ks) * 64 + lane], tap0[m][ks]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) mma(accF[mt], lds_conv[(mt * (K * KS) + KS + ks) * 64 + lane], cur1[ks]);
        if (STAMP) { asm volatile("" :: "v"(accF[0][0])); stamp(12); }

        // ---- tanh, gate; z leaves through the tile's own rows
        Frag<T> cf[KS];
        {
          float zz[RT][16];
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int mt = 0; mt < RT; ++mt)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) {
                const f32x4 z4 = tanh4_bf16(f32x4{accF[mt][4 * gq], accF[mt][4 * gq + 1], accF[mt][4 * gq + 2], accF[mt][4 * gq + 3]});
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  zz[mt][4 * gq + e] = z4[e];
                  cf[2 * mt + (gq >> 1)].set(4 * (gq & 1) + e, gate_of_z<T>(z4[e]));
                }
              }
          } else
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
              const float z = Math<T>::tanh_(accF[mt][qq]);
              zz[mt][qq] = z;
              cf[2 * mt + (qq >> 3)].set(qq & 7, gate_of_z<T>(z));
            }
          if (STAMP) { asm volatile("" :: "v"(zz[0][0])); stamp(13); }
          if (st_ok) {
            wave_lds_order();             // the reads of the own rows above are done
#pragma unroll
            for (int mt = 0; mt < RT; ++mt)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq)
                store4(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half, zz[mt][4 * gq], zz[mt][4 * gq + 1],
                       zz[mt][4 * gq + 2], zz[mt][4 * gq + 3]);
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              int rr = i * RPI + rsub;
              rr = rr < lo ? lo : (rr < hi ? rr : hi - 1);
              const f32x4 v = *reinterpret_cast<const f32x4*>(trow + (size_t)rr * LS + piece * VEC);
              __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(zg + grow(jbase + 32 * q + rr) * R + piece * VEC));
            }
          }
        }
        if (WT && st_ok) {
          // c = z sigmoid(z) of the tile, transposed, for the backward pass's dWr: through the tile's own rows like z
          wave_lds_order();
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const Frag<T>& cfr = cf[2 * mt + (gq >> 1)];
              const int e0 = 4 * (gq & 1);
              store4(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half, cfr.get(e0), cfr.get(e0 + 1), cfr.get(e0 + 2), cfr.get(e0 + 3));
            }
          wave_lds_order();
          int lw = lane;
          asm volatile("" : "+v"(lw));
          wt_store_tile<T, R>(reinterpret_cast<T*>(a.cT) + (size_t)g * a.wt_stride + ((size_t)seg * a.KT + (q - a.H / 32)) * (R * 32),
                              trow, LS, 32, lw);
        }
        stamp(14);
        // ---- 1x1 residual from registers, scaled residual add
        // (the conditioning bias of the layer above is fetched HERE, ahead of the residual MFMAs that hide its latency;
        // fetched where it is added, every tile of every layer waited a full L2 round trip.
        // The frame index is a 32-bit division: t < T < 2^31, and a 64-bit one is ~150 instructions per tile.)
        raw4 cnd[RT][4];
        bool have_cnd = false;
        if (COND && cg) {
          have_cnd = true;
          const unsigned t = (unsigned)(grow(j) - clip);      // time step of this lane's row (clamped)
          const T* crow_ = cg + ((size_t)b * a.cond_frames + t / (unsigned)a.pool) * a.cond_stride;
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) cnd[mt][gq] = Raw4g<T>::load(crow_ + 32 * mt + 8 * gq + 4 * half);
        }
        f32x16 accR[RT];
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + R + 32 * mt + 8 * gq + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) accR[mt][4 * gq + e] = bv[e];
          }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) mma(accR[mt], lds_res[(mt * KS + s) * 64 + lane], cf[s]);
        if (STAMP) { asm volatile("" :: "v"(accR[0][0])); stamp(15); }
        {
          float hv[RT][16];
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float cv = (COND && have_cnd) ? Raw4g<T>::get(cnd[mt][gq], e) : 0.0f;
                hv[mt][4 * gq + e] = (Raw4g<T>::get(xres[mt][gq], e) + accR[mt][4 * gq + e]) * kSqrtHalf + cv;   // (one fma, as layer_fwd_kernel)
              }
            }
          wave_lds_order();               // z row reads (and, without stores, the own-row reads) are done
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
              store4(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half, hv[mt][4 * gq], hv[mt][4 * gq + 1],
                     hv[mt][4 * gq + 2], hv[mt][4 * gq + 3]);
          if (st_ok && xrows) {
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              int rr = i * RPI + rsub;
              rr = rr < lo ? lo : (rr < hi ? rr : hi - 1);
              const f32x4 v = *reinterpret_cast<const f32x4*>(trow + (size_t)rr * LS + piece * VEC);
              *reinterpret_cast<f32x4*>(xg + grow(jbase + 32 * q + rr) * R + piece * VEC) = v;
            }
          }
        }
        stamp(16);
      }
      if (NWB == 2 && more) wstore(wb ^ 1, nstored, xrows);
      stamp(17);
      wg_barrier();
      stamp(18);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward chain of a group (autodiff of ops.py:23-46 through the stacking loops), top layer first.  Same segments as
// the forward kernel with the halo on the ANTI-causal side: layer g's input gradient needs df_g at t + sub_g.
//   df_g  = (Wr_g . (G_{g+1} sqrt(.5)) + dcs_g) * d(z sigmoid z)/df (z_g)              -> df_out[g]
//   G_g   = G_{g+1} sqrt(.5) + sum_k Wf_g[k] . df_g[t + (K-1-k) sub_g]                 -> g_out[g]
// G_{g+1} stays in registers between layers (fp32: exactly what layer_bwd_kernel chains from its UP half into its DOWN
// half); df_g travels through the LDS image for the shifted tap.  z and dcs arrive as whole rows one tile ahead and are
// redistributed through the tile's own image rows; df and G leave the same way.  Per layer: 256 B/row read + 256 B/row
// written (bf16, R = 64) against 768+ for one launch per layer.
// ------------------------------------------------------------------------------------------
struct GroupBwdArgs {
  const void* g_top;              // gradient wrt the group's top output [B,T,R], or null (= 0: top of the teacher stack)
  void* g_out;                    // layer g's input gradient at g_out + g*layer_stride
  void* df_out;                   // layer g's conv pre-activation gradient at df_out + g*layer_stride
  const void* z;                  // z of layer g at z + g*layer_stride
  const void* dcs;                // Ws_g . dtotal of layer g at dcs + g*layer_stride, or null (no skip path)
  int64_t layer_stride;
  const void* wconvT[kMaxGroup];  // packed [R/32][K*R/16] natural (rows = in channel), as srwn_residual_layer_bwd takes them
  const void* wresT[kMaxGroup];   // packed [R/32][R/16] permuted
  int sub[kMaxGroup];
  int hb[kMaxGroup];              // sum of sub[0 .. g-1]: how far the layers below layer g reach beyond the segment (host: no scalar-load loop per layer)
  int nl, st, Tlen, B;
  int W, H, NT, nsub, nseg;
  // WT instantiations (layer weight gradients summed in this launch): the forward kernel's weight-gradient tiles (layer g at
  // xT / cT + g*wt_stride, tile (seg, k) at ((seg*KT)+k)*R*32) and the partial sums (layer g, slab s at ((g*nslabs)+s)*n
  // floats, in the layout srwn_wgrad_layers writes); df is not stored, G only with write_all_g
  const void* xT; const void* cT; int64_t wt_stride; int KT;
  // part_f / part_r: fp32 in srwn_wgrad_layers' layout, or (P16 instantiations: a.part16) bf16 16 x 16 blocks in lane
  // order -- block (row block rb, column block ob) of a [rows, R] matrix at ((slab * nblocks) + rb * (R/16) + ob) * 256
  // elements, lane l's four values (rows 16 rb + 4 (l >> 4) + 0..3 of column 16 ob + (l & 15)) at + 4 l: one 8-byte
  // store per lane and block, a wave's block one contiguous 512 bytes (SRWN_PARTIALS_BLK16 of srwn_reduce_partials_multi)
  void* part_f; void* part_r; float* part_bf; float* part_br;
  int nslabs, write_all_g, part16;
  // ICG instantiations (the stack's FIRST group): the input conv's kernel + bias gradient (model.py:40; tf.gradients of
  // ops.py:6-20 for the 1 -> R conv) from the group's bottom gradient while it is in the image:
  //   ic_part[slab][k*R + c] = sum_t audio[t - (1-k) - shift] G_0[t][c],   ic_part[slab][2R + c] = sum_t G_0[t][c]
  // over the rows the workgroup's segments own (srwn_init_conv_wgrad's stage-1 layout; 8 / (R/16) slabs per workgroup:
  // slab (workgroup, h) sums the tiles k = h mod that many)
  const float* ic_audio; float* ic_part; int ic_shift;
  unsigned long long* stamps;     // diagnostic instantiation only (srwn_debug_stamp_buffer): waves 0 and 4 of workgroup 0
  int dbg;                        // diagnostic build only (SRWN_WT_DEBUG): 1 = skip the dWr contraction, 2 = skip the dWf one,
                                  // 128 = the second wave of every pair that shares tile fragments loads none (timing only)
};
#ifdef SRWN_DIAG
#define WT_DBG(a) ((a).dbg)
#else
#define WT_DBG(a) 0      // the shipped kernels carry no wrong-answer switches (and no registers for them)
#endif

constexpr int kWtPadRows = 64;
// Two ways of hiding the weight-gradient fragment loads behind the chain, both built and measured, both off: they need
// registers the R = 64 kernels do not have (256 of 256 in use: every extra live value is a scratch access in a hot loop).
// the dWr loop's first eight c^T fragments are requested before G is parked (one HBM round trip under the parking and its
// barrier: six launches -6.7 us, step -4 us on one box; no spills in the two-tile kernel, 13 more dwords in the three-tile
// one).  The same for the dWf loop's x^T fragments ahead of phase B: -1.5 us, inside the noise, not kept.
constexpr bool kWtEarlyR = true;      // (bf16 kernels: an fp32 fragment set is 64 registers)
constexpr bool kWtStagger = false;   // half of the waves contract dWf before their taps, half after
constexpr bool kWtEarlyC = false;    // the next layer's first c^T fragments requested a phase early   // finite (zero) rows behind the image: the shifted tap of the last weight-gradient tile reads past it

template <typename T, int RT, bool DCS, int MAXT, int NWB, int NWV = 8, bool WT = false, bool STAMP = false, bool P16 = false, bool ICG = false>
__global__ __launch_bounds__(64 * NWV) void group_bwd_kernel(GroupBwdArgs a) {
  static_assert(!P16 || (WT && sizeof(T) == 2), "16-bit partial blocks: the bf16 weight-gradient-tile mode only");
  static_assert(!ICG || WT, "input conv gradient: the weight-gradient-tile mode only");
  constexpr int R = 32 * RT, K = 2, KS = R / 16;
  Stamper<STAMP> stamp{nullptr, 0};      // (tools/gb_stamps.py) lane 0 of waves 0 and 4 -- the two waves of SIMD 0 -- of workgroup 0
  if (STAMP && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 3) == 0) stamp.p = a.stamps + (threadIdx.x >> 8) * 512;
  stamp(1);
  constexpr int NCONV = RT * K * KS, NRES = RT * KS, NW = NCONV + NRES;
  constexpr int LS = RowStage<T>::stride(R), VEC = RowStage<T>::VEC;
  constexpr int LPR = R / VEC, RPI = 64 / LPR, NI = 32 / RPI;
  constexpr int WBYTES = NW * 64 * (int)sizeof(Frag<T>);
  constexpr int WPIECES = WBYTES / 16, CPIECES = NCONV * 64 * (int)sizeof(Frag<T>) / 16;
  constexpr int KTMAX = NWV * MAXT;          // (WT) most weight-gradient tiles a segment can have
  typedef typename Raw4g<T>::type raw4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<T>* wbuf = reinterpret_cast<Frag<T>*>(smem);                       // [NWB][convT | resT]
  T* img = reinterpret_cast<T*>(smem + (size_t)NWB * WBYTES);             // [NT*32][LS]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int rsub = lane / LPR, piece = lane % LPR;

  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto wload = [&](int g, int buf) {   // layer g's [convT | resT] images -> weight buffer `buf` by LDS-DMA
    dma_image(a.wconvT[g], lds_base + buf * WBYTES, CPIECES * 16, wave, lane, NWV);
    dma_image(a.wresT[g], lds_base + buf * WBYTES + CPIECES * 16, (WPIECES - CPIECES) * 16, wave, lane, NWV);
  };
  const int ntw = (a.NT - wave + NWV - 1) / NWV;  // tiles this wave owns: q = wave + NWV*m, m < ntw

  int sit = -1;                                   // segments this workgroup has finished (WT: later ones add to its partial slab)
  for (int sblk = blockIdx.x; sblk < a.nseg; sblk += gridDim.x) {
    const int seg = xcd_segment(sblk, a.nseg);
    ++sit;
    const int per_clip = a.st * a.nsub;
    const int b = seg / per_clip;
    const int rem = seg - b * per_clip;
    int r, j0;
    if (a.nsub == 1) { r = rem; j0 = 0; }
    else { r = rem / a.nsub; j0 = (rem - r * a.nsub) * a.W; }
    const int Jr = (a.Tlen - r + a.st - 1) / a.st;
    const int Wseg = (Jr - j0) < a.W ? (Jr - j0) : a.W;
    const int jbase = j0;                                       // image row 0 = first owned position; halo behind it
    const size_t clip = (size_t)b * a.Tlen;
    auto grow = [&](int j) -> size_t {
      int jj = j < Jr ? j : Jr - 1;
      jj = jj < 0 ? 0 : jj;
      size_t t = (size_t)jj * a.st + r;
      t = t < (size_t)a.Tlen ? t : (size_t)a.Tlen - 1;
      return clip + t;
    };
    // whole rows of one tile: registers <-> the tile's own image rows <-> HBM
    auto rows_load = [&](const T* base, int q, f32x4 (&v)[NI]) {
#pragma unroll
      // (non-temporal: z, dcs and the top gradient are read once by this workgroup -- the halo rows a second time by its
      // neighbour -- and, left in the L2, push out the tile fragments that two waves of the workgroup share: same box,
      // six launches 575 -> 562 us, step 1.5397 -> 1.5289)
      for (int i = 0; i < NI; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + grow(jbase + 32 * q + i * RPI + rsub) * R + piece * VEC));
    };
    auto rows_put = [&](T* trow, const f32x4 (&v)[NI]) {
      wave_lds_order();
#pragma unroll
      for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(trow + (size_t)(i * RPI + rsub) * LS + piece * VEC) = v[i];
      wave_lds_order();
    };
    auto acc_get = [&](const T* trow, raw4 (&o)[RT][4]) {
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) o[mt][gq] = Raw4g<T>::load(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half);
    };
    // accumulator-layout values -> own rows -> HBM rows [0, hi) of the tile (hi >= 1)
    auto tile_store = [&](T* trow, T* gbase, int q, int hi, const float (&vals)[RT][16]) {
      wave_lds_order();
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          store4(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half, vals[mt][4 * gq], vals[mt][4 * gq + 1],
                 vals[mt][4 * gq + 2], vals[mt][4 * gq + 3]);
      wave_lds_order();
      if (hi > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          int rr = i * RPI + rsub;
          rr = rr < hi ? rr : hi - 1;
          const f32x4 v = *reinterpret_cast<const f32x4*>(trow + (size_t)rr * LS + piece * VEC);
          *reinterpret_cast<f32x4*>(gbase + grow(jbase + 32 * q + rr) * R + piece * VEC) = v;
        }
      }
    };

    auto tile_store_raw = [&](T* trow, T* gbase, int q, int hi, const raw4 (&vals)[RT][4]) {
      wave_lds_order();
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<raw4*>(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half) = vals[mt][gq];
      wave_lds_order();
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int rr = i * RPI + rsub;
        rr = rr < hi ? rr : hi - 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(trow + (size_t)rr * LS + piece * VEC);
        *reinterpret_cast<f32x4*>(gbase + grow(jbase + 32 * q + rr) * R + piece * VEC) = v;
      }
    };

    // ---- G of the group's top output (registers, accumulator layout), the top layer's weights
    // (kept in the storage type: what one launch per layer would read back from HBM; in fp32 mode that is exact)
    raw4 G[MAXT][RT][4];
    Frag<T> acn[8];                 // (WT) c^T fragments of the next layer's dWr contraction, requested a phase early
    const int gtop = a.nl - 1;
    // (WT) column sums of the image over the rows the segment owns -- the two bias gradients of a layer -- as products with
    // a fragment of ones: wave w < R/16 sums the 16 columns 16 w.. with one MFMA and one transposing read per tile.  (They
    // used to ride in the contraction loops as four v_dot2 per fragment on EVERY wave, a third of those loops'
    // instructions; the loops are bound by the instructions a SIMD's two waves issue.)  rows >= Wseg of the last tile are
    // masked in the ones (mask = true: the df image holds the halo's rows there; G is parked with them zeroed).
    auto colsum = [&](float* pb_slab, int ntiles, bool mask) {
      if (!WT || wave >= R / 16) return;
      int lw = lane;
      asm volatile("" : "+v"(lw));
      const T* cb = LdT16p<T>::base(img, LS, lw) + 16 * wave;
      Frag<T> ones;
#pragma unroll
      for (int e = 0; e < 8; ++e) ones.set(e, 1.0f);
      f32x4 accb = {0.f, 0.f, 0.f, 0.f};
      const int kfull = Wseg >> 5;
      // tiles the segment owns whole first (plain ones, no per-value selects in the loop: with them this loop, which only four
      // waves run, was 170 instructions per four tiles against 55 -- longer than those waves' share of the dWf contraction
      // that follows it), then the one tile it may own in part
      const int nfull = mask ? (kfull < ntiles ? kfull : ntiles) : ntiles;
      constexpr int CU = 4;      // tiles per batch of transposing reads (one LDS round trip per batch, not per tile)
#pragma unroll 1
      for (int k0 = 0; k0 < nfull; k0 += CU) {
        Frag<T> bf[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
          const int k = k0 + u < nfull ? k0 + u : nfull - 1;
          bf[u] = LdT16p<T>::template load<LS>(cb + (size_t)(32 * k) * LS, 0);
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
          if (k0 + u < nfull) mma16(accb, ones, bf[u]);      // (wave-uniform)
        }
      }
      if (mask && nfull < ntiles) {
        const int hik = Wseg - 32 * nfull;
        Frag<T> ok;
#pragma unroll
        for (int e = 0; e < 8; ++e) ok.set(e, kordW(lw >> 4, e) < hik ? 1.0f : 0.0f);
        const Frag<T> bfl = LdT16p<T>::template load<LS>(cb + (size_t)(32 * nfull) * LS, 0);
        mma16(accb, ok, bfl);
      }
      if (lw < 16) {      // (row 0 of the 16 x 16 result: every row holds the sums)
        float* pb = pb_slab + 16 * wave + lw;
        *pb = (sit > 0 ? *pb : 0.0f) + accb[0];
      }
    };
    if (WT) {
      // every row the time contractions may touch must be finite: rows a layer has not (re)written only ever meet the zeroed
      // (not owned) rows of the other operand, but 0 x NaN from stale LDS would poison a sum
      f32x4* p = reinterpret_cast<f32x4*>(img);
      for (int i = tid; i < (a.NT * 32 + kWtPadRows) * LS * (int)sizeof(T) / 16; i += 64 * NWV) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      wg_barrier();
    }
    wload(gtop, 0);
    f32x4 zr[NI], dr[NI];
    auto issue = [&](int g, int q) {
      rows_load(reinterpret_cast<const T*>(a.z) + (size_t)g * a.layer_stride, q, zr);
      if (DCS) rows_load(reinterpret_cast<const T*>(a.dcs) + (size_t)g * a.layer_stride, q, dr);
    };
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      const int q = wave + NWV * m;
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) G[m][mt][gq] = Raw4g<T>::zero();
      if (q < a.NT && a.g_top) {
        T* trow = img + (size_t)(32 * q) * LS;
        f32x4 v[NI];
        rows_load(reinterpret_cast<const T*>(a.g_top), q, v);
        rows_put(trow, v);
        acc_get(trow, G[m]);
        if (WT && !((jbase + 32 * q + col) < Jr)) {      // rows beyond the clip are clamped re-reads: G = 0 there from here on
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) G[m][mt][gq] = Raw4g<T>::zero();
        }
      }
    }
    dma_wait();
    // operands of the first tile of the top layer (one tile ahead from here on)
    issue(gtop, wave < a.NT ? wave : a.NT - 1);
    wg_barrier();

    for (int n = 0; n < a.nl; ++n) {
      const int g = a.nl - 1 - n;
      const int d = a.sub[g];
      int wb = 0;
      if (NWB == 2) {
        wb = n & 1;
        if (g > 0) wload(g - 1, wb ^ 1);
      } else if (n > 0) {
        wload(g, 0); dma_wait(); wg_barrier();
      }
      const Frag<T>* lds_conv = wbuf + (size_t)wb * NW * 64;
      const Frag<T>* lds_res = lds_conv + NCONV * 64;
      T* dfg = reinterpret_cast<T*>(a.df_out) + (size_t)g * a.layer_stride;
      T* gprev = reinterpret_cast<T*>(a.g_out) + (size_t)(g + 1) * a.layer_stride;   // where G_{g+1} goes (n > 0)
      const bool haveg = (n > 0) || (a.g_top != nullptr);
      // The halo shrinks on the way down: the layers below g reach hb = sum of their sub-dilations beyond the segment,
      // so G_g is needed for positions < Wseg + hb and df_g for positions < Wseg + hb + sub_g.  Tiles beyond that are
      // skipped (for 500 + 31 positions the 17th tile is live in three of the ten phases of a 1..16 group only --
      // with eight waves it is a whole third round of its phase).
      const int hb = a.hb[g];
      int ntA = (Wseg + hb + d + 31) / 32, ntB = (Wseg + hb + 31) / 32;
      ntA = (a.H == 0 || ntA > a.NT) ? a.NT : ntA;
      ntB = (a.H == 0 || ntB > a.NT) ? a.NT : ntB;

      // (WT) the wave's share of dWr_g: NBW 16 x 16 output blocks (16 rows of c channels x 16 G channels each)
      constexpr int NIB = R / 16, NBLK = NIB * NIB;
      constexpr int NBW = NBLK >= NWV ? NBLK / NWV : 1;
      const bool active2 = NBLK >= NWV || wave < NBLK;
      const int blk2 = wave * NBW, ib2 = blk2 / NIB, ob2 = blk2 - ib2 * NIB;
      const size_t pslab = (size_t)g * a.nslabs + blockIdx.x;      // (WT) this workgroup's partial slab of layer g
      const int ktn = (Wseg + 31) / 32;                            // (WT) weight-gradient tiles of the segment
      if (WT) {
        // ---- dWr_g = c_g^T . G_{g+1}, dbr_g = colsum(G_{g+1}) over the rows the segment owns.  The image is free between
        // two layers: every wave parks G_{g+1} of its tiles there (rows beyond the owned ones zeroed), then the R x R sum is
        // split by OUTPUT over the waves -- 16 x 16 blocks, c^T fragments straight from the forward kernel's tiles in HBM,
        // G^T by transposing reads -- so a wave carries 4 accumulator registers per block instead of the whole sum.
        const bool active = active2;
        const int ib = ib2, ob0 = ob2;
        f32x4 acc[NBW];
#pragma unroll
        for (int bb = 0; bb < NBW; ++bb) acc[bb] = f32x4{0.f, 0.f, 0.f, 0.f};
        int lw = lane;      // (opaque copy: keeps this block's lane-dependent addresses from being hoisted over the chain,
        asm volatile("" : "+v"(lw));   //  where every register counts)
        if (kWtEarlyR && sizeof(T) == 2 && haveg && active && !(WT_DBG(a) & 1)) {
          const T* ct0 = reinterpret_cast<const T*>(a.cT) + (size_t)g * a.wt_stride + (size_t)seg * a.KT * (R * 32);
#pragma unroll
          for (int j = 0; j < 8; ++j) acn[j] = wt_load(ct0 + (size_t)(j < ktn ? j : (ktn > 0 ? ktn - 1 : 0)) * (R * 32), 16 * ib, lw);
        }
        if (haveg) {
#pragma unroll
          for (int m = 0; m < MAXT; ++m) {
            const int q = wave + NWV * m;
            int hi = Wseg - 32 * q;
            hi = hi > 32 ? 32 : hi;
            if (q < a.NT && hi > 0) {
              T* trow = img + (size_t)(32 * q) * LS;
              const bool own = col < hi;
              wave_lds_order();
#pragma unroll
              for (int mt = 0; mt < RT; ++mt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                  *reinterpret_cast<raw4*>(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half) = own ? G[m][mt][gq] : Raw4g<T>::zero();
              wave_lds_order();
              if (a.write_all_g && n > 0) {      // (the conditioned decoders sum G per frame: model.py:180)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                  int rr = i * RPI + rsub;
                  rr = rr < hi ? rr : hi - 1;
                  const f32x4 v = *reinterpret_cast<const f32x4*>(trow + (size_t)rr * LS + piece * VEC);
                  *reinterpret_cast<f32x4*>(gprev + grow(jbase + 32 * q + rr) * R + piece * VEC) = v;
                }
              }
            }
          }
          stamp(20);
          wg_barrier();
          stamp(21);
          colsum(a.part_br + pslab * R, ktn, false);      // dbr_g = colsum(G_{g+1}) (rows the segment does not own are zero here)
          if (active && !(WT_DBG(a) & 1)) {
            const T* ct = reinterpret_cast<const T*>(a.cT) + (size_t)g * a.wt_stride + (size_t)seg * a.KT * (R * 32);
            // eight tiles' fragments requested at a time, the next eight before the first are used: one or two HBM round
            // trips per loop instead of one per tile (a chain of dependent round trips made this loop cost as much as the
            // whole chain).  No branches inside: tiles beyond the segment's contribute a zero fragment.
            const T* gbase = LdT16p<T>::base(img, LS, lw) + 16 * ob0;
            Frag<T> av[8], bv[8];
            if ((kWtEarlyR && sizeof(T) == 2) || (kWtEarlyC && n > 0)) {           // requested before G was parked / during the layer above
#pragma unroll
              for (int j = 0; j < 8; ++j) av[j] = acn[j];
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) av[j] = wt_load(ct + (size_t)(j < ktn ? j : (ktn > 0 ? ktn - 1 : 0)) * (R * 32), 16 * ib, lw);
            }
            // (two groups of eight per trip, the two register sets swapping roles: copying the set that arrived into the set in
            // use was 16 moves + 7 waits per group -- a tenth of this loop's instructions, and instructions are its time)
            auto group8 = [&](int k0, Frag<T> (&cur)[8], Frag<T> (&nxt)[8]) __attribute__((always_inline)) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int kn = k0 + 8 + j;
                if ((WT_DBG(a) & 128) && (wave & 1)) nxt[j] = zero_frag<T>();      // (timing experiment: the second wave of a pair loads nothing)
                else
                nxt[j] = wt_load(ct + (size_t)(kn < ktn ? kn : ktn - 1) * (R * 32), 16 * ib, lw);
              }
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                if (k < ktn) {     // (wave-uniform)
                  const T* pk = gbase + (size_t)(32 * k) * LS;
                  Frag<T> bfv[NBW];
#pragma unroll
                  for (int bb = 0; bb < NBW; ++bb) bfv[bb] = LdT16p<T>::template load<LS>(pk, 16 * bb);
                  __builtin_amdgcn_sched_group_barrier(0x100, 2 * NBW, 0);
                  __builtin_amdgcn_sched_group_barrier(0x008 | 0x002, 64, 0);
#pragma unroll
                  for (int bb = 0; bb < NBW; ++bb) {
                    mma16(acc[bb], cur[j], bfv[bb]);
                  }
                }
                __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler hoists all the transposing reads of the chunk: spills)
              }
            };
#pragma unroll 1
            for (int k0 = 0; k0 < ktn; k0 += 16) {
              group8(k0, av, bv);
              if (k0 + 8 < ktn) group8(k0 + 8, bv, av);
            }
          }
          stamp(22);
          wg_barrier();      // the image goes back to the chain
          stamp(23);
        }
        if (active) {
          // one 16 x 16 block per accumulator: lane l holds rows 4 (l >> 4) + rr of column l & 15
          if constexpr (P16) {      // bf16 blocks in lane order: one 8-byte store per lane and block
            typedef typename Raw4g<bf16_t>::type p4;
            bf16_t* pl = reinterpret_cast<bf16_t*>(a.part_r) + ((pslab * NBLK + (size_t)(ib * NIB + ob0)) * 64 + lw) * 4;
#pragma unroll
            for (int bb = 0; bb < NBW; ++bb) {
              if (sit > 0) {
                const p4 o = *reinterpret_cast<const p4*>(pl + bb * 256);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) acc[bb][rr] += (float)o[rr];
              }
              // (non-temporal: nobody reads a partial before the reduction launch; A/B -3..-6 us per step, inside the noise)
              if (!(WT_DBG(a) & 16)) __builtin_nontemporal_store(Raw4g<bf16_t>::pack(acc[bb][0], acc[bb][1], acc[bb][2], acc[bb][3]), reinterpret_cast<p4*>(pl + bb * 256));
            }
          } else {
          float* pl = reinterpret_cast<float*>(a.part_r) + pslab * (R * R) + (size_t)(16 * ib + 4 * (lw >> 4)) * R + 16 * ob0 + (lw & 15);
          if (sit > 0) {       // a later segment of this workgroup: its sums join the earlier ones (fixed order)
#pragma unroll
            for (int bb = 0; bb < NBW; ++bb) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) acc[bb][rr] += pl[rr * R + 16 * bb];
            }
          }
#pragma unroll
          for (int bb = 0; bb < NBW; ++bb) {
            if (!(WT_DBG(a) & 16)) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) pl[rr * R + 16 * bb] = acc[bb][rr];
            }
          }
          }
        }
        if (!haveg) colsum(a.part_br + pslab * R, 0, false);      // (no gradient from above: dbr = 0 is still written)
      }

      stamp(24);      // (the wave's partial of dWr is written)
      // ---- phase A: df of every owned tile
#pragma unroll
      for (int m = 0; m < MAXT; ++m) {
        const int q = wave + NWV * m;
        if (q >= ntA) continue;
        T* trow = img + (size_t)(32 * q) * LS;
        const bool ok = (jbase + 32 * q + col) < Jr;
        int hi = Wseg - 32 * q;
        hi = hi > 32 ? 32 : hi;
        if (!WT && n > 0 && hi > 0) tile_store_raw(trow, gprev, q, hi, G[m]);   // G_{g+1}: complete since the last barrier
        raw4 zz[RT][4], dc0[RT][4];
        rows_put(trow, zr);
        acc_get(trow, zz);
        if (DCS) { rows_put(trow, dr); acc_get(trow, dc0); }
        {   // next operands: the wave's next tile of this layer, or its first tile of the layer below
          const bool same = (m + 1 < MAXT) && (q + NWV < ntA);
          if (same) issue(g, q + NWV);
          else if (g > 0) issue(g - 1, wave < a.NT ? wave : a.NT - 1);
        }
        f32x16 accC[RT];
        if (DCS && !ok) {      // (rows beyond the clip: the select on the packed words, not on every value)
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) dc0[mt][gq] = Raw4g<T>::zero();
        }
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) accC[mt][4 * gq + e] = DCS ? Raw4g<T>::get(dc0[mt][gq], e) : 0.0f;
        if (haveg) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            Frag<T> bfr;
            if constexpr (WT && sizeof(T) == 2) {
              bf16x4 hv[2];
#pragma unroll
              for (int h2 = 0; h2 < 2; ++h2) {
                const raw4& gr = G[m][s >> 1][2 * (s & 1) + h2];
                f32x4 g4 = {Raw4g<T>::get(gr, 0), Raw4g<T>::get(gr, 1), Raw4g<T>::get(gr, 2), Raw4g<T>::get(gr, 3)};
                g4 = g4 * f32x4{kSqrtHalf, kSqrtHalf, kSqrtHalf, kSqrtHalf};
                hv[h2] = Raw4g<bf16_t>::pack(g4[0], g4[1], g4[2], g4[3]);
              }
              bfr.v = __builtin_shufflevector(hv[0], hv[1], 0, 1, 2, 3, 4, 5, 6, 7);
            } else
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
              bfr.set(jj, ((WT || ok) ? Raw4g<T>::get(G[m][s >> 1][2 * (s & 1) + (jj >> 2)], jj & 3) : 0.0f) * kSqrtHalf);   // (WT: G is 0 beyond the clip, see phase B)
#pragma unroll
            for (int mt = 0; mt < RT; ++mt) mma(accC[mt], lds_res[(mt * KS + s) * 64 + lane], bfr);
          }
        }
        float dv[RT][16];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const f32x4 z4 = {Raw4g<T>::get(zz[mt][gq], 0), Raw4g<T>::get(zz[mt][gq], 1), Raw4g<T>::get(zz[mt][gq], 2), Raw4g<T>::get(zz[mt][gq], 3)};
              const f32x4 a4 = {accC[mt][4 * gq], accC[mt][4 * gq + 1], accC[mt][4 * gq + 2], accC[mt][4 * gq + 3]};
              const f32x4 d4 = a4 * dgate_df4(z4);
#pragma unroll
              for (int e = 0; e < 4; ++e) dv[mt][4 * gq + e] = d4[e];
            }
        } else
        {
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              dv[mt][4 * gq + e] = accC[mt][4 * gq + e] * dgate_df<T>(Raw4g<T>::get(zz[mt][gq], e));
        }
        tile_store(trow, dfg, q, WT ? 0 : hi, dv);     // (WT: into the image only; its readers are all on the chip)
        __builtin_amdgcn_sched_barrier(0);   // keep the tile bodies apart: interleaving them only lengthens live ranges
      }
      stamp(25);
      wg_barrier();
      stamp(26);

      // (WT) the wave's share of dWf_g: output blocks (tap, 16 rows of x channels, NBF x 16 df channels)
      constexpr int NIB5 = R / 16, NBLK5 = NIB5 * NIB5;
      constexpr int NBF = 2 * NBLK5 >= NWV ? 2 * NBLK5 / NWV : 1;
      const bool active5 = WT && (2 * NBLK5 >= NWV || wave < 2 * NBLK5);
      const int blk5 = wave * NBF, tap5 = blk5 / NBLK5, rem5 = blk5 - tap5 * NBLK5, ib5 = rem5 / NIB5, ob5 = rem5 - ib5 * NIB5;
      int lw5 = lane;      // (opaque copy: keeps this block's lane-dependent addresses from being carried through the chain)
      asm volatile("" : "+v"(lw5));
      const T* xt5 = reinterpret_cast<const T*>(a.xT) + (size_t)g * a.wt_stride + (size_t)seg * a.KT * (R * 32);
      if (kWtEarlyC && WT && g > 0 && active2) {
        // the first eight c^T fragments of the layer BELOW (its dWr contraction opens that layer, between two barriers
        // where nothing else can hide their latency) are requested now and arrive while the wave runs this layer's taps
        const T* ctn = reinterpret_cast<const T*>(a.cT) + (size_t)(g - 1) * a.wt_stride + (size_t)seg * a.KT * (R * 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) acn[j] = wt_load(ctn + (size_t)(j < ktn ? j : (ktn > 0 ? ktn - 1 : 0)) * (R * 32), 16 * ib2, lw5);
      }

      auto phaseB = [&]() {
      // ---- phase B: G_g = G_{g+1} sqrt(.5) + taps of df_g
#pragma unroll
      for (int m = 0; m < MAXT; ++m) {
        const int q = wave + NWV * m;
        if (q >= ntB) continue;
        const int i0 = 32 * q + col;
        const int j = jbase + i0;
        const bool ok = j < Jr;
        const bool ok_d = (j + d) < Jr;                 // the shifted tap stays inside the clip (zero beyond it)
        int src = i0 + d;
        if (!WT) src = src < a.NT * 32 ? src : a.NT * 32 - 1;      // (WT: kWtPadRows zeroed rows lie behind the image, d < 64: a tap
                                                                    // past the image reads zeros -- clamped it would read the last row,
                                                                    // which only the select on ok_d kept out)
        f32x16 accG[RT];
        if constexpr (WT) {      // (no select: without a gradient from above G is still the zeros it was initialised with)
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              f32x4 g4 = {Raw4g<T>::get(G[m][mt][gq], 0), Raw4g<T>::get(G[m][mt][gq], 1), Raw4g<T>::get(G[m][mt][gq], 2), Raw4g<T>::get(G[m][mt][gq], 3)};
              g4 = g4 * f32x4{kSqrtHalf, kSqrtHalf, kSqrtHalf, kSqrtHalf};
#pragma unroll
              for (int e = 0; e < 4; ++e) accG[mt][4 * gq + e] = g4[e];
            }
        } else
        {
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            accG[mt][e] = (haveg && (WT || ok)) ? Raw4g<T>::get(G[m][mt][e >> 2], e & 3) * kSqrtHalf : 0.0f;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const Frag<T> f0 = load_nat(img + (size_t)src * LS + 16 * ks + 8 * half);
          // (WT: no selects -- the image is zeroed per segment and df of a row beyond the clip is computed as 0 (its dcs is
          // masked, its G is 0), so G stays 0 there layer after layer; without the zeroed image a previous segment's rows
          // may lie behind the clip's end)
          const Frag<T> bfr = (WT || ok_d) ? f0 : zero_frag<T>();
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) mma(accG[mt], lds_conv[(mt * (K * KS) + ks) * 64 + lane], bfr);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const Frag<T> f1 = load_nat(img + (size_t)i0 * LS + 16 * ks + 8 * half);
          const Frag<T> bfr = (WT || ok) ? f1 : zero_frag<T>();
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) mma(accG[mt], lds_conv[(mt * (K * KS) + KS + ks) * 64 + lane], bfr);
        }
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            G[m][mt][gq] = Raw4g<T>::pack(accG[mt][4 * gq], accG[mt][4 * gq + 1], accG[mt][4 * gq + 2], accG[mt][4 * gq + 3]);
        __builtin_amdgcn_sched_barrier(0);
      }
      };   // phaseB
      auto dwf = [&]() {
        // ---- dWf_g[k] = sum over the owned rows s of x_g[s] (x) df_g[s + (K-1-k) d], dbf_g = colsum(df_g): df_g is complete
        // in the image since the barrier between the phases (its readers only read), x^T comes from the forward kernel's tiles
        const bool active = active5;
        const int tap = tap5, ib = ib5, ob0 = ob5;
        colsum(a.part_bf + pslab * R, WT_DBG(a) & 2 ? 0 : ktn, true);      // dbf_g = colsum(df_g) over the rows the segment owns
        if (active) {
          const int lw = lw5;
          const int shift = tap == 0 ? d : 0;             // tap 0 multiplies x[t - d]: row s of x meets row s + d of df
          f32x4 acc[NBF];
#pragma unroll
          for (int bb = 0; bb < NBF; ++bb) acc[bb] = f32x4{0.f, 0.f, 0.f, 0.f};
          const T* xt = xt5;
          const int ibl = (WT_DBG(a) & 8) ? 0 : ib;     // (timing experiment: every wave loads the same quarter of each tile)
          const int ktl = (WT_DBG(a) & 2) ? 0 : ktn;
          const T* dbase = LdT16p<T>::base(img, LS, lw) + (size_t)shift * LS + 16 * ob0;
          Frag<T> av[8], bv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) av[j] = wt_load(xt + (size_t)(j < ktn ? j : (ktn > 0 ? ktn - 1 : 0)) * (R * 32), 16 * ibl, lw);
          auto group8 = [&](int k0, Frag<T> (&cur)[8], Frag<T> (&nxt)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int kn = k0 + 8 + j;
              if ((WT_DBG(a) & 128) && wave >= NWV / 2) nxt[j] = zero_frag<T>();      // (timing experiment: see dWr)
              else
              nxt[j] = wt_load(xt + (size_t)(kn < ktn ? kn : ktn - 1) * (R * 32), 16 * ibl, lw);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int k = k0 + j;
              if (k < ktn) {       // (wave-uniform)
                const T* pk = dbase + (size_t)(32 * k) * LS;
                // (all of the tile's image fragments in ONE batch of transposing reads: read -> wait -> MFMA block by block
                // is three LDS round trips per tile -- stamps: the loop was 15 600 cycles for 64 MFMAs)
                Frag<T> bfv[NBF];
#pragma unroll
                for (int bb = 0; bb < NBF; ++bb) bfv[bb] = LdT16p<T>::template load<LS>(pk, 16 * bb);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NBF, 0);      // the reads first ...
                __builtin_amdgcn_sched_group_barrier(0x008 | 0x002, 64, 0);   // ... then the products and the column sums
#pragma unroll
                for (int bb = 0; bb < NBF; ++bb) {
                  mma16(acc[bb], cur[j], bfv[bb]);
                }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          };
#pragma unroll 1
          for (int k0 = 0; k0 < ktl; k0 += 16) {      // (two groups per trip, the register sets swapping roles: see dWr)
            group8(k0, av, bv);
            if (k0 + 8 < ktl) group8(k0 + 8, bv, av);
          }
          if constexpr (P16) {      // (the [2 R, R] matrix of both taps: row block tap * R/16 + ib)
            typedef typename Raw4g<bf16_t>::type p4;
            bf16_t* pl = reinterpret_cast<bf16_t*>(a.part_f) + ((pslab * (2 * NBLK5) + (size_t)(tap * NBLK5 + ib * NIB5 + ob0)) * 64 + lw) * 4;
#pragma unroll
            for (int bb = 0; bb < NBF; ++bb) {
              if (sit > 0) {
                const p4 o = *reinterpret_cast<const p4*>(pl + bb * 256);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) acc[bb][rr] += (float)o[rr];
              }
              // (non-temporal: nobody reads a partial before the reduction launch; A/B -3..-6 us per step, inside the noise)
              if (!(WT_DBG(a) & 16)) __builtin_nontemporal_store(Raw4g<bf16_t>::pack(acc[bb][0], acc[bb][1], acc[bb][2], acc[bb][3]), reinterpret_cast<p4*>(pl + bb * 256));
            }
          } else {
          float* pl = reinterpret_cast<float*>(a.part_f) + pslab * (2 * R * R) + (size_t)tap * R * R + (size_t)(16 * ib + 4 * (lw >> 4)) * R + 16 * ob0 + (lw & 15);
          if (sit > 0) {
#pragma unroll
            for (int bb = 0; bb < NBF; ++bb) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) acc[bb][rr] += pl[rr * R + 16 * bb];
            }
          }
#pragma unroll
          for (int bb = 0; bb < NBF; ++bb) {
            if (!(WT_DBG(a) & 16)) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) pl[rr * R + 16 * bb] = acc[bb][rr];
            }
          }
          }
        }
      };   // dwf
      // Both only READ the image.  The contraction is a burst of HBM traffic (every x^T tile of the segment), the taps are
      // issue-bound: half of the waves run one first, half the other, so each covers the other's idle resource.
      // (Measured: the second copy of the two bodies and the longer live ranges cost 30-60 spilled registers in the R = 64
      // kernels, which are at 256 already -- 0.74 -> 0.98 ms per step; off.)
      if (kWtStagger && WT && wave < NWV / 2) { dwf(); phaseB(); }
      else {
        phaseB(); stamp(27); if (WT) dwf();
      }
      stamp(28);
      if (NWB == 2 && g > 0) dma_wait();
      stamp(29);
#ifdef SRWN_DIAG
      // Priced, not shipped ("partial sums across workgroups"): what summing the per-workgroup weight-gradient
      // partials INSIDE the launch would cost.  Bit 32: the publish half -- every wave drains its partial stores, the
      // workgroup's barrier, one lane's agent-scope release and a ticket on the counter of (layer, quad of workgroups with
      // equal blockIdx % 8: one XCD under round-robin dispatch).  Bit 64: the combine half as well -- the quad's last
      // arriver reads the four 49-KB partials back (sc1 loads, agent acquire) and sums them.  The sums are discarded:
      // results with these bits set are meaningless, only the launch time is read (tools/fence_probe.py).
      if (WT && (WT_DBG(a) & 32) && a.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
        __shared__ unsigned s_ticket;
        const unsigned quad = (blockIdx.x & 7u) + 8u * (blockIdx.x >> 5);      // blocks b, b+8, b+16, b+24 of one XCD
        unsigned* counters = reinterpret_cast<unsigned*>(a.stamps + 1024);     // (behind the two stamp areas)
        if (threadIdx.x == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          s_ticket = __hip_atomic_fetch_add(counters + g * 64 + (quad & 63u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (WT_DBG(a) & 64) {
          wg_barrier();
          if ((s_ticket & 3u) == 3u) {      // the last of the four: combine
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const unsigned b0 = (blockIdx.x & 7u) + (blockIdx.x & ~31u);
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            for (unsigned j = 0; j < 4; ++j) {      // six 16-byte sc1 loads per thread and slab in flight, then one wait
              const size_t slab = (size_t)g * a.nslabs + b0 + 8u * j;
              const f32x4* pf = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.part_f) + slab * (2 * R * R)) + threadIdx.x;
              const f32x4* pr = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.part_r) + slab * (R * R)) + threadIdx.x;
              constexpr int NF = 2 * R * R / 4 / (64 * NWV), NR = R * R / 4 / (64 * NWV);
              f32x4 v[NF + NR > 0 ? NF + NR : 1];
#pragma unroll
              for (int i = 0; i < NF; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(pf + i * 64 * NWV) : "memory");
#pragma unroll
              for (int i = 0; i < NR; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[NF + i]) : "v"(pr + i * 64 * NWV) : "memory");
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
              for (int i = 0; i < NF + NR; ++i) { asm volatile("" : "+v"(v[i])); sum += v[i]; }
            }
            if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) counters[4095] = 1;      // (keeps the loads alive)
          }
        }
      }
#endif
      wg_barrier();
      stamp(30);
    }
    // ---- the group's bottom gradient
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      const int q = wave + NWV * m;
      if (q >= a.NT) continue;
      int hi = Wseg - 32 * q;
      hi = hi > 32 ? 32 : hi;
      if (hi <= 0) continue;
      tile_store_raw(img + (size_t)(32 * q) * LS, reinterpret_cast<T*>(a.g_out), q, hi, G[m]);
    }
    wg_barrier();
    if constexpr (ICG) {
      // ---- the input conv's weight gradient: every owned tile's G_0 is in the image now (tile_store_raw left it there;
      // rows beyond the owned ones are masked in the A fragment).  Wave w < R/16 takes the 16 channels 16 w..: per tile one
      // A fragment built from the audio -- row 0 / 1: audio[t-1-shift] as bf16 high / low part (fp32 mode: the value / 0),
      // row 2 / 3: audio[t-shift] likewise, row 4: ones, time steps in the tiles' order kordW -- one transposing read, one
      // MFMA.  It was a launch of its own reading the 2R bytes per sample this kernel has just written.
      {
        // every wave works: column block cblk (16 channels), tiles k = h, h + NH, ... (NH subsets, each with a partial slab
        // of its own: ic_part holds NH slabs per workgroup); four tiles per trip so that their audio loads fly together
        constexpr int NCB = R / 16, NH = NWV / NCB;
        static_assert(NWV % NCB == 0, "waves split evenly over the column blocks");
        int lw = lane;
        asm volatile("" : "+v"(lw));
        const int cblk = wave % NCB, h = wave / NCB;
        const T* cb = LdT16p<T>::base(img, LS, lw) + 16 * cblk;
        const int arow = lw & 15, kg = lw >> 4;
        const float* au = a.ic_audio + clip;
        const int ktn = (Wseg + 31) / 32;
        const int tb = jbase * a.st + r - a.ic_shift - (arow < 2 ? 1 : 0);      // time of position 0 under this row's tap
        const bool tap_row = arow < 4, one_row = arow == 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = h; k < ktn; k += NH) {
          Frag<T> af;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int pos = 32 * k + kordW(kg, e);
            const bool own = pos < Wseg;
            const int tt = tb + pos * a.st;
            const bool ld = tap_row && own && tt >= 0 && tt < a.Tlen;
            const float x = ld ? au[ld ? tt : 0] : 0.0f;
            float val;
            if (sizeof(T) == 2) {
              const float hi_ = (float)(bf16_t)x;
              val = (arow & 1) ? (x - hi_) : hi_;        // rows 0, 2: high part; rows 1, 3: what bf16 dropped
            } else {
              val = (arow & 1) ? 0.0f : x;
            }
            if (one_row) val = own ? 1.0f : 0.0f;
            af.set(e, val);
          }
          const Frag<T> bfv = LdT16p<T>::template load<LS>(cb + (size_t)(32 * k) * LS, 0);
          mma16(acc, af, bfv);
        }
        // D: lane l holds rows 4 (l >> 4) + 0..3 of column l & 15
        float* pp = a.ic_part + ((size_t)blockIdx.x * NH + h) * (3 * R) + 16 * cblk + (lw & 15);
        if (lw < 16) {
          pp[0] = (sit > 0 ? pp[0] : 0.0f) + (acc[0] + acc[1]);
          pp[R] = (sit > 0 ? pp[R] : 0.0f) + (acc[2] + acc[3]);
        } else if (lw < 32) {
          pp[2 * R] = (sit > 0 ? pp[2 * R] : 0.0f) + acc[0];
        }
      }
      wg_barrier();      // (the next segment zeroes the image)
    }
  }
}

unsigned long long* g_stamps = nullptr;
// tiles the forward / backward group kernels can hold per segment image (LDS and waves x tiles per wave)
template <typename T, int RT, int MAXT, int NWB, int NWV> int fwd_nt_max() {
  constexpr int R = 32 * RT, KS = R / 16, NW = RT * 2 * KS + RT * KS;
  const size_t fixed = (size_t)NWB * NW * 64 * sizeof(Frag<T>) + (size_t)NWB * 2 * R * 4;
  const size_t row_bytes = (size_t)RowStage<T>::stride(R) * sizeof(T);
  const int nt = (int)((kLdsBudget - fixed) / (32 * row_bytes));
  return nt > NWV * MAXT ? NWV * MAXT : nt;
}
template <typename T, int RT, int MAXT, int NWB, int NWV> int bwd_nt_max(bool wt) {
  constexpr int R = 32 * RT, KS = R / 16, NW = RT * 2 * KS + RT * KS;
  const size_t row_bytes = (size_t)RowStage<T>::stride(R) * sizeof(T);
  const size_t fixed = (size_t)NWB * NW * 64 * sizeof(Frag<T>) + (wt ? (size_t)kWtPadRows * row_bytes : 0);
  const int nt = (int)((kLdsBudget - fixed) / (32 * row_bytes));
  return nt > NWV * MAXT ? NWV * MAXT : nt;
}

template <typename T, int RT, int MAXT, int NWB, int NWV = 8, bool WDMA = true, bool STAMP = false, bool WT = false>
int launch_group_fwd(GroupFwdArgs& a, bool cond, int seg_rows, hipStream_t st) {
  if (a.ic_audio && (!WT || cond || STAMP))
    return set_error(SRWN_E_UNSUPPORTED, "residual_group_fwd_ic: built for the unconditioned weight-gradient-tile kernels (pass xT / cT)");
  constexpr int R = 32 * RT, KS = R / 16, NW = RT * 2 * KS + RT * KS;
  const size_t fixed = (size_t)NWB * NW * 64 * sizeof(Frag<T>) + (size_t)NWB * 2 * R * 4;
  const size_t row_bytes = (size_t)RowStage<T>::stride(R) * sizeof(T);
  const int nt_max = fwd_nt_max<T, RT, MAXT, NWB, NWV>();
  if (WT) a.H = (a.H + 31) / 32 * 32;      // whole halo tiles: the compute tiles line up with the weight-gradient tiles
  if (nt_max * 32 - a.H < 32) return set_error(SRWN_E_UNSUPPORTED, "residual_group_fwd: halo %d too large", a.H);
  const int J = (a.Tlen + a.st - 1) / a.st;
  choose_segments(J, &a.H, a.B, a.st, nt_max, seg_rows, &a.W, &a.NT, &a.nsub);
  if (WT) {
    if (a.W != seg_rows) return set_error(SRWN_E_SHAPE, "residual_group_fwd_wt: seg_rows %d does not fit (use srwn_group_wt_geometry)", seg_rows);
    a.KT = (a.W + 31) / 32;
  }
  const long long nseg = (long long)a.B * a.st * a.nsub;
  if (nseg > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "residual_group_fwd: too many segments");
  a.nseg = (int)nseg;
  const size_t sh = fixed + (size_t)a.NT * 32 * row_bytes;
  long long blocks = nseg < num_cus() ? nseg : num_cus();
  dim3 grid((unsigned)blocks), block(64 * NWV);
#define SRWN_GF(C)                                                                                              \
  {                                                                                                             \
    auto kfn = group_fwd_kernel<T, RT, C, MAXT, NWB, NWV, WDMA, STAMP, WT>;                                     \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);  \
    if (e != hipSuccess) return set_error((int)e, "residual_group_fwd: LDS %zu: %s", sh, hipGetErrorString(e)); \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, a);                                                            \
    return check_launch("residual_group_fwd");                                                                  \
  }
  if constexpr (WT && !STAMP) {
    if (a.ic_audio) {
      auto kfn = group_fwd_kernel<T, RT, false, MAXT, NWB, NWV, WDMA, false, true, true>;
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      if (e != hipSuccess) return set_error((int)e, "residual_group_fwd_ic: LDS %zu: %s", sh, hipGetErrorString(e));
      hipLaunchKernelGGL(kfn, grid, block, sh, st, a);
      return check_launch("residual_group_fwd_ic");
    }
  }
  if (cond) SRWN_GF(true) else SRWN_GF(false)
#undef SRWN_GF
}

template <typename T, int RT, int MAXT, int NWB, int NWV = 8, bool WT = false>
int launch_group_bwd(GroupBwdArgs& a, int seg_rows, hipStream_t st) {
  constexpr int R = 32 * RT, KS = R / 16, NW = RT * 2 * KS + RT * KS;
  const size_t row_bytes = (size_t)RowStage<T>::stride(R) * sizeof(T);
  const size_t fixed = (size_t)NWB * NW * 64 * sizeof(Frag<T>) + (WT ? (size_t)kWtPadRows * row_bytes : 0);
  const int nt_max = bwd_nt_max<T, RT, MAXT, NWB, NWV>(WT);
  if (nt_max * 32 - a.H < 32) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd: halo %d too large", a.H);
  const int J = (a.Tlen + a.st - 1) / a.st;
  choose_segments(J, &a.H, a.B, a.st, nt_max, seg_rows, &a.W, &a.NT, &a.nsub);
  const long long nseg = (long long)a.B * a.st * a.nsub;
  if (nseg > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "residual_group_bwd: too many segments");
  a.nseg = (int)nseg;
  const size_t sh = fixed + (size_t)a.NT * 32 * row_bytes;
  long long blocks = nseg < num_cus() ? nseg : num_cus();
  if (WT) {
    if (a.W != seg_rows) return set_error(SRWN_E_SHAPE, "residual_group_bwd_wt: seg_rows %d does not fit (use srwn_group_wt_geometry)", seg_rows);
    a.KT = (a.W + 31) / 32;
    if (blocks > a.nslabs) return set_error(SRWN_E_SHAPE, "residual_group_bwd_wt: %lld workgroups but %d partial slabs", blocks, a.nslabs);
  }
  dim3 grid((unsigned)blocks), block(64 * NWV);
  // segments of at most two tiles per wave (the halo-free residue-class groups: 16 tiles) run the two-tile body: 16 fewer
  // live registers for G, no third (empty) tile iteration
  constexpr int MT2 = MAXT > 2 ? 2 : MAXT;
  const bool two = MAXT > 2 && a.NT <= 2 * NWV;
#define SRWN_GB(D)                                                                                              \
  {                                                                                                             \
    auto kfn = two ? group_bwd_kernel<T, RT, D, MT2, NWB, NWV, WT> : group_bwd_kernel<T, RT, D, MAXT, NWB, NWV, WT>; \
    if constexpr (WT && sizeof(T) == 2) {                                                                          \
      if (a.part16) kfn = two ? group_bwd_kernel<T, RT, D, MT2, NWB, NWV, WT, false, true> : group_bwd_kernel<T, RT, D, MAXT, NWB, NWV, WT, false, true>; \
    }                                                                                                           \
    if constexpr (WT && D) {                                                                                     \
      if (a.ic_audio) {                                                                                          \
        if (sizeof(T) == 2 && !a.part16) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd_wt: the input conv gradient goes with bf16 partial blocks (part16) in bf16 mode"); \
        kfn = two ? group_bwd_kernel<T, RT, D, MT2, NWB, NWV, WT, false, sizeof(T) == 2, true> : group_bwd_kernel<T, RT, D, MAXT, NWB, NWV, WT, false, sizeof(T) == 2, true>; \
      }                                                                                                          \
    } else {                                                                                                     \
      if (a.ic_audio) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd_wt: the input conv gradient is built for stacks with a skip path"); \
    }                                                                                                           \
    if constexpr (WT && D && sizeof(T) == 2 && RT == 2) {                                                        \
      SRWN_DIAG_ONLY(if (g_stamps && !a.part16 && !a.ic_audio) { a.stamps = g_stamps; if (!(a.dbg & 32)) kfn = two ? group_bwd_kernel<T, RT, D, MT2, NWB, NWV, WT, true> : group_bwd_kernel<T, RT, D, MAXT, NWB, NWV, WT, true>; }) \
    }                                                                                                           \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);  \
    if (e != hipSuccess) return set_error((int)e, "residual_group_bwd: LDS %zu: %s", sh, hipGetErrorString(e)); \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, a);                                                            \
    return check_launch("residual_group_bwd");                                                                  \
  }
  if (a.dcs) SRWN_GB(true) else SRWN_GB(false)
#undef SRWN_GB
}

}  // namespace

namespace {
struct WtBwd {   // the extra operands of srwn_residual_group_bwd_wt
  const void* xT; const void* cT; int64_t wt_stride;
  void* part_f; void* part_r; float* part_bf; float* part_br; int nslabs; int write_all_g; int part16;
  const float* ic_audio; float* ic_part; int ic_shift;
};
}

static int group_bwd_impl(const void* g_top, void* g_out, void* df_out, const void* z, const void* dcs,
                          int64_t layer_stride, const void* const* wconvT, const void* const* wresT,
                          const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R,
                          int32_t K, int32_t seg_rows, int32_t dtype, void* stream, const WtBwd* wt) {
  if (B == 0 || T == 0 || nlayers == 0) return 0;
  if (!g_out || (!df_out && !wt) || !z || !wconvT || !wresT || !dilations)
    return set_error(SRWN_E_NULL, "residual_group_bwd: null pointer");
  if (wt && (!wt->xT || !wt->cT || !wt->part_f || !wt->part_r || !wt->part_bf || !wt->part_br))
    return set_error(SRWN_E_NULL, "residual_group_bwd_wt: null pointer");
  if (wt && (seg_rows < 1 || wt->nslabs < 1 || wt->wt_stride < 0))
    return set_error(SRWN_E_SHAPE, "residual_group_bwd_wt: seg_rows=%d nslabs=%d (take them from srwn_group_wt_geometry)", seg_rows, wt->nslabs);
  if (K != 2) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd: filter_width %d (only 2 is built)", K);
  if (R != 32 && R != 64) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd: dilation_channels %d (built: 32, 64)", R);
  if (nlayers < 0 || nlayers > kMaxGroup || B < 0 || T < 0 || seg_rows < 0)
    return set_error(SRWN_E_SHAPE, "residual_group_bwd: nlayers=%d (max %d) B=%d T=%d", nlayers, kMaxGroup, B, T);
  if (layer_stride < (int64_t)B * T * R) return set_error(SRWN_E_SHAPE, "residual_group_bwd: layer_stride %lld", (long long)layer_stride);
  if (!g_top && !dcs) return set_error(SRWN_E_SHAPE, "residual_group_bwd: no top gradient and no skip path: every gradient would be zero");
  GroupBwdArgs a;
  a.g_top = g_top; a.g_out = g_out; a.df_out = df_out; a.z = z; a.dcs = dcs; a.layer_stride = layer_stride;
  a.xT = a.cT = nullptr; a.wt_stride = 0; a.KT = 0; a.part_f = a.part_r = nullptr; a.part_bf = a.part_br = nullptr; a.nslabs = 0; a.write_all_g = 0; a.part16 = 0;
  a.ic_audio = nullptr; a.ic_part = nullptr; a.ic_shift = 0;
  a.dbg = 0;
  SRWN_DIAG_ONLY(static const int wt_dbg = [] { const char* e = getenv("SRWN_WT_DEBUG"); return e ? atoi(e) : 0; }(); a.dbg = wt_dbg;)
  a.stamps = nullptr;
  if (wt) {
    a.xT = wt->xT; a.cT = wt->cT; a.wt_stride = wt->wt_stride; a.part_f = wt->part_f; a.part_r = wt->part_r;
    a.part_bf = wt->part_bf; a.part_br = wt->part_br; a.nslabs = wt->nslabs; a.write_all_g = wt->write_all_g ? 1 : 0;
    a.part16 = wt->part16 ? 1 : 0;
    if (wt->ic_audio) {
      if (!wt->ic_part) return set_error(SRWN_E_NULL, "residual_group_bwd_wt: ic_audio without ic_partials");
      if (wt->ic_shift < 0 || wt->ic_shift > 1) return set_error(SRWN_E_SHAPE, "residual_group_bwd_wt: ic_shift %d", wt->ic_shift);
      a.ic_audio = wt->ic_audio; a.ic_part = wt->ic_part; a.ic_shift = wt->ic_shift;
    }
    if (a.part16 && dtype != SRWN_BF16) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd_wt: 16-bit partial blocks are the bf16 mode's (dtype %d)", dtype);
  }
  for (int g = 0; g < kMaxGroup; ++g) {
    const bool in = g < nlayers;
    a.wconvT[g] = in ? wconvT[g] : nullptr; a.wresT[g] = in ? wresT[g] : nullptr;
    a.sub[g] = 1;
    if (in && (!a.wconvT[g] || !a.wresT[g])) return set_error(SRWN_E_NULL, "residual_group_bwd: layer %d: null weights", g);
  }
  a.nl = nlayers; a.Tlen = T; a.B = B;
  if (group_geometry(dilations, nlayers, &a.st, a.sub, &a.H) != 0)
    return set_error(SRWN_E_SHAPE, "residual_group_bwd: dilations must be >= 1");
  for (int g = 0, acc = 0; g < kMaxGroup; ++g) { a.hb[g] = acc; acc += g < nlayers ? a.sub[g] : 0; }
  if (a.H > 31) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd: halo %d > 31 (sum of dilations / their gcd)", a.H);
  hipStream_t st = (hipStream_t)stream;
  if (wt) {
    if (a.H > 31) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd_wt: halo %d > 31 (sum of dilations / their gcd)", a.H);
    if (dtype == SRWN_BF16) {
      if (R == 32) return launch_group_bwd<bf16_t, 1, 3, 2, 8, true>(a, seg_rows, st);
      return launch_group_bwd<bf16_t, 2, 3, 2, 8, true>(a, seg_rows, st);
    } else if (dtype == SRWN_F32) {
      if (R == 32) return launch_group_bwd<float, 1, 1, 1, 8, true>(a, seg_rows, st);
      return launch_group_bwd<float, 2, 1, 1, 8, true>(a, seg_rows, st);
    }
    return set_error(SRWN_E_DTYPE, "residual_group_bwd_wt: dtype %d", dtype);
  }
  if (dtype == SRWN_BF16) {
    if (R == 32) return launch_group_bwd<bf16_t, 1, 3, 2>(a, seg_rows, st);
    // (twelve waves of two tiles, as the forward kernel runs, need 65 spilled registers here: 0.56 -> 0.73 ms per step)
    return launch_group_bwd<bf16_t, 2, 3, 2>(a, seg_rows, st);
  } else if (dtype == SRWN_F32) {
    if (R == 32) return launch_group_bwd<float, 1, 1, 1>(a, seg_rows, st);
    return launch_group_bwd<float, 2, 1, 1>(a, seg_rows, st);
  }
  return set_error(SRWN_E_DTYPE, "residual_group_bwd: dtype %d", dtype);
}

extern "C" int srwn_residual_group_bwd(const void* g_top, void* g_out, void* df_out, const void* z, const void* dcs,
                                       int64_t layer_stride, const void* const* wconvT, const void* const* wresT,
                                       const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R,
                                       int32_t K, int32_t seg_rows, int32_t dtype, void* stream) {
  return group_bwd_impl(g_top, g_out, df_out, z, dcs, layer_stride, wconvT, wresT, dilations, nlayers, B, T, R, K, seg_rows,
                        dtype, stream, nullptr);
}

extern "C" int srwn_residual_group_bwd_wt(const void* g_top, void* g_out, int32_t write_all_g, const void* z,
                                          const void* dcs, int64_t layer_stride, const void* xT, const void* cT,
                                          int64_t wt_layer_stride, const void* const* wconvT, const void* const* wresT,
                                          const int32_t* dilations, int32_t nlayers, void* part_f, void* part_r,
                                          float* part_bf, float* part_br, int32_t part16, const float* ic_audio,
                                          float* ic_partials, int32_t ic_shift, int32_t nslabs, int32_t B, int32_t T,
                                          int32_t R, int32_t K, int32_t seg_rows, int32_t dtype, void* stream) {
  const WtBwd wt{xT, cT, wt_layer_stride, part_f, part_r, part_bf, part_br, nslabs, write_all_g, part16, ic_audio, ic_partials, ic_shift};
  return group_bwd_impl(g_top, g_out, nullptr, z, dcs, layer_stride, wconvT, wresT, dilations, nlayers, B, T, R, K, seg_rows,
                        dtype, stream, &wt);
}

static int group_fwd_impl(const void* x0, void* x_out, void* z_out, int64_t layer_stride,
                          const void* const* wconv, const void* const* wres,
                          const float* const* bias_f, const float* const* bias_r,
                          const void* const* cond_next, int32_t cond_frames, int32_t pool_stride,
                          int32_t cond_row_stride, const int32_t* dilations, int32_t nlayers, int32_t B,
                          int32_t T, int32_t R, int32_t K, int32_t seg_rows, int32_t dtype,
                          void* stream, void* xT, void* cT, int64_t wt_stride, int32_t store_inner_x,
                          const float* ic_audio = nullptr, const float* ic_w = nullptr, const float* ic_b = nullptr,
                          int32_t ic_shift = 0) {
  if (B == 0 || T == 0 || nlayers == 0) return 0;
  const bool wt = xT != nullptr || cT != nullptr;
  if (ic_audio) {      // the input conv fused in: x0 is not read
    if (!ic_w || !ic_b) return set_error(SRWN_E_NULL, "residual_group_fwd_ic: null pointer");
    if (cond_next) return set_error(SRWN_E_UNSUPPORTED, "residual_group_fwd_ic: not built for the conditioned decoders");
    if (ic_shift < 0 || ic_shift > 1) return set_error(SRWN_E_SHAPE, "residual_group_fwd_ic: shift %d", ic_shift);
    x0 = ic_audio;      // (non-null for the checks below)
  }
  if (wt && (!xT || !cT || seg_rows < 1 || wt_stride < 0))
    return set_error(SRWN_E_SHAPE, "residual_group_fwd_wt: xT, cT and seg_rows (srwn_group_wt_geometry) are all required");
  if (!x0 || !x_out || !z_out || !wconv || !wres || !bias_f || !bias_r || !dilations)
    return set_error(SRWN_E_NULL, "residual_group_fwd: null pointer");
  if (K != 2) return set_error(SRWN_E_UNSUPPORTED, "residual_group_fwd: filter_width %d (only 2 is built)", K);
  if (R != 32 && R != 64) return set_error(SRWN_E_UNSUPPORTED, "residual_group_fwd: dilation_channels %d (built: 32, 64)", R);
  if (nlayers < 0 || nlayers > kMaxGroup || B < 0 || T < 0 || seg_rows < 0)
    return set_error(SRWN_E_SHAPE, "residual_group_fwd: nlayers=%d (max %d) B=%d T=%d", nlayers, kMaxGroup, B, T);
  if (layer_stride < (int64_t)B * T * R) return set_error(SRWN_E_SHAPE, "residual_group_fwd: layer_stride %lld", (long long)layer_stride);
  GroupFwdArgs a;
  a.safe_wait = safe_wait();
  a.x0 = x0; a.x_out = x_out; a.z_out = z_out; a.layer_stride = layer_stride;
  a.xT = xT; a.cT = cT; a.wt_stride = wt_stride; a.KT = 0; a.store_inner_x = store_inner_x ? 1 : 0;
  a.ic_audio = ic_audio; a.ic_w = ic_w; a.ic_b = ic_b; a.ic_shift = ic_shift;
  if (ic_audio) a.x0 = nullptr;
  bool any_cond = false;
  for (int g = 0; g < kMaxGroup; ++g) {
    const bool in = g < nlayers;
    a.wconv[g] = in ? wconv[g] : nullptr; a.wres[g] = in ? wres[g] : nullptr;
    a.bias_f[g] = in ? bias_f[g] : nullptr; a.bias_r[g] = in ? bias_r[g] : nullptr;
    a.cond[g] = (in && cond_next) ? cond_next[g] : nullptr;
    a.sub[g] = 1;
    if (in && (!a.wconv[g] || !a.wres[g] || !a.bias_f[g] || !a.bias_r[g]))
      return set_error(SRWN_E_NULL, "residual_group_fwd: layer %d: null weights", g);
    any_cond = any_cond || a.cond[g] != nullptr;
  }
  if (any_cond && (pool_stride < 1 || cond_row_stride < R || cond_row_stride % 8 || (int64_t)cond_frames * pool_stride < T))
    return set_error(SRWN_E_SHAPE, "residual_group_fwd: cond frames %d x pool %d < T %d", cond_frames, pool_stride, T);
  a.cond_frames = cond_frames; a.pool = pool_stride > 0 ? pool_stride : 1; a.cond_stride = cond_row_stride;
  a.nl = nlayers; a.Tlen = T; a.B = B; a.stamps = nullptr;
  if (group_geometry(dilations, nlayers, &a.st, a.sub, &a.H) != 0)
    return set_error(SRWN_E_SHAPE, "residual_group_fwd: dilations must be >= 1");
  if (a.H > 31) return set_error(SRWN_E_UNSUPPORTED, "residual_group_fwd: halo %d > 31 (sum of dilations / their gcd)", a.H);
  hipStream_t st = (hipStream_t)stream;
  // waves per workgroup (bf16, R = 64).  The kernel is bound by the instructions one wave can issue (one per ~5 cycles;
  // tools/micro/valubench.hip): twelve waves of <= 168 registers and two tiles each fill the VALU pipe that eight
  // waves of three tiles leave ~30 % idle (forward groups 412 -> 377 us per step); sixteen (128 registers) spill.
  // (SRWN_GF_WAVES=8 keeps the eight-wave body reachable: tests/test_gpu_group.py runs it against the twelve-wave one)
  static const int gf_waves = [] { const char* e = getenv("SRWN_GF_WAVES"); return (e && atoi(e) == 8) ? 8 : 12; }();
  const bool waves12 = gf_waves == 12;
  if (wt) {      // the weight-gradient tiles of x and c written as well (one instantiation per dtype / width)
    if (a.H > 31) return set_error(SRWN_E_UNSUPPORTED, "residual_group_fwd_wt: halo %d > 31 (sum of dilations / their gcd)", a.H);
    if (dtype == SRWN_BF16) {
      if (R == 32) return launch_group_fwd<bf16_t, 1, 3, 2, 8, true, false, true>(a, any_cond, seg_rows, st);
      // (eight waves of three tiles: twelve waves of 168 registers, the plain kernel's choice, spill 30-47 registers once the
      // tile stores are in the body: 0.43 vs 0.38 ms per step)
      SRWN_DIAG_ONLY(if (g_stamps) { a.stamps = g_stamps; return launch_group_fwd<bf16_t, 2, 3, 2, 8, true, true, true>(a, any_cond, seg_rows, st); })   // (tools/stamp_probe.py)
      return launch_group_fwd<bf16_t, 2, 3, 2, 8, true, false, true>(a, any_cond, seg_rows, st);
    } else if (dtype == SRWN_F32) {
      if (R == 32) return launch_group_fwd<float, 1, 1, 1, 8, true, false, true>(a, any_cond, seg_rows, st);
      return launch_group_fwd<float, 2, 1, 1, 8, true, false, true>(a, any_cond, seg_rows, st);
    }
    return set_error(SRWN_E_DTYPE, "residual_group_fwd_wt: dtype %d", dtype);
  }
  if (dtype == SRWN_BF16) {
    if (R == 32) return launch_group_fwd<bf16_t, 1, 3, 2>(a, any_cond, seg_rows, st);
    SRWN_DIAG_ONLY(if (g_stamps) { a.stamps = g_stamps; return launch_group_fwd<bf16_t, 2, 3, 2, 8, true, true>(a, any_cond, seg_rows, st); })
    // (the conditioned body needs 30 registers more than twelve waves leave it -- it spills them -- so it keeps eight
    // waves: the student's step 6.88 -> 6.77 ms)
    if (waves12 && !any_cond) return launch_group_fwd<bf16_t, 2, 2, 2, 12>(a, any_cond, seg_rows, st);
    return launch_group_fwd<bf16_t, 2, 3, 2>(a, any_cond, seg_rows, st);
  } else if (dtype == SRWN_F32) {
    if (R == 32) return launch_group_fwd<float, 1, 1, 1>(a, any_cond, seg_rows, st);
    return launch_group_fwd<float, 2, 1, 1>(a, any_cond, seg_rows, st);
  }
  return set_error(SRWN_E_DTYPE, "residual_group_fwd: dtype %d", dtype);
}

extern "C" int srwn_residual_group_fwd(const void* x0, void* x_out, void* z_out, int64_t layer_stride,
                                       const void* const* wconv, const void* const* wres,
                                       const float* const* bias_f, const float* const* bias_r,
                                       const void* const* cond_next, int32_t cond_frames, int32_t pool_stride,
                                       int32_t cond_row_stride, const int32_t* dilations, int32_t nlayers, int32_t B,
                                       int32_t T, int32_t R, int32_t K, int32_t seg_rows, int32_t dtype,
                                       void* stream) {
  return group_fwd_impl(x0, x_out, z_out, layer_stride, wconv, wres, bias_f, bias_r, cond_next, cond_frames, pool_stride,
                        cond_row_stride, dilations, nlayers, B, T, R, K, seg_rows, dtype, stream, nullptr, nullptr, 0, 1);
}

extern "C" int srwn_residual_group_fwd_wt(const void* x0, void* x_out, void* z_out, int64_t layer_stride, void* xT,
                                          void* cT, int64_t wt_layer_stride, int32_t store_inner_x, const void* const* wconv,
                                          const void* const* wres, const float* const* bias_f,
                                          const float* const* bias_r, const void* const* cond_next,
                                          int32_t cond_frames, int32_t pool_stride, int32_t cond_row_stride,
                                          const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R,
                                          int32_t K, int32_t seg_rows, int32_t dtype, void* stream) {
  if (!xT || !cT) return (B == 0 || T == 0 || nlayers == 0) ? 0 : set_error(SRWN_E_NULL, "residual_group_fwd_wt: null pointer");
  return group_fwd_impl(x0, x_out, z_out, layer_stride, wconv, wres, bias_f, bias_r, cond_next, cond_frames, pool_stride,
                        cond_row_stride, dilations, nlayers, B, T, R, K, seg_rows, dtype, stream, xT, cT, wt_layer_stride,
                        store_inner_x);
}

// The FIRST group of a stack with the stack's input conv fused in (model.py:40 / 172-173; K = 2 taps, 1 -> R channels,
// RightShift as `shift`): what srwn_causal_conv1d_fwd would have written to x0 is computed into the segment image, in
// the same arithmetic (bit-identical activations), and never reaches HBM.
extern "C" int srwn_residual_group_fwd_ic(const float* audio, const float* init_w, const float* init_b, int32_t shift,
                                          void* x_out, void* z_out, int64_t layer_stride, void* xT, void* cT,
                                          int64_t wt_layer_stride, int32_t store_inner_x, const void* const* wconv,
                                          const void* const* wres, const float* const* bias_f,
                                          const float* const* bias_r, const int32_t* dilations, int32_t nlayers,
                                          int32_t B, int32_t T, int32_t R, int32_t K, int32_t seg_rows, int32_t dtype,
                                          void* stream) {
  if (B == 0 || T == 0 || nlayers == 0) return 0;
  if (!audio) return set_error(SRWN_E_NULL, "residual_group_fwd_ic: null pointer");
  if (!xT || !cT) return set_error(SRWN_E_NULL, "residual_group_fwd_ic: xT and cT are required (the weight-gradient-tile kernels)");
  return group_fwd_impl(audio, x_out, z_out, layer_stride, wconv, wres, bias_f, bias_r, nullptr, 1, 1, R, dilations, nlayers,
                        B, T, R, K, seg_rows, dtype, stream, xT, cT, wt_layer_stride, xT ? store_inner_x : 1, audio, init_w,
                        init_b, shift);
}

// The segment cut both _wt kernels of a group must be given (seg_rows_in = 0: the library's choice), the weight-gradient
// tiles per segment, the elements of one layer's xT (= cT) buffer and the partial slabs per layer the backward launch writes.
extern "C" int srwn_group_wt_geometry(const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R,
                                      int32_t dtype, int32_t seg_rows_in, int32_t* seg_rows, int32_t* tiles_per_seg,
                                      int64_t* elems_per_layer, int32_t* nslabs) {
  if (!dilations || nlayers < 1 || nlayers > kMaxGroup || B < 1 || T < 1 || seg_rows_in < 0)
    return set_error(SRWN_E_SHAPE, "group_wt_geometry: nlayers=%d B=%d T=%d", nlayers, B, T);
  if (R != 32 && R != 64) return set_error(SRWN_E_UNSUPPORTED, "group_wt_geometry: dilation_channels %d (built: 32, 64)", R);
  int st = 0, H = 0, sub[kMaxGroup];
  if (group_geometry(dilations, nlayers, &st, sub, &H) != 0) return set_error(SRWN_E_SHAPE, "group_wt_geometry: dilations must be >= 1");
  if (H > 31) return set_error(SRWN_E_UNSUPPORTED, "group_wt_geometry: halo %d > 31", H);
  int ntf, ntb;
  if (dtype == SRWN_BF16) {
    ntf = R == 32 ? fwd_nt_max<bf16_t, 1, 3, 2, 8>() : fwd_nt_max<bf16_t, 2, 3, 2, 8>();
    ntb = R == 32 ? bwd_nt_max<bf16_t, 1, 3, 2, 8>(true) : bwd_nt_max<bf16_t, 2, 3, 2, 8>(true);
  } else if (dtype == SRWN_F32) {
    ntf = R == 32 ? fwd_nt_max<float, 1, 1, 1, 8>() : fwd_nt_max<float, 2, 1, 1, 8>();
    ntb = R == 32 ? bwd_nt_max<float, 1, 1, 1, 8>(true) : bwd_nt_max<float, 2, 1, 1, 8>(true);
  } else {
    return set_error(SRWN_E_DTYPE, "group_wt_geometry: dtype %d", dtype);
  }
  if (H > 0) ntf -= 1;                      // the forward kernel rounds its halo up to a whole tile
  const int nt = ntf < ntb ? ntf : ntb;
  if (nt * 32 - H < 32) return set_error(SRWN_E_UNSUPPORTED, "group_wt_geometry: halo %d too large", H);
  const int J = (T + st - 1) / st;
  int W, NT, nsub, Hc = H;
  choose_segments(J, &Hc, B, st, nt, seg_rows_in, &W, &NT, &nsub);
  const long long nseg = (long long)B * st * nsub;
  if (nseg > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "group_wt_geometry: too many segments");
  const int KT = (W + 31) / 32;
  if (seg_rows) *seg_rows = W;
  if (tiles_per_seg) *tiles_per_seg = KT;
  if (elems_per_layer) *elems_per_layer = (int64_t)nseg * KT * R * 32;
  if (nslabs) *nslabs = (int32_t)(nseg < num_cus() ? nseg : num_cus());
  return 0;
}

// how the layers of a stack are grouped for the fused kernels: greedy runs whose halo stays <= max_halo and whose
// length stays <= max_layers; writes the first layer of each group to starts[] (size >= nlayers + 1, terminated by
// nlayers) and returns the number of groups.
extern "C" int32_t srwn_group_plan(const int32_t* dilations, int32_t nlayers, int32_t max_halo, int32_t max_layers,
                                   int32_t* starts) {
  if (!dilations || !starts || nlayers < 0) return 0;
  if (max_layers < 1) max_layers = 1;
  if (max_layers > kMaxGroup) max_layers = kMaxGroup;
  auto gcd = [](int x, int y) { while (y) { int t = x % y; x = y; y = t; } return x; };
  int n = 0, l = 0;
  while (l < nlayers) {
    starts[n++] = l;
    int st = dilations[l] > 0 ? dilations[l] : 1, len = 1;
    while (l + len < nlayers && len < max_layers) {
      const int dn = dilations[l + len] > 0 ? dilations[l + len] : 1;
      const int st2 = gcd(st, dn);
      int H = 0;
      for (int i = 0; i <= len; ++i) H += (dilations[l + i] > 0 ? dilations[l + i] : 1) / st2;
      if (H > max_halo) break;
      st = st2;
      ++len;
    }
    l += len;
  }
  starts[n] = nlayers;
  return n;
}

// Diagnostic hook (no reference counterpart), live in the -DSRWN_DIAG build only (libsrwn_diag.so, build.py --diag):
// registers a device buffer of 1024 uint64; while one is registered, the bf16 R = 64 group kernels, the skip sum and the
// one-launch head run in their stamped instantiations and workgroup 0 appends (tag << 48 | shader clock) at its phase
// boundaries (tools/stamp_probe.py).  Pass NULL to return to the production instantiation.  The shipped library holds
// no stamped instantiation: it accepts NULL and refuses a buffer.
namespace srwn {
unsigned long long* debug_stamps() { return g_stamps; }
int safe_wait() {
  static const int v = [] { const char* e = getenv("SRWN_SAFE_WAIT"); return (e && atoi(e) != 0) ? 1 : 0; }();
  return v;
}
}  // namespace srwn
extern "C" int srwn_debug_stamp_buffer(void* device_buffer) {
#ifndef SRWN_DIAG
  if (device_buffer) return set_error(SRWN_E_UNSUPPORTED, "debug_stamp_buffer: this is the shipped library; build the diagnostic one (build.py --diag) and load it with SRWN_LIB_PATH");
#endif
  g_stamps = reinterpret_cast<unsigned long long*>(device_buffer);
  return 0;
}

// Cost-based cut of a stack into groups for a given problem size: minimises  sum over groups of
//   ceil(segments / CUs) * (fixed + layers * ceil(tiles per segment / 8))
// (8 waves per workgroup take one 32-step tile each per round; `fixed` = launch + segment prologue in tile-rounds), with
// the segment geometry the launchers will choose.  For 8 x 16000 steps and 3 x [1..512] it prefers {1,2,4} {8,16}
// {32..512} (16 tiles = two full rounds per layer everywhere) to {1..16} {32..512} (17 tiles = three rounds).
extern "C" int32_t srwn_group_plan_auto(const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R,
                                        int32_t dtype, int32_t max_layers, int32_t* starts) {
  if (!dilations || !starts || nlayers < 0) return 0;
  if (nlayers == 0) { starts[0] = 0; return 0; }
  if (max_layers < 1) max_layers = 1;
  if (max_layers > kMaxGroup) max_layers = kMaxGroup;
  const int rt = R / 32, ks = R / 16, nw = rt * 2 * ks + rt * ks;
  const size_t esz = dtype == SRWN_F32 ? 4 : 2, frag = 8 * esz * 64;
  const int nwb = dtype == SRWN_F32 ? 1 : 2, maxt = dtype == SRWN_F32 ? 1 : 3;
  const size_t row_bytes = (size_t)(R + 16 / esz) * esz;
  int nt_max = (int)((kLdsBudget - (size_t)nwb * nw * frag - (size_t)nwb * 2 * R * 4) / (32 * row_bytes));
  if (nt_max > 8 * maxt) nt_max = 8 * maxt;
  const double fixed = 1.5;
  std::vector<double> best(nlayers + 1, 1e300);
  std::vector<int> prev(nlayers + 1, -1);
  best[0] = 0.0;
  for (int l = 0; l < nlayers; ++l) {
    if (best[l] >= 1e300) continue;
    for (int len = 1; len <= max_layers && l + len <= nlayers; ++len) {
      int st = 0, H = 0, sub[kMaxGroup];
      if (group_geometry(dilations + l, len, &st, sub, &H) != 0) break;
      if (H > 31 || nt_max * 32 - H < 32) break;
      int W, NT, nsub;
      const int J = (T + st - 1) / st;
      choose_segments(J, &H, B, st, nt_max, 0, &W, &NT, &nsub);
      const double nseg = (double)B * st * nsub;
      const double passes = std::ceil(nseg / num_cus());
      const double c = passes * (fixed + len * std::ceil(NT / 8.0));
      if (best[l] + c < best[l + len]) { best[l + len] = best[l] + c; prev[l + len] = l; }
    }
  }
  std::vector<int> cuts;
  for (int l = nlayers; l > 0; l = prev[l]) cuts.push_back(prev[l]);
  int n = 0;
  for (int i = (int)cuts.size() - 1; i >= 0; --i) starts[n++] = cuts[i];
  starts[n] = nlayers;
  return n;
}
