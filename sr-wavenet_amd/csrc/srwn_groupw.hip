// Backward chain of a layer group AND the group's layer weight gradients in one launch (autodiff of ops.py:23-46 through
// the stacking loops model.py:42-47 / 176-189 / 428-453; TF builds it in AdamOptimizer.minimize, model.py:31).
// gfx950 (MI355X) only.  Segments, time decomposition and the chain itself: srwn_group.hip (group_bwd_kernel); this
// kernel computes the same chain
//   df_g = (Wr_g . (G_{g+1} sqrt(.5)) + dcs_g) * d(z sigmoid z)/df (z_g)
//   G_g  = G_{g+1} sqrt(.5) + sum_k Wf_g[k] . df_g[t + (K-1-k) sub_g]
// and, while df_g and G_{g+1} are on the chip, the time contractions srwn_wgrad_layers makes from their HBM copies:
//   dWr_g    = c_g^T . G_{g+1}            (c = z sigmoid z; the sqrt(.5) is applied by the final reduction)
//   dWf_g[k] = x_g[t - (K-1-k) d]^T . df_g      = sum over s of x_g[s] (x) df_g[s + (K-1-k) d]   (s = the x row)
//   dbr_g = colsum(G_{g+1}),  dbf_g = colsum(df_g)
// so df and G are never written (G only on request: the conditioned decoders sum it per frame) and x, z are read once:
// per layer and row 3 x 2R bytes read (z, dcs, x), nothing written, against 4 x 2R + 4 x 2R for the chain kernel
// followed by the weight-gradient pass.
//
// Why one wave per SIMD.  The weight gradients of a layer are 3 x R x R fp32 sums over time.  Split by OUTPUT over the
// waves, every wave would need the operand tiles of every other wave in LDS at once (3 more whole-segment images: 230 KB
// with the df image; split by rounds of 8 tiles it is still 173 KB); split by TIME (each wave sums its own tiles) the
// sums are 192 registers per wave at R = 64.  So the workgroup is 4 waves of up to 512 registers (one per SIMD, the
// register file is the same 512 KB per CU either way): a wave keeps the 192 accumulators, runs the chain for its own tiles
// and contracts the tiles it has just produced, transposed through its own LDS rows (ds_read_b64_tr_b16), so no operand
// crosses waves except df's shifted tap (the image the chain needs anyway).  At the end of a layer the four partial sums
// meet in LDS (fixed order: deterministic) and leave as one fp32 partial per segment and layer, summed over segments by
// srwn_reduce_partials.
//
// A segment's rows are summed where they are OWNED: rows of the halo belong to the neighbouring segment, so the staged
// G tile and x tile are zeroed beyond the owned rows (the chain itself uses the register copy / the df image).
#include <cmath>
#include <cstdlib>
#include "srwn_common.h"
#include "srwn_group.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;
using namespace srwn::grp;

namespace {

constexpr int kPadRows = 32;   // zero rows behind the df image: the shifted tap of the last tile reads up to H <= 31 rows beyond it

struct GroupBwdWArgs {
  const void* g_top;              // gradient wrt the group's top output [B,T,R], or null (= 0)
  void* g_out;                    // layer 0's input gradient at g_out; with write_all_g layer g's at g_out + g*layer_stride
  const void* x;                  // layer g's input at x + g*layer_stride
  const void* z;                  // z of layer g
  const void* dcs;                // Ws_g . dtotal of layer g, or null (no skip path)
  int64_t layer_stride;
  const void* wconvT[kMaxGroup];
  const void* wresT[kMaxGroup];
  float* part_f; float* part_r; float* part_bf; float* part_br;   // layer g, slab s at ((g*nslabs)+s)*n floats
  int nslabs, write_all_g;
  int sub[kMaxGroup];
  int nl, st, Tlen, B;
  int W, H, NT, nsub, nseg;
  unsigned long long* stamps;     // diagnostic builds only (srwn_debug_stamp_buffer)
};

// In-kernel time stamps (as in srwn_group.hip): lane 0 of waves 0 and 1 of workgroup 0, a buffer nothing else reads.
template <bool STAMP> struct WStamper {
  unsigned long long* p; int n;
  __device__ __forceinline__ void operator()(int) {}
};
template <> struct WStamper<true> {
  unsigned long long* p; int n;
  __device__ __forceinline__ void operator()(int tag) {
    if (p && n < 512) { p[n] = ((unsigned long long)tag << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull); ++n; }
  }
};

// Fragment of 16 tile rows (the contraction index: time) x 32 columns of a row-major [row][channel] LDS tile: lane l
// holds column col0 + (l & 31) and the eight rows row0 + kord(h, j), kord(h, j) = 2h + (j >> 2) + 4 (j & 3), h = l >> 5.
// Any bijection of the 16 rows serves as long as both operands of a product use the same one; this one makes the four
// rows a transposing read touches 4 apart, which for rows of 36 dwords (R = 64 bf16 + 16 B) puts the two 16-lane groups
// of a read cycle on disjoint banks (rows 0,1,2,3 as in srwn_wgrad2.hip would collide two-way at this stride).
template <typename T> struct LdT;
template <> struct LdT<bf16_t> {
  static __device__ __forceinline__ Frag<bf16_t> load(const bf16_t* tile, int stride, int row0, int col0, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1;
    const bf16_t* base = tile + (size_t)(row0 + 2 * h + 4 * q) * stride + col0 + 16 * (g & 1) + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + stride));
    Frag<bf16_t> f;
    f.v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    return f;
  }
};
template <> struct LdT<float> {
  static __device__ __forceinline__ Frag<float> load(const float* tile, int stride, int row0, int col0, int lane) {
    const int c = col0 + (lane & 31), h = lane >> 5;
    const float* base = tile + (size_t)(row0 + 2 * h) * stride + c;
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.set(j, base[(size_t)((j >> 2) + 4 * (j & 3)) * stride]);
    return f;
  }
};
__device__ __forceinline__ constexpr int kordT(int h, int j) { return 2 * h + (j >> 2) + 4 * (j & 3); }

// WGM: which sums this instantiation keeps (1 = dWr + dbr, 2 = dWf[0], 4 = dWf[1] + dbf).  bf16 keeps all three;
// fp32 fragments are twice the registers, so the exact-fp32 mode runs the launch once per sum (the chain is repeated).
template <typename T, int RT, bool DCS, int MAXT, int NWB, int NWV, int WGM, bool STAMP = false>
__global__ __launch_bounds__(64 * NWV) void group_bwdw_kernel(GroupBwdWArgs a) {
  constexpr int R = 32 * RT, K = 2, KS = R / 16, KT = 2;
  constexpr int NCONV = RT * K * KS, NRES = RT * KS, NW = NCONV + NRES;
  constexpr int LS = RowStage<T>::stride(R), VEC = RowStage<T>::VEC;
  constexpr int LPR = R / VEC, RPI = 64 / LPR, NI = 32 / RPI;
  constexpr int WBYTES = NW * 64 * (int)sizeof(Frag<T>);
  constexpr int WPIECES = WBYTES / 16, CPIECES = NCONV * 64 * (int)sizeof(Frag<T>) / 16;
  constexpr int NF = 16 * RT * RT;          // accumulator registers of one R x R sum
  constexpr int SH = NF / NWV;              // of which a wave finishes this many
  static_assert(NF % NWV == 0 && SH >= 1, "flush shape");
  typedef typename Raw4g<T>::type raw4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<T>* wbuf = reinterpret_cast<Frag<T>*>(smem);                       // [NWB][convT | resT]
  T* img = reinterpret_cast<T*>(smem + (size_t)NWB * WBYTES);             // [NT*32 + kPadRows][LS]: df of the layer
  float* dump = reinterpret_cast<float*>(smem + (size_t)NWB * WBYTES);    // end of a layer: the partial sums meet here

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int rsub = lane / LPR, piece = lane % LPR;
  T* xslot = img + ((size_t)a.NT * 32 + kPadRows) * LS + (size_t)wave * 32 * LS;   // the wave's own staging tile

  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto wload = [&](int g, int buf) {
    dma_image(a.wconvT[g], lds_base + buf * WBYTES, CPIECES * 16, wave, lane, NWV);
    dma_image(a.wresT[g], lds_base + buf * WBYTES + CPIECES * 16, (WPIECES - CPIECES) * 16, wave, lane, NWV);
  };
  auto zero_pad = [&]() {
    f32x4* p = reinterpret_cast<f32x4*>(img + (size_t)a.NT * 32 * LS);
    for (int i = tid; i < kPadRows * LS * (int)sizeof(T) / 16; i += 64 * NWV) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // every row of the image finite before anything is multiplied by a zeroed (not owned) row of the other operand:
  // 0 x NaN from stale LDS would poison a sum
  auto zero_image = [&]() {
    f32x4* p = reinterpret_cast<f32x4*>(img);
    for (int i = tid; i < (a.NT * 32 + kPadRows) * LS * (int)sizeof(T) / 16; i += 64 * NWV) p[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  Frag<T> ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones.set(j, 1.0f);
  WStamper<STAMP> stamp{nullptr, 0};
  if (STAMP && blockIdx.x == 0 && lane == 0 && wave < 2) stamp.p = a.stamps + wave * 512;
  stamp(1);

  int it = 0;
  for (int seg = blockIdx.x; seg < a.nseg; seg += gridDim.x, ++it) {
    const int per_clip = a.st * a.nsub;
    const int b = seg / per_clip;
    const int rem = seg - b * per_clip;
    int r, j0;
    if (a.nsub == 1) { r = rem; j0 = 0; }
    else { r = rem / a.nsub; j0 = (rem - r * a.nsub) * a.W; }
    const int Jr = (a.Tlen - r + a.st - 1) / a.st;
    const int Wseg = (Jr - j0) < a.W ? (Jr - j0) : a.W;
    const int jbase = j0;
    const size_t clip = (size_t)b * a.Tlen;
    auto grow = [&](int j) -> size_t {
      int jj = j < Jr ? j : Jr - 1;
      jj = jj < 0 ? 0 : jj;
      size_t t = (size_t)jj * a.st + r;
      t = t < (size_t)a.Tlen ? t : (size_t)a.Tlen - 1;
      return clip + t;
    };
    auto rows_load = [&](const T* base, int q, f32x4 (&v)[NI]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const f32x4*>(base + grow(jbase + 32 * q + i * RPI + rsub) * R + piece * VEC);
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
    // accumulator-layout registers -> the tile's rows; rows [0, hi) -> HBM as whole rows when gbase != null
    auto tile_put_raw = [&](T* trow, T* gbase, int q, int hi, const raw4 (&vals)[RT][4], bool mask_rows) {
      const bool own = !mask_rows || col < hi;
      wave_lds_order();
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<raw4*>(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half) = own ? vals[mt][gq] : Raw4g<T>::zero();
      wave_lds_order();
      if (gbase != nullptr && hi > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          int rr = i * RPI + rsub;
          rr = rr < hi ? rr : hi - 1;
          const f32x4 v = *reinterpret_cast<const f32x4*>(trow + (size_t)rr * LS + piece * VEC);
          *reinterpret_cast<f32x4*>(gbase + grow(jbase + 32 * q + rr) * R + piece * VEC) = v;
        }
      }
    };

    // ---- G of the group's top output (registers, accumulator layout), the top layer's weights
    raw4 G[MAXT][RT][4];
    const int gtop = a.nl - 1;
    if (it > 0) wg_barrier();             // the previous segment's last readers of the image are done
    zero_image();
    wg_barrier();
    wload(gtop, 0);
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
      }
    }
    dma_wait();
    // operands one tile ahead: z, dcs of a chain tile (zr, dr) or x of a tap tile (zr)
    f32x4 zr[NI], dr[NI];
    auto issueA = [&](int g, int q) {
      rows_load(reinterpret_cast<const T*>(a.z) + (size_t)g * a.layer_stride, q, zr);
      if (DCS) rows_load(reinterpret_cast<const T*>(a.dcs) + (size_t)g * a.layer_stride, q, dr);
    };
    auto issueX = [&](int g, int q) { rows_load(reinterpret_cast<const T*>(a.x) + (size_t)g * a.layer_stride, q, zr); };
    issueA(gtop, wave < a.NT ? wave : a.NT - 1);
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
      T* gprev = a.write_all_g ? reinterpret_cast<T*>(a.g_out) + (size_t)(g + 1) * a.layer_stride : nullptr;
      const bool haveg = (n > 0) || (a.g_top != nullptr);
      int hb = 0;
      for (int h = 0; h < g; ++h) hb += a.sub[h];
      int ntA = (Wseg + hb + d + 31) / 32, ntB = (Wseg + hb + 31) / 32;
      ntA = (a.H == 0 || ntA > a.NT) ? a.NT : ntA;
      ntB = (a.H == 0 || ntB > a.NT) ? a.NT : ntB;

      f32x16 accWr[RT][RT], accWf0[RT][RT], accWf1[RT][RT];
      float bsr[RT], bsf[RT];
#pragma unroll
      for (int mi = 0; mi < RT; ++mi) {
        bsr[mi] = 0.0f; bsf[mi] = 0.0f;
#pragma unroll
        for (int nb = 0; nb < RT; ++nb)
#pragma unroll
          for (int e = 0; e < 16; ++e) { accWr[mi][nb][e] = 0.0f; accWf0[mi][nb][e] = 0.0f; accWf1[mi][nb][e] = 0.0f; }
      }

      stamp(20);
      // ---- phase A: df of every owned tile; dWr, dbr from (c, G_{g+1}) of the rows the segment owns
#pragma unroll 1
      for (int m = 0; m < MAXT; ++m) {
        const int q = wave + NWV * m;
        if (q < ntA) {
          T* trow = img + (size_t)(32 * q) * LS;
          const bool ok = (jbase + 32 * q + col) < Jr;
          int hi = Wseg - 32 * q;
          hi = hi > 32 ? 32 : hi;
          hi = hi < 0 ? 0 : hi;
          const bool wgr = (WGM & 1) && haveg && hi > 0;
          stamp(21);
          Frag<T> gT[KT][RT];
#pragma unroll
          for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int nb = 0; nb < RT; ++nb) gT[ks][nb] = zero_frag<T>();
          if (wgr || (gprev != nullptr && n > 0 && hi > 0)) {
            tile_put_raw(trow, n > 0 ? gprev : nullptr, q, hi, G[0], true);
            if (wgr) {
#pragma unroll
              for (int ks = 0; ks < KT; ++ks)
#pragma unroll
                for (int nb = 0; nb < RT; ++nb) {
                  gT[ks][nb] = LdT<T>::load(trow, LS, 16 * ks, 32 * nb, lane);
                  bsr[nb] = frag_dot(bsr[nb], gT[ks][nb], ones);
                }
            }
          }
          if (STAMP) { asm volatile("" :: "v"(gT[0][0].get(0)), "v"(bsr[0])); stamp(22); }
          raw4 zz[RT][4], dc0[RT][4];
          rows_put(trow, zr);
          acc_get(trow, zz);
          if (wgr) {
#pragma unroll
            for (int ks = 0; ks < KT; ++ks)
#pragma unroll
              for (int mi = 0; mi < RT; ++mi) {
                Frag<T> cT = LdT<T>::load(trow, LS, 16 * ks, 32 * mi, lane);
#pragma unroll
                for (int j = 0; j < 8; ++j) cT.set(j, gate_of_z<T>(cT.get(j)));
#pragma unroll
                for (int nb = 0; nb < RT; ++nb) mma(accWr[mi][nb], cT, gT[ks][nb]);
              }
          }
          if (STAMP) { asm volatile("" :: "v"(accWr[0][0][0]), "v"(Raw4g<T>::get(zz[0][0], 0))); stamp(23); }
          if (DCS) { rows_put(trow, dr); acc_get(trow, dc0); }
          else wave_lds_order();
          if (STAMP) { if (DCS) asm volatile("" :: "v"(Raw4g<T>::get(dc0[0][0], 0))); stamp(24); }
          // next operands: the wave's next chain tile, or x of its first tap tile
          if (q + NWV < ntA) issueA(g, q + NWV);
          else if (wave < ntB) issueX(g, wave);
          f32x16 accC[RT];
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
              for (int e = 0; e < 4; ++e) accC[mt][4 * gq + e] = (DCS && ok) ? Raw4g<T>::get(dc0[mt][gq], e) : 0.0f;
          if (haveg) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              Frag<T> bfr;
#pragma unroll
              for (int jj = 0; jj < 8; ++jj)
                bfr.set(jj, (ok ? Raw4g<T>::get(G[0][s >> 1][2 * (s & 1) + (jj >> 2)], jj & 3) : 0.0f) * kSqrtHalf);
#pragma unroll
              for (int mt = 0; mt < RT; ++mt) mma(accC[mt], lds_res[(mt * KS + s) * 64 + lane], bfr);
            }
          }
          // df of the tile into the image (the shifted tap of phase B and the time contraction read it there)
          wave_lds_order();
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              float dv[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) dv[e] = accC[mt][4 * gq + e] * dgate_df<T>(Raw4g<T>::get(zz[mt][gq], e));
              store4(trow + (size_t)col * LS + 32 * mt + 8 * gq + 4 * half, dv[0], dv[1], dv[2], dv[3]);
            }
          if (STAMP) { wave_lds_order(); stamp(25); }
        }
        {   // the next tile's G moves into G[0]
          raw4 t0[RT][4];
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) t0[mt][gq] = G[0][mt][gq];
#pragma unroll
          for (int mm = 0; mm + 1 < MAXT; ++mm)
#pragma unroll
            for (int mt = 0; mt < RT; ++mt)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) G[mm][mt][gq] = G[mm + 1][mt][gq];
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) G[MAXT - 1][mt][gq] = t0[mt][gq];
        }
      }
      stamp(26);
      wg_barrier();
      stamp(27);

      // ---- phase B: G_g = G_{g+1} sqrt(.5) + taps of df_g; dWf, dbf from (x_g, df_g) of the owned rows
#pragma unroll 1
      for (int m = 0; m < MAXT; ++m) {
        const int q = wave + NWV * m;
        if (q < ntB) {
          const int i0 = 32 * q + col;
          const int j = jbase + i0;
          const bool ok = j < Jr;
          const bool ok_d = (j + d) < Jr;
          int hi = Wseg - 32 * q;
          hi = hi > 32 ? 32 : hi;
          hi = hi < 0 ? 0 : hi;
          stamp(31);
          if ((WGM & 6) && hi > 0) {
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              const int rr = i * RPI + rsub;
              *reinterpret_cast<f32x4*>(xslot + (size_t)rr * LS + piece * VEC) = rr < hi ? zr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            wave_lds_order();
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
              Frag<T> xT[RT], mk;
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) mk.set(jj, (16 * ks + kordT(half, jj)) < hi ? 1.0f : 0.0f);
#pragma unroll
              for (int mi = 0; mi < RT; ++mi) xT[mi] = LdT<T>::load(xslot, LS, 16 * ks, 32 * mi, lane);
#pragma unroll
              for (int nb = 0; nb < RT; ++nb) {
                if (WGM & 4) {
                  const Frag<T> f0 = LdT<T>::load(img, LS, 32 * q + 16 * ks, 32 * nb, lane);
                  bsf[nb] = frag_dot(bsf[nb], f0, mk);
#pragma unroll
                  for (int mi = 0; mi < RT; ++mi) mma(accWf1[mi][nb], xT[mi], f0);
                }
                if (WGM & 2) {
                  const Frag<T> fd = LdT<T>::load(img, LS, 32 * q + d + 16 * ks, 32 * nb, lane);
#pragma unroll
                  for (int mi = 0; mi < RT; ++mi) mma(accWf0[mi][nb], xT[mi], fd);
                }
              }
            }
          }
          if (STAMP) { asm volatile("" :: "v"(accWf0[0][0][0]), "v"(accWf1[0][0][0]), "v"(bsf[0])); stamp(32); }
          // next operands: x of the wave's next tap tile, or z / dcs of its first chain tile of the layer below
          if (q + NWV < ntB) issueX(g, q + NWV);
          else if (g > 0) issueA(g - 1, wave);
          int src = i0 + d;
          src = src < a.NT * 32 ? src : a.NT * 32 - 1;
          f32x16 accG[RT];
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              accG[mt][e] = (haveg && ok) ? Raw4g<T>::get(G[0][mt][e >> 2], e & 3) * kSqrtHalf : 0.0f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const Frag<T> f0 = load_nat(img + (size_t)src * LS + 16 * ks + 8 * half);
            const Frag<T> bfr = ok_d ? f0 : zero_frag<T>();
#pragma unroll
            for (int mt = 0; mt < RT; ++mt) mma(accG[mt], lds_conv[(mt * (K * KS) + ks) * 64 + lane], bfr);
          }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const Frag<T> f1 = load_nat(img + (size_t)i0 * LS + 16 * ks + 8 * half);
            const Frag<T> bfr = ok ? f1 : zero_frag<T>();
#pragma unroll
            for (int mt = 0; mt < RT; ++mt) mma(accG[mt], lds_conv[(mt * (K * KS) + KS + ks) * 64 + lane], bfr);
          }
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
              G[0][mt][gq] = Raw4g<T>::pack(accG[mt][4 * gq], accG[mt][4 * gq + 1], accG[mt][4 * gq + 2], accG[mt][4 * gq + 3]);
          if (STAMP) { asm volatile("" :: "v"(G[0][0][0])); stamp(33); }
        }
        {
          raw4 t0[RT][4];
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) t0[mt][gq] = G[0][mt][gq];
#pragma unroll
          for (int mm = 0; mm + 1 < MAXT; ++mm)
#pragma unroll
            for (int mt = 0; mt < RT; ++mt)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) G[mm][mt][gq] = G[mm + 1][mt][gq];
#pragma unroll
          for (int mt = 0; mt < RT; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) G[MAXT - 1][mt][gq] = t0[mt][gq];
        }
      }
      stamp(36);
      if (NWB == 2 && g > 0) dma_wait();
      wg_barrier();
      stamp(37);

      // ---- the layer's partial sums: the waves' registers meet in LDS (the image is dead), one partial per segment
      // leaves in the layout srwn_wgrad_layers writes (so srwn_reduce_partials finishes both the same way)
      auto flush = [&](const f32x16 (&acc)[RT][RT], float* dst, const float (*bias)[RT], float* dstb) {
#pragma unroll
        for (int mi = 0; mi < RT; ++mi)
#pragma unroll
          for (int nb = 0; nb < RT; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) dump[((size_t)wave * NF + (mi * RT + nb) * 16 + e) * 64 + lane] = acc[mi][nb][e];
        float* bd = dump + (size_t)NWV * NF * 64;
        if (bias) {
#pragma unroll
          for (int nb = 0; nb < RT; ++nb) bd[(wave * RT + nb) * 64 + lane] = (*bias)[nb];
        }
        stamp(40);
        wg_barrier();
        stamp(41);
#pragma unroll
        for (int s = 0; s < SH; ++s) {
          const int f = wave * SH + s;
          float t = 0.0f;
#pragma unroll
          for (int w = 0; w < NWV; ++w) t += dump[((size_t)w * NF + f) * 64 + lane];
          const int blk = f >> 4, e = f & 15, mi = blk / RT, nb = blk - mi * RT;
          float* p = dst + (size_t)(32 * mi + crow(e, half)) * R + 32 * nb + col;
          if (it > 0) t += *p;
          *p = t;
        }
        if (bias && tid < R) {
          const int nb = tid >> 5, cl = tid & 31;
          float t = 0.0f;
#pragma unroll
          for (int w = 0; w < NWV; ++w) t += bd[(w * RT + nb) * 64 + cl] + bd[(w * RT + nb) * 64 + cl + 32];
          if (it > 0) t += dstb[tid];
          dstb[tid] = t;
        }
        stamp(42);
        wg_barrier();
        stamp(43);
      };
      const size_t ls = (size_t)g * a.nslabs + blockIdx.x;
      if (WGM & 1) flush(accWr, a.part_r + ls * (R * R), &bsr, a.part_br + ls * R);
      if (WGM & 2) flush(accWf0, a.part_f + ls * (2 * R * R), nullptr, nullptr);
      if (WGM & 4) flush(accWf1, a.part_f + ls * (2 * R * R) + R * R, &bsf, a.part_bf + ls * R);
      // the dump leaves fp32 bit patterns in the image (any of which may read as a bf16 NaN): rows a later layer reads
      // without having rewritten them (tiles beyond its shrunken halo, the zero rows) only ever meet zeroed operand rows,
      // but 0 x NaN is NaN -> clear.  (The barrier between the phases orders this before the readers.)
      zero_image();
      stamp(38);
    }
    // ---- the group's bottom gradient
#pragma unroll
    for (int m = 0; m < MAXT; ++m) {
      const int q = wave + NWV * m;
      if (q >= a.NT) continue;
      int hi = Wseg - 32 * q;
      hi = hi > 32 ? 32 : hi;
      if (hi <= 0) continue;
      tile_put_raw(img + (size_t)(32 * q) * LS, reinterpret_cast<T*>(a.g_out), q, hi, G[m], false);
    }
  }
}

template <typename T, int RT, int MAXT, int NWB, int NWV, int WGM, bool STAMP = false>
int launch_group_bwdw(GroupBwdWArgs& a, int seg_rows, hipStream_t st) {
  constexpr int R = 32 * RT, KS = R / 16, NW = RT * 2 * KS + RT * KS, NF = 16 * RT * RT;
  const size_t fixed = (size_t)NWB * NW * 64 * sizeof(Frag<T>);
  const size_t row_bytes = (size_t)RowStage<T>::stride(R) * sizeof(T);
  const size_t extra = (size_t)(kPadRows + NWV * 32) * row_bytes;          // zero rows + one staging tile per wave
  const size_t dump_bytes = ((size_t)NWV * NF + (size_t)NWV * RT) * 64 * sizeof(float);
  int nt_max = (int)((kLdsBudget - fixed - extra) / (32 * row_bytes));
  if (nt_max > NWV * MAXT) nt_max = NWV * MAXT;
  if (fixed + dump_bytes > (size_t)kLdsBudget || nt_max * 32 - a.H < 32)
    return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd_wgrad: halo %d too large", a.H);
  const int J = (a.Tlen + a.st - 1) / a.st;
  choose_segments(J, &a.H, a.B, a.st, nt_max, seg_rows, &a.W, &a.NT, &a.nsub);
  const long long nseg = (long long)a.B * a.st * a.nsub;
  if (nseg > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "residual_group_bwd_wgrad: too many segments");
  a.nseg = (int)nseg;
  size_t body = (size_t)a.NT * 32 * row_bytes + extra;
  if (body < dump_bytes) body = dump_bytes;
  const size_t sh = fixed + body;
  long long blocks = nseg < num_cus() ? nseg : num_cus();
  if (blocks > a.nslabs)
    return set_error(SRWN_E_SHAPE, "residual_group_bwd_wgrad: %lld workgroups but room for %d partial slabs (srwn_group_wgrad_slabs)", blocks, a.nslabs);
  dim3 grid((unsigned)blocks), block(64 * NWV);
#define SRWN_GBW(D)                                                                                             \
  {                                                                                                             \
    auto kfn = group_bwdw_kernel<T, RT, D, MAXT, NWB, NWV, WGM, STAMP>;                                         \
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);  \
    if (e != hipSuccess) return set_error((int)e, "residual_group_bwd_wgrad: LDS %zu: %s", sh, hipGetErrorString(e)); \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, a);                                                            \
    return check_launch("residual_group_bwd_wgrad");                                                            \
  }
  if (a.dcs) SRWN_GBW(true) else SRWN_GBW(false)
#undef SRWN_GBW
}

}  // namespace

// partial slabs a launch may write per layer (= the most workgroups it starts): size part_* with it
extern "C" int32_t srwn_group_wgrad_slabs(void) { return (int32_t)num_cus(); }

extern "C" int srwn_residual_group_bwd_wgrad(const void* g_top, void* g_out, int32_t write_all_g, const void* x,
                                             const void* z, const void* dcs, int64_t layer_stride,
                                             const void* const* wconvT, const void* const* wresT,
                                             const int32_t* dilations, int32_t nlayers, float* part_f, float* part_r,
                                             float* part_bf, float* part_br, int32_t nslabs, int32_t B, int32_t T,
                                             int32_t R, int32_t K, int32_t seg_rows, int32_t dtype, void* stream) {
  if (B == 0 || T == 0 || nlayers == 0) return 0;
  if (!g_out || !x || !z || !wconvT || !wresT || !dilations || !part_f || !part_r || !part_bf || !part_br)
    return set_error(SRWN_E_NULL, "residual_group_bwd_wgrad: null pointer");
  if (K != 2) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd_wgrad: filter_width %d (only 2 is built)", K);
  if (R != 32 && R != 64) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd_wgrad: dilation_channels %d (built: 32, 64)", R);
  if (nlayers < 0 || nlayers > kMaxGroup || B < 0 || T < 0 || seg_rows < 0 || nslabs < 1)
    return set_error(SRWN_E_SHAPE, "residual_group_bwd_wgrad: nlayers=%d (max %d) B=%d T=%d nslabs=%d", nlayers, kMaxGroup, B, T, nslabs);
  if (layer_stride < (int64_t)B * T * R) return set_error(SRWN_E_SHAPE, "residual_group_bwd_wgrad: layer_stride %lld", (long long)layer_stride);
  if (!g_top && !dcs) return set_error(SRWN_E_SHAPE, "residual_group_bwd_wgrad: no top gradient and no skip path: every gradient would be zero");
  GroupBwdWArgs a;
  a.g_top = g_top; a.g_out = g_out; a.x = x; a.z = z; a.dcs = dcs; a.layer_stride = layer_stride;
  a.part_f = part_f; a.part_r = part_r; a.part_bf = part_bf; a.part_br = part_br;
  a.nslabs = nslabs; a.write_all_g = write_all_g ? 1 : 0; a.stamps = nullptr;
  for (int g = 0; g < kMaxGroup; ++g) {
    const bool in = g < nlayers;
    a.wconvT[g] = in ? wconvT[g] : nullptr; a.wresT[g] = in ? wresT[g] : nullptr;
    a.sub[g] = 1;
    if (in && (!a.wconvT[g] || !a.wresT[g])) return set_error(SRWN_E_NULL, "residual_group_bwd_wgrad: layer %d: null weights", g);
  }
  a.nl = nlayers; a.Tlen = T; a.B = B;
  if (group_geometry(dilations, nlayers, &a.st, a.sub, &a.H) != 0)
    return set_error(SRWN_E_SHAPE, "residual_group_bwd_wgrad: dilations must be >= 1");
  if (a.H >= kPadRows) return set_error(SRWN_E_UNSUPPORTED, "residual_group_bwd_wgrad: halo %d > %d (sum of dilations / their gcd)", a.H, kPadRows - 1);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SRWN_BF16) {
    if (R == 32) return launch_group_bwdw<bf16_t, 1, 3, 2, 8, 7>(a, seg_rows, st);
    static const int dbg_mask = [] { const char* e = getenv("SRWN_GW_MASK"); return e ? atoi(e) : 7; }();   // timing experiments
    if (debug_stamps()) { a.stamps = debug_stamps(); return launch_group_bwdw<bf16_t, 2, 5, 2, 4, 7, true>(a, seg_rows, st); }
    if (dbg_mask == 0) return launch_group_bwdw<bf16_t, 2, 5, 2, 4, 0>(a, seg_rows, st);
    if (dbg_mask == 1) return launch_group_bwdw<bf16_t, 2, 5, 2, 4, 1>(a, seg_rows, st);
    if (dbg_mask == 6) return launch_group_bwdw<bf16_t, 2, 5, 2, 4, 6>(a, seg_rows, st);
    return launch_group_bwdw<bf16_t, 2, 5, 2, 4, 7>(a, seg_rows, st);
  } else if (dtype == SRWN_F32) {
    // exact-fp32 mode: one sum per launch (fragments are twice the registers); the chain and g_out repeat identically
    const int H0 = a.H;
    for (int pass = 0; pass < 3; ++pass) {
      a.H = H0;
      int rc;
      if (R == 32) {
        rc = pass == 0 ? launch_group_bwdw<float, 1, 2, 1, 4, 1>(a, seg_rows, st)
           : pass == 1 ? launch_group_bwdw<float, 1, 2, 1, 4, 2>(a, seg_rows, st)
                       : launch_group_bwdw<float, 1, 2, 1, 4, 4>(a, seg_rows, st);
      } else {
        rc = pass == 0 ? launch_group_bwdw<float, 2, 2, 1, 4, 1>(a, seg_rows, st)
           : pass == 1 ? launch_group_bwdw<float, 2, 2, 1, 4, 2>(a, seg_rows, st)
                       : launch_group_bwdw<float, 2, 2, 1, 4, 4>(a, seg_rows, st);
      }
      if (rc != 0) return rc;
    }
    return 0;
  }
  return set_error(SRWN_E_DTYPE, "residual_group_bwd_wgrad: dtype %d", dtype);
}
