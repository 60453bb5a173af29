// Backward kernels: fused residual-layer data gradient, time-contraction weight gradients,
// deterministic partial reduction.  gfx950 (MI355X) only.
#include <cstdlib>
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

// ------------------------------------------------------------------------------------------
// fused residual layer backward (autodiff of ops.py:23-46), one launch per layer l, top to bottom.
//   UP   : G_{l+1}[t] = G_{l+2}[t]*sqrt(.5) + sum_k Wf_{l+1}[k] . df_{l+1}[t + (K-1-k) d_{l+1}]   -> g_out
//          (data gradient of the layer above; anti-causal taps, zero beyond T)
//   DOWN : dc = Wr_l . (G_{l+1} sqrt(.5)) + dcs_l ;  df_l = dc * d(z sigmoid(z))/df                -> df_out
//          dcs_l = Ws_l . dtotal comes precomputed from srwn_skip_dgrad_all (DCS), or is computed
//          here from dtotal (legacy path for shapes that kernel does not cover).
//   The UP accumulator tile is the B operand of the Wr product (no LDS / HBM round trip).
//   Flags: layer L-1 runs DOWN only (G_L = 0: the last dense output is unused, model.py:45-50);
//          the call below layer 0 runs UP only (gradient wrt the input conv output).
//          Flow stacks (model.py:415-453) have no skip path (SK = 2: dc = Wr . G sqrt(.5) only) and DO use the
//          last dense output: their layer L-1 runs with UPM = 2, reading G_L from g_out instead of building it.
//   Same execution shape as the forward kernel: persistent waves over 32-step tiles, the next tile's
//   operands in flight (ping-pong register sets, unconditional clamped loads), whole-row stores via LDS.
// ------------------------------------------------------------------------------------------
struct LayerBwdArgs {
  const void* g_in;      // G_{l+2} [B,T,R] (GIN)
  const void* df_up;     // df_{l+1} [B,T,R]
  const void* wconvT;    // packed [R/32][K*R/16] natural: rows = in channel i, k = tap*R + o
  void* g_out;           // G_{l+1}
  const void* wresT;     // packed [R/32][R/16] permuted: rows = n, k = m  (Wr[n][m])
  const void* wskipT;    // packed [R/32][S/16] natural: rows = n, k = s   (Ws[n][s])   (!DCS)
  const void* dtotal;    // [B*T, S]                                                    (!DCS)
  const void* dcs;       // dcs_l [B,T,R]                                               (DCS)
  const void* z;         // z_l [B,T,R]
  void* df_out;          // df_l
  int Tlen, dil_up, S, ntb, ntiles;
};

template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> {
  typedef bf16x4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
  static __device__ __forceinline__ float get(const type& v, int e) { return (float)v[e]; }
};
template <> struct Raw4<float> {
  typedef f32x4 type;
  static __device__ __forceinline__ type load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ float get(const type& v, int e) { return v[e]; }
};

template <typename T, int RT, int K, int UPM, bool GIN, bool DOWN, int SK>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && SK != 0) ? 2 : 1) void layer_bwd_kernel(LayerBwdArgs a) {
  constexpr int R = 32 * RT, KS = R / 16;
  constexpr bool UP = UPM == 1, HAVEG = UPM != 0, DCS = SK == 1, LEGACY = SK == 0;
  typedef typename Raw4<T>::type raw4;
  const int KSS = a.S / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Frag<T>* lds_conv = reinterpret_cast<Frag<T>*>(smem);                    // [RT*K*KS][64]   (UP)
  Frag<T>* lds_res = lds_conv + (UP ? RT * K * KS * 64 : 0);               // [RT*KS][64]     (HAVEG && DOWN)
  Frag<T>* lds_skip = lds_res + ((HAVEG && DOWN) ? RT * KS * 64 : 0);      // [RT*KSS][64]    (DOWN && LEGACY)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ROWS (bf16, precomputed-dcs and flow paths): operands are fetched as WHOLE ROWS (16 B per lane, 8 lines per
  // instruction) into registers two tiles ahead, dropped into a wave-private padded LDS tile and read back in
  // fragment / accumulator layout.  Fetching them directly in those layouts touches 32 cache lines per
  // instruction and caps the kernel at 3.9 TB/s (tools/micro/membench3.hip: 25.4 us vs 19.1 us per launch).
  constexpr bool ROWS = sizeof(T) == 2 && !LEGACY;
  constexpr int LS = RowStage<T>::stride(R), VEC = RowStage<T>::VEC;
  constexpr int LPR = R / VEC, RPI = 64 / LPR, NI = 32 / RPI;
  T* stage_all = reinterpret_cast<T*>(lds_skip + ((DOWN && LEGACY) ? RT * KSS * 64 : 0));
  T* stage = stage_all + wave * (32 * LS);
  if (UP) lds_dma_copy(a.wconvT, lds_conv, RT * K * KS * 64 * (int)sizeof(Frag<T>), wave, lane, 4);
  if (HAVEG && DOWN) lds_dma_copy(a.wresT, lds_res, RT * KS * 64 * (int)sizeof(Frag<T>), wave, lane, 4);
  if (DOWN && LEGACY) lds_dma_copy(a.wskipT, lds_skip, RT * KSS * 64 * (int)sizeof(Frag<T>), wave, lane, 4);
  // (barrier after the first tile's loads are issued: one round trip for weights + first operands)

  const int col = lane & 31, half = lane >> 5;
  struct Tile {
    raw4 gin[RT][4];
    Frag<T> dfu[K][KS];
    raw4 dcs[RT][4];
    raw4 zz[RT][4];
  };

  auto load_tile = [&](int tile_, Tile& t) {
    const int tile = tile_ < a.ntiles ? tile_ : a.ntiles - 1;
    const int b = tile / a.ntb;
    const int tc = (tile - b * a.ntb) * 32 + col;
    const int tcc = tc < a.Tlen ? tc : a.Tlen - 1;
    const size_t boff = (size_t)b * a.Tlen;
    const size_t rowi = boff + tcc;
    if (UPM == 2) {   // G_{l+1} already lies in g_out (flow head gradient)
      const T* gin = reinterpret_cast<const T*>(a.g_out) + rowi * R;
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) t.gin[mt][g] = Raw4<T>::load(gin + 32 * mt + 8 * g + 4 * half);
    }
    if (UP) {
      if (GIN) {
        const T* gin = reinterpret_cast<const T*>(a.g_in) + rowi * R;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) t.gin[mt][g] = Raw4<T>::load(gin + 32 * mt + 8 * g + 4 * half);
      }
      const T* dfu = reinterpret_cast<const T*>(a.df_up);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int tk = tcc + (K - 1 - k) * a.dil_up;
        const int tkc = tk < a.Tlen ? tk : a.Tlen - 1;
        const T* row = dfu + (boff + tkc) * R;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) t.dfu[k][ks] = load_nat(row + 16 * ks + 8 * half);
      }
    }
    if (DOWN) {
      if (DCS) {
        const T* dr = reinterpret_cast<const T*>(a.dcs) + rowi * R;
#pragma unroll
        for (int mt = 0; mt < RT; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g) t.dcs[mt][g] = Raw4<T>::load(dr + 32 * mt + 8 * g + 4 * half);
      }
      const T* zr = reinterpret_cast<const T*>(a.z) + rowi * R;
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) t.zz[mt][g] = Raw4<T>::load(zr + 32 * mt + 8 * g + 4 * half);
    }
  };

  struct RTile {
    f32x4 gin[NI], dfu[K][NI], dcs[NI], zz[NI];
  };
  const int rsub = lane / LPR, piece = lane % LPR;
  auto load_rows = [&](int tile_, RTile& t) {
    const int tile = tile_ < a.ntiles ? tile_ : a.ntiles - 1;
    const int b = tile / a.ntb;
    const int t0 = (tile - b * a.ntb) * 32;
    const size_t boff = (size_t)b * a.Tlen;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int tc = t0 + i * RPI + rsub;
      const int tcc = tc < a.Tlen ? tc : a.Tlen - 1;
      const size_t off = (boff + tcc) * R + piece * VEC;
      if (UPM == 2) t.gin[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(a.g_out) + off);
      if (UP) {
        if (GIN) t.gin[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(a.g_in) + off);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int tk = tcc + (K - 1 - k) * a.dil_up;
          const int tkc = tk < a.Tlen ? tk : a.Tlen - 1;
          t.dfu[k][i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(a.df_up) + (boff + tkc) * R + piece * VEC);
        }
      }
      if (DOWN) {
        if (DCS) t.dcs[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(a.dcs) + off);
        t.zz[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const T*>(a.z) + off);
      }
    }
  };
  // registers (whole rows) -> the wave's LDS row tile -> registers in the layouts `process` consumes, one operand
  // at a time through the same 32 x LS buffer the stores use (LDS runs one wave's instructions in order)
  auto put_rows = [&](const f32x4 (&v)[NI]) {
    wave_lds_order();                     // earlier reads of the buffer are done
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(stage + (i * RPI + rsub) * LS + piece * VEC) = v[i];
    wave_lds_order();
  };
  auto get_acc = [&](raw4 (&o)[RT][4]) {
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) o[mt][g] = Raw4<T>::load(stage + col * LS + 32 * mt + 8 * g + 4 * half);
  };
  auto fetch_tile = [&](const RTile& r, Tile& t) {
    if (UPM == 2 || (UP && GIN)) { put_rows(r.gin); get_acc(t.gin); }
    if (UP) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        put_rows(r.dfu[k]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) t.dfu[k][ks] = load_nat(stage + col * LS + 16 * ks + 8 * half);
      }
    }
    if (DOWN) {
      if (DCS) { put_rows(r.dcs); get_acc(t.dcs); }
      put_rows(r.zz); get_acc(t.zz);
    }
  };

  auto process = [&](int tile, const Tile& t) {
    const int b = tile / a.ntb;
    const int t0 = (tile - b * a.ntb) * 32;
    const int tc = t0 + col;
    const bool ok = tc < a.Tlen;
    const int rows_valid = a.Tlen - t0;
    const size_t boff = (size_t)b * a.Tlen;
    f32x16 accG[RT];
    if (UPM == 2) {
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) accG[mt][4 * g + e] = ok ? Raw4<T>::get(t.gin[mt][g], e) : 0.0f;
    }
    if (UP) {
      // residual path: G_{l+2} * sqrt(.5) in accumulator layout
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            accG[mt][4 * g + e] = (GIN && ok) ? Raw4<T>::get(t.gin[mt][g], e) * kSqrtHalf : 0.0f;
      // conv data gradient: taps read df_up at t + (K-1-k)*d (zero beyond the clip)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool valid = ok && (tc + (K - 1 - k) * a.dil_up < a.Tlen);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const Frag<T> bf = valid ? t.dfu[k][ks] : zero_frag<T>();
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) {
            const Frag<T> af = lds_conv[(mt * (K * KS) + k * KS + ks) * 64 + lane];
            mma(accG[mt], af, bf);
          }
        }
      }
      float gv[RT][16];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) gv[mt][q] = accG[mt][q];
      store_rows_via_lds<T, RT>(stage, reinterpret_cast<T*>(a.g_out) + (boff + t0) * R, R, gv, rows_valid, lane);
    }
    if (DOWN) {
      f32x16 accC[RT];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) accC[mt][4 * g + e] = (DCS && ok) ? Raw4<T>::get(t.dcs[mt][g], e) : 0.0f;
      if (HAVEG) {
        // dres = G_{l+1} * sqrt(.5): the accumulator tile is the B operand (permuted k order)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          Frag<T> bf;
#pragma unroll
          for (int j = 0; j < 8; ++j) bf.set(j, accG[s >> 1][8 * (s & 1) + j] * kSqrtHalf);
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) {
            const Frag<T> af = lds_res[(mt * KS + s) * 64 + lane];
            mma(accC[mt], af, bf);
          }
        }
      }
      if (LEGACY) {
        const T* dt = reinterpret_cast<const T*>(a.dtotal) + (boff + (ok ? tc : 0)) * a.S + 8 * half;
        for (int ks = 0; ks < KSS; ++ks) {
          const Frag<T> bf = ok ? load_nat(dt + 16 * ks) : zero_frag<T>();
#pragma unroll
          for (int mt = 0; mt < RT; ++mt) {
            const Frag<T> af = lds_skip[(mt * KSS + ks) * 64 + lane];
            mma(accC[mt], af, bf);
          }
        }
      }
      float dv[RT][16];
#pragma unroll
      for (int mt = 0; mt < RT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            dv[mt][4 * g + e] = accC[mt][4 * g + e] * dgate_df<T>(Raw4<T>::get(t.zz[mt][g], e));
      store_rows_via_lds<T, RT>(stage, reinterpret_cast<T*>(a.df_out) + (boff + t0) * R, R, dv, rows_valid, lane);
    }
  };

  const int stride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  if constexpr (ROWS) {
    // two waves per SIMD (<= 256 registers): the other wave's loads cover this one's LDS and MFMA phases
    RTile r;
    Tile t;
    load_rows(tile, r);
    __syncthreads();
    while (tile < a.ntiles) {
      fetch_tile(r, t);
      if (tile + stride < a.ntiles) load_rows(tile + stride, r);
      process(tile, t);
      tile += stride;
    }
  } else {
    Tile ta, tb;
    load_tile(tile, ta);
    __syncthreads();
    while (tile < a.ntiles) {
      load_tile(tile + stride, tb);
      process(tile, ta);
      tile += stride;
      if (tile >= a.ntiles) break;
      load_tile(tile + stride, ta);
      process(tile, tb);
      tile += stride;
    }
  }
}

static int bwd_blocks_per_cu(int dflt) { return dflt; }

template <typename T, int RT>
static int launch_layer_bwd(LayerBwdArgs a, int B, int up, bool gin, bool down, int sk, hipStream_t st) {
  constexpr int K = 2, R = 32 * RT, KS = R / 16;
  size_t frags = 0;
  if (up == 1) frags += RT * K * KS;
  if (up && down) frags += RT * KS;
  if (down && sk == 0) frags += (size_t)RT * (a.S / 16);
  const bool rows = sizeof(T) == 2 && sk != 0;   // whole-row operand staging (kernel: ROWS)
  const size_t sh = frags * 64 * sizeof(Frag<T>) + (size_t)4 * 32 * RowStage<T>::stride(R) * sizeof(T);
  a.ntb = (a.Tlen + 31) / 32;
  const long long ntiles = (long long)B * a.ntb;
  a.ntiles = (int)ntiles;
  long long blocks = (ntiles + 3) / 4;
  const int bpc = bwd_blocks_per_cu(rows ? 2 : 1);
  if (blocks > 256LL * bpc) blocks = 256LL * bpc;
  dim3 grid((unsigned)blocks), block(256);
#define SRWN_LB(U, G, D, C)                                                                                    \
  if (up == U && gin == G && down == D && sk == C) {                                                          \
    auto kfn = layer_bwd_kernel<T, RT, K, U, G, D, C>;                                                         \
    if (sh > 32768) {                                                                                          \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
      if (e != hipSuccess) return set_error((int)e, "layer_bwd: LDS %zu: %s", sh, hipGetErrorString(e));       \
    }                                                                                                          \
    hipLaunchKernelGGL(kfn, grid, block, sh, st, a);                                                           \
    return check_launch("residual_layer_bwd");                                                                 \
  }
  SRWN_LB(1, true, true, 1)
  SRWN_LB(1, false, true, 1)
  SRWN_LB(0, false, true, 1)
  SRWN_LB(1, true, true, 0)
  SRWN_LB(1, false, true, 0)
  SRWN_LB(0, false, true, 0)
  SRWN_LB(1, true, false, 2)
  SRWN_LB(1, false, false, 2)
  SRWN_LB(1, true, true, 2)
  SRWN_LB(1, false, true, 2)
  SRWN_LB(2, false, true, 2)
#undef SRWN_LB
  return set_error(SRWN_E_UNSUPPORTED, "residual_layer_bwd: flag combination not built");
}

extern "C" int srwn_residual_layer_bwd(const void* g_in, const void* df_up, const void* wconvT_up, void* g_out,
                                       const void* wresT, const void* wskipT, const void* dtotal, const void* dcs,
                                       const void* z, void* df_out, int32_t B, int32_t T, int32_t R, int32_t S,
                                       int32_t K, int32_t dilation_up, int32_t has_up, int32_t has_down,
                                       int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!has_up && !has_down) return set_error(SRWN_E_SHAPE, "residual_layer_bwd: neither UP nor DOWN");
  if (has_up == 1 && (!df_up || !wconvT_up || !g_out)) return set_error(SRWN_E_NULL, "residual_layer_bwd: UP needs df_up, wconvT_up, g_out");
  if (has_up == 2 && (!g_out || !has_down || S != 0)) return set_error(SRWN_E_NULL, "residual_layer_bwd: has_up=2 reads G from g_out, needs DOWN and S=0");
  if (has_up < 0 || has_up > 2) return set_error(SRWN_E_SHAPE, "residual_layer_bwd: has_up=%d", has_up);
  if (has_down && (!z || !df_out)) return set_error(SRWN_E_NULL, "residual_layer_bwd: DOWN needs z, df_out");
  if (has_down && S != 0 && !dcs && (!wskipT || !dtotal)) return set_error(SRWN_E_NULL, "residual_layer_bwd: DOWN needs dcs, or wskipT + dtotal");
  if (has_down && S == 0 && !has_up) return set_error(SRWN_E_SHAPE, "residual_layer_bwd: no skip path (S=0) and no G: df would be zero");
  if (has_up && has_down && !wresT) return set_error(SRWN_E_NULL, "residual_layer_bwd: UP+DOWN needs wresT");
  if (K != 2) return set_error(SRWN_E_UNSUPPORTED, "residual_layer_bwd: filter_width %d (only 2 is built)", K);
  if (B < 0 || T < 0 || (S != 0 && S < 16) || S % 16 || (has_up == 1 && dilation_up < 1))
    return set_error(SRWN_E_SHAPE, "residual_layer_bwd: B=%d T=%d S=%d d=%d", B, T, S, dilation_up);
  if ((long long)B * ((T + 31) / 32) > 0x7fffffffLL) return set_error(SRWN_E_SHAPE, "residual_layer_bwd: too many tiles");
  LayerBwdArgs a{g_in, df_up, wconvT_up, g_out, wresT, wskipT, dtotal, dcs, z, df_out, T, dilation_up, S, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  const int up = has_up;
  const bool down = has_down != 0, gin = up == 1 && g_in != nullptr;
  // skip-path mode: 1 = dcs given, 0 = computed here from dtotal, 2 = none (S = 0, or UP-only call)
  const int use_dcs = !down ? 2 : (S == 0 ? 2 : (dcs != nullptr ? 1 : 0));
  if (dtype == SRWN_BF16) {
    if (R == 32) return launch_layer_bwd<bf16_t, 1>(a, B, up, gin, down, use_dcs, st);
    if (R == 64) return launch_layer_bwd<bf16_t, 2>(a, B, up, gin, down, use_dcs, st);
  } else if (dtype == SRWN_F32) {
    if (R == 32) return launch_layer_bwd<float, 1>(a, B, up, gin, down, use_dcs, st);
    if (R == 64) return launch_layer_bwd<float, 2>(a, B, up, gin, down, use_dcs, st);
  } else {
    return set_error(SRWN_E_DTYPE, "residual_layer_bwd: dtype %d", dtype);
  }
  return set_error(SRWN_E_UNSUPPORTED, "residual_layer_bwd: dilation_channels %d (built: 32, 64)", R);
}

// ------------------------------------------------------------------------------------------
// weight gradient: time-contraction GEMM, batched over layers
//   out[l][i][o] = scale * sum_{row} pro(In_l[row - shift_l][i] (+ cond)) * Dout_l[row][o]
//   (rows = flattened [B*T]; a shifted row must stay inside its batch element, else it contributes 0)
//   Stage 1: block (slab, nblk, l*MB+mblk) accumulates a slab of rows -> partials[l][slab][Cin][Cout] (fp32)
//            and, if requested, the column sums of Dout (bias gradient) -> bias_partials[l][slab][Cout].
//   Stage 2: srwn_reduce_partials sums slabs in a fixed order (deterministic, f64 accumulate).
//   Both operands are contracted over the slow (time) axis of channels-last memory, so the tiles
//   are staged in LDS as they lie in memory and read back with the transposing LDS read
//   (ds_read_b64_tr_b16) in bf16 mode, or element-wise in fp32 mode.
// ------------------------------------------------------------------------------------------
constexpr int kWgKC = 32;       // rows staged per step (2 k-steps of 16)
constexpr int kWgMaxBatch = 64;

struct WgradArgs {
  const void* in;  int64_t in_batch_stride;  int cin;   // In: [rows, cin] per batch entry (elements)
  const void* dout; int64_t dout_batch_stride; int cout; // Dout: [rows, cout]
  const void* cond; int64_t cond_batch_stride; int cond_frames; int pool; int cond_stride;  // optional add on In
  float* partials; float* bias_partials;
  int64_t rows; int Tlen; int rows_per_slab; int nslabs;
  int shifts[kWgMaxBatch];
};

template <typename T> struct LdsFrag;
template <> struct LdsFrag<bf16_t> {
  // natural-layout tile [row][stride] -> fragment: 8 consecutive rows (16*ks + 8h + j) at column col0 + (lane&31)
  static __device__ __forceinline__ Frag<bf16_t> load(const bf16_t* tile, int stride, int row0, int col0, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1;
    const bf16_t* base = tile + (size_t)(row0 + 8 * h + q) * stride + col0 + 16 * (g & 1) + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * stride));
    Frag<bf16_t> f;
    f.v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    return f;
  }
};
template <> struct LdsFrag<float> {
  static __device__ __forceinline__ Frag<float> load(const float* tile, int stride, int row0, int col0, int lane) {
    const int c = col0 + (lane & 31), h = lane >> 5;
    const float* base = tile + (size_t)(row0 + 8 * h) * stride + c;
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.set(j, base[(size_t)j * stride]);
    return f;
  }
};

// block tile = (WM*MTW*32) x (WN*NTW*32), WM*WN = 4 waves
template <typename T, int MTW, int NTW, int WM, int PRO, bool COND>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  constexpr int WN = 4 / WM;
  constexpr int BM = WM * MTW * 32, BN = WN * NTW * 32;
  constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte staging load
  constexpr int SIN = BM + VEC, SOUT = BN + VEC;  // padded LDS row strides (16-byte multiples)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* lin = reinterpret_cast<T*>(smem);        // [kWgKC][SIN]
  T* lout = lin + kWgKC * SIN;                // [kWgKC][SOUT]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int slab = blockIdx.x;
  const int nblk = blockIdx.y;
  const int mblocks = (a.cin + BM - 1) / BM;
  const int layer = blockIdx.z / mblocks, mblk = blockIdx.z % mblocks;
  const int shift = a.shifts[layer];
  const T* in = reinterpret_cast<const T*>(a.in) + (int64_t)layer * a.in_batch_stride;
  const T* dout = reinterpret_cast<const T*>(a.dout) + (int64_t)layer * a.dout_batch_stride;
  const T* cond = COND ? reinterpret_cast<const T*>(a.cond) + (int64_t)layer * a.cond_batch_stride : nullptr;
  const int ci0 = mblk * BM, co0 = nblk * BN;

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.0f;
  float bsum = 0.0f;  // thread j < BN sums column co0 + j of Dout (bias gradient)

  const int64_t r_begin = (int64_t)slab * a.rows_per_slab;
  const int64_t r_end = (r_begin + a.rows_per_slab < a.rows) ? r_begin + a.rows_per_slab : a.rows;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += kWgKC) {
    __syncthreads();  // previous step's fragment reads are done
    // ---- stage In tile (shifted rows, optional gate / cond add)
    for (int idx = threadIdx.x; idx < kWgKC * (BM / VEC); idx += 256) {
      const int rr = idx / (BM / VEC), cv = (idx % (BM / VEC)) * VEC;
      const int64_t row = r0 + rr;
      const int t = (int)(row % a.Tlen);
      const bool valid = (row < r_end) && (t - shift >= 0) && (t - shift < a.Tlen) && (ci0 + cv < a.cin);
      float v[VEC];
      if (valid) {
        const T* p = in + (row - shift) * a.cin + ci0 + cv;
        if (sizeof(T) == 2) {
          Frag<T> f = load_nat(p);
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] = f.get(e);
        } else {
          f32x4 f = load4(p);
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] = f[e];
        }
        if (COND) {
          const int64_t bidx = row / a.Tlen;
          const T* cp = cond + (bidx * a.cond_frames + (t - shift) / a.pool) * a.cond_stride + ci0 + cv;
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] += (float)cp[e];
        }
        if (PRO == SRWN_PRO_GATE) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] = gate_of_z<T>(v[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = 0.0f;
      }
      T* d = lin + rr * SIN + cv;
#pragma unroll
      for (int e = 0; e < VEC; ++e) d[e] = (T)v[e];
    }
    // ---- stage Dout tile
    for (int idx = threadIdx.x; idx < kWgKC * (BN / VEC); idx += 256) {
      const int rr = idx / (BN / VEC), cv = (idx % (BN / VEC)) * VEC;
      const int64_t row = r0 + rr;
      const bool valid = (row < r_end) && (co0 + cv < a.cout);
      T* d = lout + rr * SOUT + cv;
      if (valid) {
        const T* p = dout + row * a.cout + co0 + cv;
        if (sizeof(T) == 2) {
          *reinterpret_cast<bf16x8*>(d) = *reinterpret_cast<const bf16x8*>(p);
        } else {
          *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(p);
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] = (T)0.0f;
      }
    }
    __syncthreads();
    if (a.bias_partials && mblk == 0 && (int)threadIdx.x < BN) {
#pragma unroll 8
      for (int rr = 0; rr < kWgKC; ++rr) bsum += (float)lout[rr * SOUT + threadIdx.x];
    }
#pragma unroll
    for (int ks = 0; ks < kWgKC / 16; ++ks) {
      Frag<T> af[MTW], bf[NTW];
#pragma unroll
      for (int m = 0; m < MTW; ++m) af[m] = LdsFrag<T>::load(lin, SIN, 16 * ks, (wm * MTW + m) * 32, lane);
#pragma unroll
      for (int n = 0; n < NTW; ++n) bf[n] = LdsFrag<T>::load(lout, SOUT, 16 * ks, (wn * NTW + n) * 32, lane);
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) mma(acc[m][n], af[m], bf[n]);
    }
  }

  // ---- write the slab's partial tile: partials[layer][slab][cin][cout]
  float* pbase = a.partials + ((int64_t)layer * a.nslabs + slab) * (int64_t)a.cin * a.cout;
  const int col = lane & 31, half = lane >> 5;
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int o = co0 + (wn * NTW + n) * 32 + col;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = ci0 + (wm * MTW + m) * 32 + crow(q, half);
        if (i < a.cin && o < a.cout) pbase[(int64_t)i * a.cout + o] = acc[m][n][q];
      }
    }
  if (a.bias_partials && mblk == 0 && (int)threadIdx.x < BN) {
    const int o = co0 + threadIdx.x;
    if (o < a.cout) a.bias_partials[((int64_t)layer * a.nslabs + slab) * a.cout + o] = bsum;
  }
}

template <typename T, int MTW, int NTW, int WM>
static int launch_wgrad(const WgradArgs& a, int nbatch, int pro, hipStream_t st) {
  constexpr int WN = 4 / WM, BM = WM * MTW * 32, BN = WN * NTW * 32, VEC = 16 / sizeof(T);
  const size_t sh = (size_t)kWgKC * ((BM + VEC) + (BN + VEC)) * sizeof(T);
  const int mblocks = (a.cin + BM - 1) / BM, nblocks = (a.cout + BN - 1) / BN;
  dim3 grid((unsigned)a.nslabs, (unsigned)nblocks, (unsigned)(nbatch * mblocks)), block(256);
  const bool cond = a.cond != nullptr;
#define SRWN_WG(P, C)                                                              \
  if (pro == P && cond == C) {                                                     \
    hipLaunchKernelGGL((wgrad_kernel<T, MTW, NTW, WM, P, C>), grid, block, sh, st, a); \
    return check_launch("wgrad");                                                  \
  }
  SRWN_WG(SRWN_PRO_NONE, false)
  SRWN_WG(SRWN_PRO_GATE, false)
  SRWN_WG(SRWN_PRO_NONE, true)
#undef SRWN_WG
  return set_error(SRWN_E_UNSUPPORTED, "wgrad: pro %d with cond=%d not built", pro, (int)cond);
}

extern "C" int32_t srwn_wgrad_slabs(int64_t rows) {
  // ~3072 rows per slab (swept 2048..6144 on config 2: fewer, longer slabs shrink the partial volume until the
  // per-group launches stop filling the chip), at most 256 slabs
  static const int64_t per = [] { const char* e = getenv("SRWN_WG_SLAB_ROWS"); int64_t v = e ? atoll(e) : 3072; return v < 256 ? 256 : v; }();
  int64_t n = (rows + per - 1) / per;
  if (n < 1) n = 1;
  if (n > 256) n = 256;
  return (int32_t)n;
}

extern "C" int srwn_wgrad(const void* in, int64_t in_batch_stride, int32_t cin, const void* dout,
                          int64_t dout_batch_stride, int32_t cout, const void* cond, int64_t cond_batch_stride,
                          int32_t cond_frames, int32_t pool_stride, int32_t cond_row_stride, const int32_t* shifts,
                          int32_t nbatch,
                          float* partials, float* bias_partials, int64_t rows, int32_t T, int32_t nslabs,
                          int32_t pro, int32_t dtype, void* stream) {
  if (rows == 0 || nbatch == 0) return 0;
  if (!in || !dout || !partials) return set_error(SRWN_E_NULL, "wgrad: null pointer");
  if (nbatch < 0 || nbatch > kWgMaxBatch) return set_error(SRWN_E_SHAPE, "wgrad: nbatch %d (max %d)", nbatch, kWgMaxBatch);
  const int vec = (dtype == SRWN_BF16) ? 8 : 4;
  if (rows < 0 || T < 1 || rows % T || cin < vec || cin % vec || cout < vec || cout % vec || nslabs < 1)
    return set_error(SRWN_E_SHAPE, "wgrad: rows=%lld T=%d cin=%d cout=%d nslabs=%d", (long long)rows, T, cin, cout, nslabs);
  if (cond && (pool_stride < 1 || cond_row_stride < cin || (int64_t)cond_frames * pool_stride < T))
    return set_error(SRWN_E_SHAPE, "wgrad: cond frames %d x pool %d < T %d", cond_frames, pool_stride, T);
  WgradArgs a;
  a.in = in; a.in_batch_stride = in_batch_stride; a.cin = cin;
  a.dout = dout; a.dout_batch_stride = dout_batch_stride; a.cout = cout;
  a.cond = cond; a.cond_batch_stride = cond_batch_stride; a.cond_frames = cond_frames; a.pool = pool_stride > 0 ? pool_stride : 1; a.cond_stride = cond_row_stride;
  a.partials = partials; a.bias_partials = bias_partials;
  a.rows = rows; a.Tlen = T; a.nslabs = nslabs;
  int64_t rps = (rows + nslabs - 1) / nslabs;
  rps = (rps + kWgKC - 1) / kWgKC * kWgKC;
  a.rows_per_slab = (int)rps;
  for (int i = 0; i < kWgMaxBatch; ++i) a.shifts[i] = (shifts && i < nbatch) ? shifts[i] : 0;
  for (int i = 0; i < nbatch; ++i)
    if (a.shifts[i] < -(1 << 30) || a.shifts[i] > (1 << 30)) return set_error(SRWN_E_SHAPE, "wgrad: shift %d", a.shifts[i]);   // (a tap that leaves the clip contributes 0)
  hipStream_t st = (hipStream_t)stream;
  const bool big = (cin % 64 == 0) && (cout % 256 == 0);
  const bool mid = (cin % 64 == 0) && (cout % 64 == 0);
  if (dtype == SRWN_BF16) {
    if (big) return launch_wgrad<bf16_t, 2, 2, 1>(a, nbatch, pro, st);
    if (mid) return launch_wgrad<bf16_t, 1, 1, 2>(a, nbatch, pro, st);
    return launch_wgrad<bf16_t, 1, 1, 1>(a, nbatch, pro, st);
  } else if (dtype == SRWN_F32) {
    if (big) return launch_wgrad<float, 2, 2, 1>(a, nbatch, pro, st);
    if (mid) return launch_wgrad<float, 1, 1, 2>(a, nbatch, pro, st);
    return launch_wgrad<float, 1, 1, 1>(a, nbatch, pro, st);
  }
  return set_error(SRWN_E_DTYPE, "wgrad: dtype %d", dtype);
}

// out[l*out_batch_stride + i] = scale * sum_s partials[(l*part_batch_mul*nslabs + s)*n + i]
__global__ void reduce_partials_kernel(const float* __restrict__ partials, int nslabs, int64_t n, int nbatch,
                                       int part_batch_mul, float scale, float* __restrict__ out,
                                       int64_t out_batch_stride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y;
  if (i >= n || l >= nbatch) return;
  const float* p = partials + (int64_t)l * part_batch_mul * nslabs * n + i;
  double s = 0.0;
  int k = 0;
  for (; k + 8 <= nslabs; k += 8) {      // eight loads in flight; the additions keep the slab order (same bits as before)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(k + j) * n];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (double)v[j];
  }
  for (; k < nslabs; ++k) s += (double)p[(int64_t)k * n];
  out[(int64_t)l * out_batch_stride + i] = (float)(s * (double)scale);
}

// few outputs, many slabs (bias / head gradients): 16 lanes share an output, each sums every 16th slab, then a
// fixed xor-butterfly -- same result every run, 16x shorter dependent chain
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float* __restrict__ partials, int nslabs,
                                                                   int64_t n, int nbatch, int part_batch_mul,
                                                                   float scale, float* __restrict__ out,
                                                                   int64_t out_batch_stride) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = blockIdx.y, sub = threadIdx.x & 15;
  double s = 0.0;
  if (i < n) {
    const float* p = partials + (int64_t)l * part_batch_mul * nslabs * n + i;
    int k = sub;      // (eight loads in flight per lane, the same order of summation: see reduce_partials_multi_kernel)
    for (; k + 16 * 7 < nslabs; k += 16 * 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(k + 16 * j) * n];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (double)v[j];
    }
    for (; k < nslabs; k += 16) s += (double)p[(int64_t)k * n];
  }
#pragma unroll
  for (int w = 8; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
  if (i < n && sub == 0) out[(int64_t)l * out_batch_stride + i] = (float)(s * (double)scale);
}

// Several reductions in one launch (a training step finishes eleven partial buffers; launched one by one they cost
// ~6 us each plus a dependent-launch boundary).  A block finds its job by its index and does exactly what the
// single-job kernels do: the same slab order, the same bits.
// (the partial slabs are read exactly once: non-temporal loads)
#define SRWN_NT_LOAD(p) __builtin_nontemporal_load(p)
constexpr int kRpMaxJobs = 16;
struct RpJob {
  const float* partials; float* out; int64_t n; int64_t out_batch_stride;
  int nslabs, nbatch, part_batch_mul; float scale; int wide; unsigned blocks_x, block0;
  int blk_cols;      // wide == 3: `partials` holds bf16 16 x 16 blocks in lane order (SRWN_PARTIALS_BLK16)
  int sg;            // wide == 3: the slabs of an output are split over sg threads of a block (1, 2, 4 or 8)
};
struct RpMulti { RpJob j[kRpMaxJobs]; int njobs; };

__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(RpMulti m) {
  int k = 0;
  while (k + 1 < m.njobs && blockIdx.x >= m.j[k + 1].block0) ++k;
  const RpJob& job = m.j[k];
  const unsigned b = blockIdx.x - job.block0;
  const unsigned bx = b % job.blocks_x;
  const int l = (int)(b / job.blocks_x);
  const int nslabs = job.nslabs;
  const int64_t n = job.n;
  if (job.wide == 4) {
    // one output = the sum of ALL nslabs * n values (the loss partials of a training step), in srwn_reduce_loss's order
    // and precision: 256 strided f64 sums, then a tree over them in LDS -- the same bits as that launch
    __shared__ double red[256];
    const int64_t total = (int64_t)nslabs * n;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < total; i += 256) acc += (double)job.partials[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) job.out[0] = (float)(red[0] * (double)job.scale);
    return;
  }
  if (job.wide == 3) {
    // bf16 blocks in lane order: a thread takes two neighbouring lanes of one block (16 bytes per slab: four rows of two
    // columns), eight slabs in flight, f64 sums.  The slabs of an output are split over job.sg threads of the block whose
    // f64 partial sums meet in LDS: one thread per 16 bytes of output and 256 slabs each was 720 waves on the chip -- 24 KB
    // of loads in flight per CU, 4.3 TB/s, where a plain read stream reaches 6.3 (tools/micro/cuingest.hip).  Same bits
    // for any sg: a sum of a few hundred bf16 values is exact in f64, so its association does not matter.
    __shared__ double red[256 * 8];
    const int sg = job.sg, qpb = 256 / sg;                    // (block-uniform) outputs per block
    const int qi = (int)threadIdx.x % qpb, g = (int)threadIdx.x / qpb;
    const int64_t q = (int64_t)bx * qpb + qi;      // (block, lane pair)
    const bool live = q * 8 < n;
    const int per = nslabs / sg, k0 = g * per, k1 = (g == sg - 1) ? nslabs : k0 + per;
    double s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.0;
    if (live) {
      const bf16_t* p = reinterpret_cast<const bf16_t*>(job.partials) + (int64_t)l * job.part_batch_mul * nslabs * n + q * 8;
      int k = k0;
      for (; k + 8 <= k1; k += 8) {
        srwn::bf16x8 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = SRWN_NT_LOAD(reinterpret_cast<const srwn::bf16x8*>(p + (int64_t)(k + j) * n));
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += (double)(float)v[j][e];
      }
      for (; k < k1; ++k) {
        const srwn::bf16x8 v = *reinterpret_cast<const srwn::bf16x8*>(p + (int64_t)k * n);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (double)(float)v[e];
      }
    }
    if (sg > 1) {
      if (g > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(g * qpb + qi) * 8 + e] = s[e];
      }
      __syncthreads();
      if (g > 0) return;
      for (int gg = 1; gg < sg; ++gg) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += red[(gg * qpb + qi) * 8 + e];
      }
    }
    if (!live) return;
    const int64_t blk = q >> 5;
    const int pr = (int)(q & 31);
    const int nib = job.blk_cols / 16;
    const int lane = 2 * pr;
    const int64_t row0 = 16 * (blk / nib) + 4 * (lane >> 4);
    const int col = 16 * (int)(blk % nib) + (lane & 15);
    const double sc = (double)job.scale;
    float* o = job.out + (int64_t)l * job.out_batch_stride + row0 * job.blk_cols + col;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {      // lane `lane` holds s[0..3] (rows), lane + 1 holds s[4..7]: two neighbouring columns
      typedef float f2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f2*>(o + (int64_t)rr * job.blk_cols) = f2{(float)(s[rr] * sc), (float)(s[4 + rr] * sc)};
    }
  } else if (job.wide == 1) {
    const int64_t i = (int64_t)bx * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    double s = 0.0;
    if (i < n) {
      const float* p = job.partials + (int64_t)l * job.part_batch_mul * nslabs * n + i;
      // eight loads in flight per lane, summed in the order they were always summed: one load per trip made a small job
      // of many slabs (the input conv's 512 slabs of 192 floats: 32 trips, each a whole HBM round trip) the longest thing
      // in a launch that moves 190 MB
      int q = sub;
      for (; q + 16 * 7 < nslabs; q += 16 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(q + 16 * j) * n];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (double)v[j];
      }
      for (; q < nslabs; q += 16) s += (double)p[(int64_t)q * n];
    }
#pragma unroll
    for (int w = 8; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
    if (i < n && sub == 0) job.out[(int64_t)l * job.out_batch_stride + i] = (float)(s * (double)job.scale);
  } else if (job.wide == 2) {
    // many slabs of a large block (the fused backward kernels leave one slab per workgroup): four outputs per thread, 16-byte
    // loads, eight slabs in flight; every output is still summed slab by slab in order (same bits as one output per thread)
    const int64_t i = ((int64_t)bx * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float* p = job.partials + (int64_t)l * job.part_batch_mul * nslabs * n + i;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int q = 0;
    for (; q + 8 <= nslabs; q += 8) {
      srwn::f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = SRWN_NT_LOAD(reinterpret_cast<const srwn::f32x4*>(p + (int64_t)(q + j) * n));
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0 += (double)v[j][0]; s1 += (double)v[j][1]; s2 += (double)v[j][2]; s3 += (double)v[j][3]; }
    }
    for (; q < nslabs; ++q) {
      const srwn::f32x4 v = *reinterpret_cast<const srwn::f32x4*>(p + (int64_t)q * n);
      s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
    }
    const double sc = (double)job.scale;
    *reinterpret_cast<srwn::f32x4*>(job.out + (int64_t)l * job.out_batch_stride + i) =
        srwn::f32x4{(float)(s0 * sc), (float)(s1 * sc), (float)(s2 * sc), (float)(s3 * sc)};
  } else {
    const int64_t i = (int64_t)bx * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = job.partials + (int64_t)l * job.part_batch_mul * nslabs * n + i;
    double s = 0.0;
    int q = 0;
    for (; q + 8 <= nslabs; q += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(q + j) * n];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (double)v[j];
    }
    for (; q < nslabs; ++q) s += (double)p[(int64_t)q * n];
    job.out[(int64_t)l * job.out_batch_stride + i] = (float)(s * (double)job.scale);
  }
}

extern "C" int srwn_reduce_partials_multi(const SrwnReduceJob* jobs, int32_t njobs, void* stream) {
  if (njobs == 0) return 0;
  if (!jobs) return set_error(SRWN_E_NULL, "reduce_partials_multi: null pointer");
  if (njobs < 0 || njobs > kRpMaxJobs) return set_error(SRWN_E_SHAPE, "reduce_partials_multi: %d jobs (1..%d)", njobs, kRpMaxJobs);
  RpMulti m;
  m.njobs = 0;
  uint64_t blocks = 0;
  for (int k = 0; k < njobs; ++k) {
    const SrwnReduceJob& q = jobs[k];
    if (q.n == 0 || q.nbatch == 0) continue;
    if (!q.partials || !q.out) return set_error(SRWN_E_NULL, "reduce_partials_multi: job %d: null pointer", k);
    if (q.nslabs < 1 || q.n < 0 || q.nbatch < 0 || q.nbatch > 65535)
      return set_error(SRWN_E_SHAPE, "reduce_partials_multi: job %d: nslabs=%d n=%lld nbatch=%d", k, q.nslabs, (long long)q.n, q.nbatch);
    RpJob& j = m.j[m.njobs++];
    j.partials = reinterpret_cast<const float*>(q.partials); j.out = q.out; j.n = q.n; j.out_batch_stride = q.out_batch_stride;
    j.nslabs = q.nslabs; j.nbatch = q.nbatch; j.part_batch_mul = q.partials_batched ? 1 : 0; j.scale = q.scale;
    j.wide = (q.n <= 4096 && q.nslabs >= 32) ? 1 : 0;           // the choice srwn_reduce_partials makes
    // >= 128 slabs of >= 1024 outputs (16-byte aligned blocks): the four-outputs-per-thread body
    if (q.nslabs >= 128 && q.n >= 1024 && q.n % 4 == 0 && q.out_batch_stride % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(q.partials) | reinterpret_cast<uintptr_t>(q.out)) % 16 == 0)
      j.wide = 2;
    j.blk_cols = 0; j.sg = 1;
    if (q.layout == SRWN_PARTIALS_BLK16) {
      if (q.blk_cols < 16 || q.blk_cols % 16 || q.n % (16 * (int64_t)q.blk_cols) || q.out_batch_stride % 2 ||
          (reinterpret_cast<uintptr_t>(q.partials) % 16) || (reinterpret_cast<uintptr_t>(q.out) % 8))
        return set_error(SRWN_E_SHAPE, "reduce_partials_multi: job %d: BLK16 layout needs n = rows*blk_cols in whole 16 x 16 blocks (n=%lld, blk_cols=%d)", k, (long long)q.n, q.blk_cols);
      j.wide = 3; j.blk_cols = q.blk_cols;
      // split the slabs of an output over up to 8 threads until the job brings ~16 waves per CU (groups of >= 8 slabs)
      j.sg = 1;
      while (j.sg < 8 && q.nslabs / (2 * j.sg) >= 8 && (q.n / 8) * (int64_t)q.nbatch * j.sg < (int64_t)256 * 16 * 64) j.sg *= 2;
    } else if (q.layout == SRWN_PARTIALS_SUM) {
      if (q.nbatch != 1) return set_error(SRWN_E_SHAPE, "reduce_partials_multi: job %d: SUM layout has one output", k);
      j.wide = 4;
    } else if (q.layout != SRWN_PARTIALS_F32) {
      return set_error(SRWN_E_UNSUPPORTED, "reduce_partials_multi: job %d: layout %d", k, q.layout);
    }
    j.blocks_x = (unsigned)(j.wide == 4 ? 1 : j.wide == 3 ? (q.n / 8 + 256 / j.sg - 1) / (256 / j.sg) : j.wide == 2 ? (q.n / 4 + 255) / 256 : j.wide ? (q.n + 15) / 16 : (q.n + 255) / 256);
    j.block0 = (unsigned)blocks;
    blocks += (uint64_t)j.blocks_x * (uint64_t)q.nbatch;
    if (blocks > 0x7fffffffull) return set_error(SRWN_E_SHAPE, "reduce_partials_multi: grid too large");
  }
  if (m.njobs == 0) return 0;
  hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m);
  return check_launch("reduce_partials_multi");
}

extern "C" int srwn_reduce_partials(const float* partials, int32_t nslabs, int64_t n, int32_t nbatch,
                                    int32_t partials_batched, float scale, float* out, int64_t out_batch_stride,
                                    void* stream) {
  if (n == 0 || nbatch == 0) return 0;
  if (!partials || !out) return set_error(SRWN_E_NULL, "reduce_partials: null pointer");
  if (nslabs < 1 || n < 0 || nbatch < 0 || nbatch > 65535) return set_error(SRWN_E_SHAPE, "reduce_partials: nslabs=%d n=%lld nbatch=%d", nslabs, (long long)n, nbatch);
  if (n <= 4096 && nslabs >= 32) {
    dim3 grid((unsigned)((n + 15) / 16), (unsigned)nbatch), block(256);
    hipLaunchKernelGGL(reduce_partials_wide_kernel, grid, block, 0, (hipStream_t)stream, partials, nslabs, n, nbatch,
                       partials_batched ? 1 : 0, scale, out, out_batch_stride);
    return check_launch("reduce_partials");
  }
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)nbatch), block(256);
  hipLaunchKernelGGL(reduce_partials_kernel, grid, block, 0, (hipStream_t)stream, partials, nslabs, n, nbatch,
                     partials_batched ? 1 : 0, scale, out, out_batch_stride);
  return check_launch("reduce_partials");
}

// sum of pool_stride consecutive time rows per frame: adjoint of the nearest-neighbour upsample
// (ops.py:64-74) for the conditioning gradient: out[l][b,e,c] = sum_{t in frame e} g[l][b,t,c], batched over layers.
// Block (frame, b, l): C/8 lanes per row (16 B each in bf16), 256/(C/8) rows per pass, fp32 partials through LDS.
template <typename T> struct FsRow;
template <> struct FsRow<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)r[j];
  }
};
template <> struct FsRow<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
  }
};

template <typename T, int C>
__global__ __launch_bounds__(256) void frame_sum_kernel(const T* __restrict__ g, int64_t g_batch_stride,
                                                        T* __restrict__ out, int64_t out_batch_stride, int Tlen,
                                                        int frames, int pool, float scale) {
  constexpr int LPR = C / 8, RPI = 256 / LPR;
  __shared__ float red[256 * 8];
  const int e = blockIdx.x, b = blockIdx.y, l = blockIdx.z;
  const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
  const int t0 = e * pool;
  const int t1 = min(t0 + pool, Tlen);
  const T* src = g + (int64_t)l * g_batch_stride + (int64_t)b * Tlen * C + 8 * sub;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = t0 + rloc; t < t1; t += RPI) {
    float v[8];
    FsRow<T>::load(src + (int64_t)t * C, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x, sb = c >> 3, cj = c & 7;
    float s = 0.0f;
    for (int r = 0; r < RPI; ++r) s += red[(r * LPR + sb) * 8 + cj];
    out[(int64_t)l * out_batch_stride + ((int64_t)b * frames + e) * C + c] = (T)(s * scale);
  }
}

template <typename T>
__global__ void frame_sum_generic_kernel(const T* __restrict__ g, int64_t g_batch_stride, T* __restrict__ out,
                                         int64_t out_batch_stride, int B, int Tlen, int C, int frames, int pool,
                                         float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * frames * C;
  if (i >= total) return;
  const int l = blockIdx.y;
  const int c = (int)(i % C);
  const int64_t be = i / C;
  const int e = (int)(be % frames), b = (int)(be / frames);
  float s = 0.0f;
  const int t0 = e * pool;
  const T* src = g + (int64_t)l * g_batch_stride;
  for (int t = t0; t < t0 + pool && t < Tlen; ++t) s += (float)src[((int64_t)b * Tlen + t) * C + c];
  out[(int64_t)l * out_batch_stride + i] = (T)(s * scale);
}

template <typename T>
static int launch_frame_sum(const void* g, int64_t gbs, void* out, int64_t obs, int nbatch, int B, int T_, int C,
                            int frames, int pool, float scale, hipStream_t st) {
  if (C == 128 || C == 64 || C == 32) {
    dim3 grid(frames, B, nbatch), block(256);
    if (C == 128)
      hipLaunchKernelGGL((frame_sum_kernel<T, 128>), grid, block, 0, st, (const T*)g, gbs, (T*)out, obs, T_, frames, pool, scale);
    else if (C == 64)
      hipLaunchKernelGGL((frame_sum_kernel<T, 64>), grid, block, 0, st, (const T*)g, gbs, (T*)out, obs, T_, frames, pool, scale);
    else
      hipLaunchKernelGGL((frame_sum_kernel<T, 32>), grid, block, 0, st, (const T*)g, gbs, (T*)out, obs, T_, frames, pool, scale);
  } else {
    const int64_t total = (int64_t)B * frames * C;
    dim3 grid((unsigned)((total + 255) / 256), nbatch), block(256);
    hipLaunchKernelGGL(frame_sum_generic_kernel<T>, grid, block, 0, st, (const T*)g, gbs, (T*)out, obs, B, T_, C,
                       frames, pool, scale);
  }
  return check_launch("frame_sum");
}

extern "C" int srwn_frame_sum_batched(const void* g, int64_t g_batch_stride, void* out, int64_t out_batch_stride,
                                      int32_t nbatch, int32_t B, int32_t T, int32_t C, int32_t frames,
                                      int32_t pool_stride, float scale, int32_t dtype, void* stream) {
  if (B == 0 || frames == 0 || nbatch == 0) return 0;
  if (!g || !out) return set_error(SRWN_E_NULL, "frame_sum: null pointer");
  if (B < 0 || T < 1 || C < 1 || frames < 1 || pool_stride < 1 || nbatch < 0 || nbatch > 65535 || B > 65535)
    return set_error(SRWN_E_SHAPE, "frame_sum: B=%d T=%d C=%d frames=%d pool=%d nbatch=%d", B, T, C, frames,
                     pool_stride, nbatch);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SRWN_F32)
    return launch_frame_sum<float>(g, g_batch_stride, out, out_batch_stride, nbatch, B, T, C, frames, pool_stride, scale, st);
  if (dtype == SRWN_BF16)
    return launch_frame_sum<bf16_t>(g, g_batch_stride, out, out_batch_stride, nbatch, B, T, C, frames, pool_stride, scale, st);
  return set_error(SRWN_E_DTYPE, "frame_sum: dtype %d", dtype);
}

extern "C" int srwn_frame_sum(const void* g, void* out, int32_t B, int32_t T, int32_t C, int32_t frames,
                              int32_t pool_stride, int32_t dtype, void* stream) {
  return srwn_frame_sum_batched(g, 0, out, 0, 1, B, T, C, frames, pool_stride, 1.0f, dtype, stream);
}

// x[b,t,c] += bias[b, t/pool, c] in place: the first layer's conditioning bias on the input conv's output
// (model.py:181-183); every later layer receives its bias from the layer below (srwn_residual_layer_fwd).
template <typename T>
__global__ __launch_bounds__(256) void add_frame_bias_kernel(T* __restrict__ x, const T* __restrict__ bias,
                                                             int64_t bias_row_stride, int Tlen, int C, int frames,
                                                             int pool, int64_t rows) {
  const int lpr = C / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = idx / lpr;
  const int sub = (int)(idx % lpr);
  if (row >= rows) return;
  const int64_t b = row / Tlen;
  const int t = (int)(row - b * Tlen);
  float v[8], c[8];
  FsRow<T>::load(x + row * C + 8 * sub, v);
  FsRow<T>::load(bias + (b * frames + min(t / pool, frames - 1)) * bias_row_stride + 8 * sub, c);
  T* d = x + row * C + 8 * sub;
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = (T)(v[j] + c[j]);
}

extern "C" int srwn_add_frame_bias(void* x, const void* bias, int64_t bias_row_stride, int32_t B, int32_t T, int32_t C,
                                   int32_t frames, int32_t pool_stride, int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!x || !bias) return set_error(SRWN_E_NULL, "add_frame_bias: null pointer");
  if (B < 0 || T < 0 || C < 8 || C % 8 || frames < 1 || pool_stride < 1 || bias_row_stride < C)
    return set_error(SRWN_E_SHAPE, "add_frame_bias: B=%d T=%d C=%d frames=%d pool=%d", B, T, C, frames, pool_stride);
  const int64_t rows = (int64_t)B * T;
  dim3 grid((unsigned)((rows * (C / 8) + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SRWN_BF16)
    hipLaunchKernelGGL(add_frame_bias_kernel<bf16_t>, grid, block, 0, st, (bf16_t*)x, (const bf16_t*)bias, bias_row_stride, T, C, frames, pool_stride, rows);
  else if (dtype == SRWN_F32)
    hipLaunchKernelGGL(add_frame_bias_kernel<float>, grid, block, 0, st, (float*)x, (const float*)bias, bias_row_stride, T, C, frames, pool_stride, rows);
  else
    return set_error(SRWN_E_DTYPE, "add_frame_bias: dtype %d", dtype);
  return check_launch("add_frame_bias");
}
