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
};

template <typename T> __device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <typename T, int MT, int KSC, int PRO, int EPI>
__global__ __launch_bounds__(256, (EPI == SRWN_EPI_SOFTMAX_CE) ? 1 : 2) void rowgemm_kernel(RgArgs a) {
  constexpr int FB = sizeof(Frag<T>) * 64;          // bytes of one fragment image (1 KiB bf16, 2 KiB f32)
  constexpr int CHUNK_B = MT * KSC * FB;            // bytes of one weight chunk in LDS
  constexpr int PIECES = CHUNK_B / 1024;            // 1-KiB glds pieces per chunk
  static_assert(PIECES % 4 == 0, "chunk must split evenly over 4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][CHUNK_B]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t row = tile * 32 + col;
  const bool valid = row < a.rows;
  const int nchunks = a.ks_total / KSC;
  const char* wbase = reinterpret_cast<const char*>(a.wpack);

  // piece p of chunk c: fragment run (mt, byte offset) -> global address; LDS image is [mt][KSC][64] frags
  auto stage = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < PIECES / 4; ++i) {
      const int p = wave * (PIECES / 4) + i;
      const int mt = p / (KSC * FB / 1024), within = (p % (KSC * FB / 1024)) * 1024;
      const char* g = wbase + ((size_t)mt * a.ks_total + (size_t)c * KSC) * FB + within + lane * 16;
      char* l = smem + buf * CHUNK_B + mt * (KSC * FB) + within;
      glds16<T>(g, l);
    }
  };
  auto load_b = [&](int c, Frag<T>* dst) {
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      const int kg = 16 * (c * KSC + ks);
      const int chunk = kg / a.chunk_len, within = kg - chunk * a.chunk_len;
      const T* p = reinterpret_cast<const T*>(a.x) + (int64_t)chunk * a.x_chunk_stride +
                   (valid ? row : 0) * a.x_row_stride + within + 8 * half;
      dst[ks] = valid ? load_nat(p) : zero_frag<T>();
    }
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int n = 32 * mt + crow(q, half);
      acc[mt][q] = (a.bias && n < a.cout_valid) ? a.bias[n] : 0.0f;
    }

  Frag<T> bcur[KSC], bnext[KSC];
  stage(0, 0);
  load_b(0, bnext);
  __syncthreads();   // (drains the glds: vmcnt(0) + barrier)
  for (int c = 0; c < nchunks; ++c) {
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) {
      bcur[ks] = bnext[ks];
      if (PRO == SRWN_PRO_GATE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bcur[ks].set(j, gate_of_z<T>(bcur[ks].get(j)));
      }
    }
    if (c + 1 < nchunks) {
      stage(c + 1, (c + 1) & 1);
      load_b(c + 1, bnext);
    }
    const Frag<T>* lw = reinterpret_cast<const Frag<T>*>(smem + (c & 1) * CHUNK_B) + lane;
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const Frag<T> af = lw[(mt * KSC + ks) * 64];
        mma(acc[mt], af, bcur[ks]);
      }
    __syncthreads();
  }

  // ------------------------------------------------------------------------------------ epilogues
  if (EPI == SRWN_EPI_SOFTMAX_CE) {
    const int tgt = valid ? a.targets[row] : -1;
    float m = -INFINITY, vt = 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = 32 * mt + crow(q, half);
        if (n < a.cout_valid) m = fmaxf(m, acc[mt][q]);
        if (n == tgt) vt = acc[mt][q];
      }
    m = fmaxf(m, __shfl_xor(m, 32));
    vt += __shfl_xor(vt, 32);
    float s = 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = 32 * mt + crow(q, half);
        if (n < a.cout_valid) s += __expf(acc[mt][q] - m);
      }
    s += __shfl_xor(s, 32);
    const float lse = m + __logf(s);
    float loss = (valid && half == 0) ? (lse - vt) : 0.0f;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) loss += __shfl_xor(loss, off);
    if (lane == 0 && tile * 32 < a.rows) a.loss_partials[tile] = loss;
    if (!valid) return;
    if (a.logits_out) {
      float* lr = a.logits_out + row * (int64_t)a.cout_valid;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int n = 32 * mt + crow(q, half);
          if (n < a.cout_valid) lr[n] = acc[mt][q];
        }
    }
    if (a.y) {
      T* dr = reinterpret_cast<T*>(a.y) + row * a.y_row_stride;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = 32 * mt + 8 * g + 4 * half + e;
            const float p = (n < a.cout_valid) ? __expf(acc[mt][4 * g + e] - lse) : 0.0f;
            v[e] = (p - (n == tgt ? 1.0f : 0.0f)) * a.grad_scale;
          }
          store4(dr + 32 * mt + 8 * g + 4 * half, v[0], v[1], v[2], v[3]);
        }
    }
    return;
  }

  if (!valid) return;
  T* yrow = reinterpret_cast<T*>(a.y) + row * a.y_row_stride;
  const T* arow = (EPI == SRWN_EPI_MASK) ? reinterpret_cast<const T*>(a.aux) + row * a.aux_row_stride : nullptr;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = 32 * mt + 8 * g + 4 * half;
      if (n0 >= a.cout_valid) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * g + e];
      if (EPI == SRWN_EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
      } else if (EPI == SRWN_EPI_MASK) {
        const f32x4 mk = load4(arow + n0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (mk[e] > 0.0f) ? v[e] : 0.0f;
      }
      store4(yrow + n0, v[0], v[1], v[2], v[3]);
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
};

template <typename T, int RT, int KSS, int NBUF>
__global__ __launch_bounds__(256) void colgemm_kernel(CgArgs a) {
  constexpr int R = 32 * RT;
  constexpr int FB = sizeof(Frag<T>) * 64;
  constexpr int CHUNK_B = RT * KSS * FB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][CHUNK_B] weights | 4 row stages
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  T* stage = reinterpret_cast<T*>(smem + NBUF * CHUNK_B) + wave * (32 * RowStage<T>::stride(R));
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const int64_t row = row0 + col;
  const bool valid = row < a.rows;
  const int rows_valid = (a.rows - row0) < 32 ? (int)(a.rows - row0) : 32;   // may be <= 0 for idle waves
  const char* wbase = reinterpret_cast<const char*>(a.wpack);

  lds_dma_copy(wbase, smem, CHUNK_B, wave, lane, 4);
  Frag<T> bf[KSS];
  {
    const T* p = reinterpret_cast<const T*>(a.x) + (valid ? row : 0) * a.x_row_stride + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks) bf[ks] = load_nat(p + 16 * ks);
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks) bf[ks] = valid ? bf[ks] : zero_frag<T>();
  }
  __syncthreads();
  for (int l = 0; l < a.nlayers; ++l) {
    const int buf = (NBUF == 2) ? (l & 1) : 0;
    if (NBUF == 2 && l + 1 < a.nlayers)
      lds_dma_copy(wbase + (size_t)(l + 1) * CHUNK_B, smem + ((l + 1) & 1) * CHUNK_B, CHUNK_B, wave, lane, 4);
    const Frag<T>* lw = reinterpret_cast<const Frag<T>*>(smem + buf * CHUNK_B) + lane;
    f32x16 acc[RT];
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][q] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KSS; ++ks)
#pragma unroll
      for (int mt = 0; mt < RT; ++mt) {
        const Frag<T> af = lw[(mt * KSS + ks) * 64];
        mma(acc[mt], af, bf[ks]);
      }
    float v[RT][16];
#pragma unroll
    for (int mt = 0; mt < RT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) v[mt][q] = acc[mt][q];
    T* ytile = reinterpret_cast<T*>(a.y) + (int64_t)l * a.y_layer_stride + (rows_valid > 0 ? row0 : 0) * R;
    store_rows_via_lds<T, RT>(stage, ytile, R, v, rows_valid, lane);
    __syncthreads();
    if (NBUF == 1 && l + 1 < a.nlayers) {
      lds_dma_copy(wbase + (size_t)(l + 1) * CHUNK_B, smem, CHUNK_B, wave, lane, 4);
      __syncthreads();
    }
  }
}

template <typename T, int RT, int KSS, int NBUF>
static int launch_cg(const CgArgs& a, hipStream_t st) {
  constexpr int R = 32 * RT;
  const size_t sh = (size_t)NBUF * RT * KSS * sizeof(Frag<T>) * 64 + (size_t)4 * 32 * RowStage<T>::stride(R) * sizeof(T);
  auto kfn = colgemm_kernel<T, RT, KSS, NBUF>;
  hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  if (e != hipSuccess) return set_error((int)e, "skip_dgrad_all: LDS %zu: %s", sh, hipGetErrorString(e));
  hipLaunchKernelGGL(kfn, dim3((unsigned)((a.rows + 127) / 128)), dim3(256), sh, st, a);
  return check_launch("skip_dgrad_all");
}

extern "C" int srwn_skip_dgrad_all(const void* dtotal, const void* wskipT_all, void* dcs, int64_t dcs_layer_stride,
                                   int32_t nlayers, int64_t rows, int32_t R, int32_t S, int32_t dtype, void* stream) {
  if (rows == 0 || nlayers == 0) return 0;
  if (!dtotal || !wskipT_all || !dcs) return set_error(SRWN_E_NULL, "skip_dgrad_all: null pointer");
  if (rows < 0 || nlayers < 0) return set_error(SRWN_E_SHAPE, "skip_dgrad_all: rows=%lld layers=%d", (long long)rows, nlayers);
  CgArgs a{dtotal, S, wskipT_all, dcs, dcs_layer_stride, nlayers, rows};
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

template <typename T, int MT, int KSC>
static int launch_rg(const RgArgs& a, int pro, int epi, hipStream_t st) {
  constexpr int CHUNK_B = MT * KSC * (int)sizeof(Frag<T>) * 64;
  const size_t sh = 2 * (size_t)CHUNK_B;
  dim3 grid((unsigned)((a.rows + 127) / 128)), block(256);
#define SRWN_RG(P, E)                                                                                        \
  if (pro == P && epi == E) {                                                                                \
    auto kfn = rowgemm_kernel<T, MT, KSC, P, E>;                                                             \
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

// Returns 1 if the shape is served by the row-streaming kernel (then *rc holds the launch result).
int rowgemm_dispatch(const void* x, int64_t x_row_stride, int64_t x_chunk_stride, int chunk_len, int Cin,
                     const void* wpack, const float* bias, void* y, int64_t y_row_stride, int cout_pad,
                     int cout_valid, int64_t rows, const void* aux, int64_t aux_row_stride, const int32_t* targets,
                     float* loss_partials, float* logits_out, float grad_scale, int pro, int epi, int dtype,
                     hipStream_t st, int* rc) {
  if (cout_pad != 256 || (Cin % 64) != 0 || rows < 1) return 0;
  RgArgs a{x, x_row_stride, x_chunk_stride, chunk_len, Cin / 16, wpack, bias, y, y_row_stride, cout_valid, rows,
           aux, aux_row_stride, targets, loss_partials, logits_out, grad_scale};
  if (dtype == SRWN_BF16) *rc = launch_rg<bf16_t, 8, 4>(a, pro, epi, st);
  else if (dtype == SRWN_F32) *rc = launch_rg<float, 8, 2>(a, pro, epi, st);
  else return 0;
  return 1;
}

}  // namespace srwn
