// Weight gradient of all skip 1x1s (tf.gradients of ops.py:44 summed over model.py:50) from the forward group kernel's
// TRANSPOSED gate outputs:  dWs[l][n][s] = sum_t c_l[t][n] * dskip[t][s]   -- a contraction over time.
//
// srwn_wgrad256 reads z in its natural [time][channel] layout, recomputes the gate (an exp and a reciprocal per element,
// 245 M elements), writes both operands to LDS through registers (ds_write_b128: 79 B/clk/CU) and reads both back with
// transposing reads: 194 us for 126 GFLOP.  Here the A operand is what srwn_residual_group_fwd_wt already left in HBM
// for the backward group kernel's dWr: c in "weight-gradient tiles" (64 channels x 32 positions as four fragments in lane order, a lane's 16 bytes = eight
// time steps of one channel = one v_mfma_f32_16x16x32_bf16 A fragment), loaded straight into registers; only dskip goes
// through LDS, by LDS-DMA (no VGPR round trip, no ds_write), dense 512-byte rows with the 32-byte blocks of a row XOR-ed
// by (row >> 1) & 7 -- the transposing reads of a 16-column block touch rows 2 apart, so eight rows' blocks land on
// eight different bank groups: conflict-free without row padding, and the tile is exactly 16 wave-instructions of DMA.
//
// A workgroup = 8 waves = up to four layers with the same segment geometry (dilations with the same stride / segment
// length: their tiles cover the same positions, so one dskip tile serves all four) x all 256 skip channels; wave =
// (layer, half of the skip channels): 4 x 8 tiles of 16x16, 32 MFMAs per 32 positions against 4 fragment loads and 16
// transposing reads.  A tile position p of segment (clip b, residue r, first position j0) is time r + st (j0 + p): the
// dskip rows of a tile are st rows apart, fetched row by row; positions beyond the segment read a row of zeros (cT holds
// finite values there).  Loads run three tiles (dskip) / two tiles (c) ahead of the products with hand-counted waits.
// fp32 partials per slab of segments in srwn_wgrad256's layout ([slab][layer*64 + n][256]), summed in fixed order by
// srwn_reduce_partials.  bf16, 64 residual / 256 skip channels.
#include "srwn_common.h"
#include "srwn_group.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;
using namespace srwn::grp;

namespace {

constexpr int kMaxBlk = 16;
constexpr int kNB = 4;                   // dskip tiles in LDS
constexpr int kTileB = 32 * 512;         // bytes of one dskip tile: 32 positions x 256 channels

struct WtBlk { int nl; int layer[4]; int st, W, nsub, KT; };
struct WtSkipArgs {
  const void* cT; long long wt_stride;
  const void* d; long long d_row_stride;
  void* partials; float* bias_partials;      // partials: fp32 [slab][layer*64 + n][256], or (part16) the same matrix as bf16
  int part16;                                //           16 x 16 blocks in lane order (SRWN_PARTIALS_BLK16, 256 columns)
  int B, Tlen, nslabs, mtotal;
  int safe_wait;                         // SRWN_SAFE_WAIT: vmcnt(0) instead of the counted waits
  int bias_blk;                          // the block whose idle waves sum dskip's columns (-1: none does)
  WtBlk blk[kMaxBlk];
};

__device__ uint4 g_zero_row[32];         // 512 bytes of zeros: the dskip row of a position nobody owns

typedef bf16_t T;

__device__ __forceinline__ f32x4 gload16_untracked(const void* p) {
  f32x4 v;
  // (nt: the c tiles are read once here; round 4 A/B -7 us.  The same hint on the backward group kernel's and the skip
  // sum's streams cost +44 and +20 us: their neighbouring workgroups re-read halo rows / rows of the same lines)
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// where one chunk (= one weight-gradient tile) of a slab lives
struct Cursor {
  int idx, seg, k, wrem;                 // wrem = owned positions from the tile's first one on (<= 0: none)
  long long drow0;                       // dskip row of the segment's position 0
};

__global__ __launch_bounds__(512) void wgrad_skip_wt_kernel(WtSkipArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = blockIdx.x;
  const WtBlk& bk = a.blk[blockIdx.y];
  const int st = bk.st, W = bk.W, nsub = bk.nsub, KT = bk.KT;
  const int slot = wave >> 1, nh = wave & 1;
  const bool live = slot < bk.nl;
  const int layer = bk.layer[live ? slot : 0];
  const int per_clip = st * nsub;
  const int nseg = a.B * per_clip;
  const int sps = (nseg + a.nslabs - 1) / a.nslabs;
  const int s0 = slab * sps;
  const int s1 = (s0 + sps < nseg) ? s0 + sps : nseg;
  const int nchunk = (s1 > s0 ? s1 - s0 : 0) * KT;
  const int nit = (nchunk + 2) / 3 * 3;

  auto seg_setup = [&](Cursor& c) {
    const int b = c.seg / per_clip;
    const int rem = c.seg - b * per_clip;
    int r, j0;
    if (nsub == 1) { r = rem; j0 = 0; }
    else { r = rem / nsub; j0 = (rem - r * nsub) * W; }
    const int Jr = (a.Tlen - r + st - 1) / st;
    const int wseg = (Jr - j0) < W ? (Jr - j0) : W;
    c.wrem = wseg;
    c.drow0 = (long long)b * a.Tlen + r + (long long)st * j0;
  };
  auto start = [&](Cursor& c) {
    c.idx = 0; c.seg = nchunk > 0 ? s0 : 0; c.k = 0; c.wrem = 0; c.drow0 = 0;
    if (nchunk > 0) seg_setup(c);
  };
  auto advance = [&](Cursor& c) {
    ++c.idx;
    if (c.idx >= nchunk) { c.wrem = 0; return; }           // beyond the slab: nothing owned, pointers stay where they are
    ++c.k; c.wrem -= 32;
    if (c.k == KT) { c.k = 0; ++c.seg; seg_setup(c); }
  };

  // ---- dskip tile -> LDS: wave w issues DMA pieces w and w + 8 (rows 2w, 2w+1 and 2w+16, 2w+17); lane: row = 2 piece +
  // (lane >> 5), destination block (lane & 31) >> 1 holds source block ((lane & 31) >> 1) ^ (piece & 7) = ... ^ w
  const int drow_a = 2 * wave + (lane >> 5);
  const int dcol = 32 * ((((lane & 31) >> 1) ^ wave)) + 16 * (lane & 1);            // source byte inside the row
  const char* dbase = reinterpret_cast<const char*>(a.d) + dcol;
  const char* zrow = reinterpret_cast<const char*>(g_zero_row) + dcol;
  const long long drow_bytes = a.d_row_stride * (long long)sizeof(T);
  const unsigned lds0 = (unsigned)(size_t)smem;
  auto dma_tile = [&](const Cursor& c, int buf) {
    const long long row0 = c.drow0 + (long long)st * (32 * c.k);
    const char* ga = (drow_a < c.wrem) ? dbase + (row0 + (long long)st * drow_a) * drow_bytes : zrow;
    const char* gb = (drow_a + 16 < c.wrem) ? dbase + (row0 + (long long)st * (drow_a + 16)) * drow_bytes : zrow;
    const unsigned l = lds0 + buf * kTileB + wave * 1024;
    glds16_untracked(ga, __builtin_amdgcn_readfirstlane(l));
    glds16_untracked(gb, __builtin_amdgcn_readfirstlane(l + 8 * 1024));
  };
  // ---- c tile of this wave's layer -> registers (four fragments = 64 channels x 32 positions)
  const T* ct_layer = reinterpret_cast<const T*>(a.cT) + (size_t)layer * a.wt_stride + lane * 8;   // (wt_load's layout)
  auto load_a = [&](const Cursor& c, f32x4 (&f)[4]) {
    const T* t = ct_layer + ((size_t)c.seg * KT + c.k) * (64 * 32);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) f[mb] = gload16_untracked(t + mb * 512);
  };

  // ---- transposing reads: lane (kg, q, p) reads rows 16 (kg >> 1) + 2 (kg & 1) + 4 q (+1), 8 bytes at 8 p of block nb
  const int kg = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int hl = ((kg & 1) + 2 * q) & 7;
  const int rowoff = (16 * (kg >> 1) + 2 * (kg & 1) + 4 * q) * 512 + 8 * p;
  int boff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) boff[i] = rowoff + 32 * (8 * nh + (i ^ hl));

  f32x4 acc[4][8];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[mb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.bias_partials != nullptr && (int)blockIdx.y == a.bias_blk && slot == 3;
  float bsum[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bsum[i] = 0.0f;
  Frag<T> ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones.set(j, 1.0f);

  Cursor ca, cd;
  start(ca); start(cd);
  f32x4 A0[4], A1[4], A2[4];
  // prologue: as if the three iterations before the first had run
  dma_tile(cd, 0); advance(cd);
  load_a(ca, A0); advance(ca);
  dma_tile(cd, 1); advance(cd);
  load_a(ca, A1); advance(ca);
  dma_tile(cd, 2); advance(cd);

  typedef short s16x4 __attribute__((ext_vector_type(4)));
  auto bfrag = [&](int buf, int i) {
    const char* bp = smem + buf * kTileB + boff[i];
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bp));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bp + 512));
    Frag<T> bf;
    bf.v = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    return bf;
  };
  // B fragments run two groups of four MFMAs (128 cycles) ahead of their use -- one group ahead left every group waiting
  // on its transposing reads -- and the first two of the NEXT chunk (published by this iteration's barrier) are read
  // behind the last products, so a chunk starts without a bubble
  Frag<T> bq0 = zero_frag<T>(), bq1 = zero_frag<T>();
  auto compute = [&](int buf, int nbuf, const f32x4 (&A)[4]) {
    Frag<T> af[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) af[mb].v = __builtin_bit_cast(bf16x8, A[mb]);
    Frag<T> b[10];
    b[0] = bq0; b[1] = bq1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      b[i + 2] = (i + 2 < 8) ? bfrag(buf, i + 2) : bfrag(nbuf, i + 2 - 8);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) mma16(acc[mb][i], af[mb], b[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
    bq0 = b[8]; bq1 = b[9];
  };
  // the column sums of dskip (the skip biases' gradient, the same for every layer): by the two waves of a block with
  // fewer than four layers that have no products to do (a test per fragment inside compute() splits the MFMA sequence
  // into blocks the register allocator answers with 80 to 460 spills; a block of four has no idle wave: colsum kernel)
  auto bias_only = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = frag_dot(bsum[i], bfrag(buf, i), ones);
  };
  // one iteration: everything but the youngest 6 loads has landed (this chunk's c fragments, the next chunk's dskip tile);
  // the barrier publishes that tile and frees the one read an iteration ago
#define SRWN_WT_ITER(IT, ACUR, ANEW)                                                                            \
  {                                                                                                             \
    /* ONE statement defines the landed fragments on every path: two alternative ones (a full drain under SRWN_SAFE_WAIT) \
       made the compiler place the copies that reconcile them BEFORE the wait of one branch -- reads of registers whose  \
       load was still in flight (found by build.py's asmcheck).  The full drain follows the counted wait instead. */    \
    asm volatile("s_waitcnt vmcnt(6)" : "+v"(ACUR[0]), "+v"(ACUR[1]), "+v"(ACUR[2]), "+v"(ACUR[3])::"memory");       \
    if (a.safe_wait) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ACUR[0]), "+v"(ACUR[1]), "+v"(ACUR[2]), "+v"(ACUR[3])::"memory"); \
    wg_barrier();                                                                                               \
    load_a(ca, ANEW); advance(ca);                                                                              \
    dma_tile(cd, ((IT) + 3) & (kNB - 1)); advance(cd);                                                          \
    if (live) compute((IT) & (kNB - 1), ((IT) + 1) & (kNB - 1), ACUR);                                          \
    else if (do_bias) bias_only((IT) & (kNB - 1));                                                              \
  }
  if (nit > 0) {   // the first chunk's first two B fragments: tile 0 (and 1) landed, the barrier publishes them
    if (a.safe_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    wg_barrier();
    if (live) { bq0 = bfrag(0, 0); bq1 = bfrag(0, 1); }
  }
  for (int it = 0; it < nit; it += 3) {
    SRWN_WT_ITER(it, A0, A2)
    SRWN_WT_ITER(it + 1, A1, A0)
    SRWN_WT_ITER(it + 2, A2, A1)
  }
#undef SRWN_WT_ITER
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(A0[0]), "+v"(A1[0]), "+v"(A2[0])::"memory");   // nothing in flight at exit

  if (live && a.part16) {      // block (row block layer*4 + mb, column block 8 nh + i): one 8-byte store per lane
    bf16_t* pb = reinterpret_cast<bf16_t*>(a.partials) + (size_t)slab * a.mtotal * 256;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const size_t blk = (size_t)(layer * 4 + mb) * 16 + 8 * nh + i;
        __builtin_nontemporal_store(bf16x4{(bf16_t)acc[mb][i][0], (bf16_t)acc[mb][i][1], (bf16_t)acc[mb][i][2], (bf16_t)acc[mb][i][3]},
                                    reinterpret_cast<bf16x4*>(pb + (blk * 64 + lane) * 4));
      }
  } else if (live) {
    float* pb = reinterpret_cast<float*>(a.partials) + ((size_t)slab * a.mtotal + (size_t)layer * 64) * 256 + 128 * nh + (lane & 15);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[(size_t)(16 * mb + 4 * (lane >> 4) + r) * 256 + 16 * i] = acc[mb][i][r];
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 16) a.bias_partials[(size_t)slab * 256 + 128 * nh + 16 * i + lane] = v;
    }
  }
}

// column sums of d [rows, 256] per slab of rows (only when every block has four layers: no idle wave to do it)
__global__ __launch_bounds__(256) void colsum256_kernel(const T* d, long long row_stride, long long rows, int nslabs, float* out) {
  const long long rps = (rows + nslabs - 1) / nslabs;
  const long long r0 = (long long)blockIdx.x * rps;
  const long long r1 = (r0 + rps < rows) ? r0 + rps : rows;
  float s0 = 0.0f, s1 = 0.0f;
  const int c = 2 * (threadIdx.x & 127), half = threadIdx.x >> 7;
  for (long long r = r0 + half; r < r1; r += 2) {
    const unsigned v = *reinterpret_cast<const unsigned*>(d + r * row_stride + c);
    s0 += __builtin_bit_cast(float, v << 16);
    s1 += __builtin_bit_cast(float, v & 0xffff0000u);
  }
  __shared__ float red[2][256];
  red[half][c] = s0; red[half][c + 1] = s1;
  __syncthreads();
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x];
}

// layers with the same (stride, segment length) in blocks of at most four
int plan_blocks(const int32_t* st, const int32_t* seg_rows, int nlayers, int T, WtBlk* out) {
  int nb = 0;
  bool used[512] = {false};
  if (nlayers > 512) return -1;
  for (int l = 0; l < nlayers; ++l) {
    if (used[l]) continue;
    if (nb == kMaxBlk) return -1;
    WtBlk& b = out[nb++];
    b.nl = 0; b.st = st[l]; b.W = seg_rows[l];
    const int J = (T + b.st - 1) / b.st;
    b.nsub = (J + b.W - 1) / b.W;
    b.KT = (b.W + 31) / 32;
    for (int m = l; m < nlayers && b.nl < 4; ++m)
      if (!used[m] && st[m] == b.st && seg_rows[m] == b.W) { b.layer[b.nl++] = m; used[m] = true; }
    for (int i = b.nl; i < 4; ++i) b.layer[i] = b.layer[0];
  }
  return nb;
}

}  // namespace

extern "C" int32_t srwn_wgrad_skip_wt_slabs(const int32_t* st, const int32_t* seg_rows, int32_t nlayers, int32_t T) {
  if (!st || !seg_rows || nlayers < 1 || T < 1) return 0;
  for (int l = 0; l < nlayers; ++l) if (st[l] < 1 || seg_rows[l] < 1) return 0;
  WtBlk blk[kMaxBlk];
  const int nb = plan_blocks(st, seg_rows, nlayers, T, blk);
  if (nb < 1) return 0;
  const int s = num_cus() / nb;
  return s < 1 ? 1 : s;
}

extern "C" int srwn_wgrad_skip_wt(const void* cT, int64_t wt_layer_stride, const int32_t* st, const int32_t* seg_rows,
                                  int32_t nlayers, const void* d, int64_t d_row_stride, void* partials,
                                  float* bias_partials, int32_t part16, int32_t nslabs, int32_t B, int32_t T, int32_t R,
                                  int32_t S, int32_t dtype, void* stream) {
  if (B == 0 || T == 0 || nlayers == 0) return 0;
  if (!cT || !st || !seg_rows || !d || !partials) return set_error(SRWN_E_NULL, "wgrad_skip_wt: null pointer");
  if (dtype != SRWN_BF16 || R != 64 || S != 256)
    return set_error(SRWN_E_UNSUPPORTED, "wgrad_skip_wt: built for bf16, 64 residual / 256 skip channels (dtype %d R=%d S=%d)", dtype, R, S);
  if (B < 0 || T < 0 || nlayers < 0 || nslabs < 1 || d_row_stride < S || (d_row_stride % 8) || wt_layer_stride < 0)
    return set_error(SRWN_E_SHAPE, "wgrad_skip_wt: B=%d T=%d layers=%d nslabs=%d d_row_stride=%lld", B, T, nlayers, nslabs,
                     (long long)d_row_stride);
  for (int l = 0; l < nlayers; ++l)
    if (st[l] < 1 || seg_rows[l] < 1) return set_error(SRWN_E_SHAPE, "wgrad_skip_wt: layer %d stride %d segment %d", l, st[l], seg_rows[l]);
  WtSkipArgs a;
  a.cT = cT; a.wt_stride = wt_layer_stride; a.d = d; a.d_row_stride = d_row_stride; a.partials = partials;
  a.part16 = part16 ? 1 : 0;
  a.bias_partials = bias_partials; a.B = B; a.Tlen = T; a.nslabs = nslabs; a.mtotal = nlayers * 64;
  const int nb = plan_blocks(st, seg_rows, nlayers, T, a.blk);
  if (nb < 1) return set_error(SRWN_E_UNSUPPORTED, "wgrad_skip_wt: more than %d layer blocks", kMaxBlk);
  for (int i = 0; i < nb; ++i) {
    const long long nseg = (long long)B * a.blk[i].st * a.blk[i].nsub;
    if (nseg > 0x7fffffffLL || nseg * a.blk[i].KT * 64 * 32 > wt_layer_stride)
      return set_error(SRWN_E_SHAPE, "wgrad_skip_wt: %lld segments x %d tiles exceed wt_layer_stride %lld", nseg, a.blk[i].KT,
                       (long long)wt_layer_stride);
  }
  a.bias_blk = -1;
  a.safe_wait = safe_wait();
  if (bias_partials)
    for (int i = 0; i < nb && a.bias_blk < 0; ++i)
      if (a.blk[i].nl < 4) a.bias_blk = i;
  if (bias_partials && a.bias_blk < 0)
    hipLaunchKernelGGL(colsum256_kernel, dim3((unsigned)nslabs), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const bf16_t*>(d), (long long)d_row_stride, (long long)B * T, nslabs, bias_partials);
  const size_t sh = (size_t)kNB * kTileB;
  hipError_t e = hipFuncSetAttribute((const void*)wgrad_skip_wt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  if (e != hipSuccess) return set_error((int)e, "wgrad_skip_wt: LDS %zu: %s", sh, hipGetErrorString(e));
  hipLaunchKernelGGL(wgrad_skip_wt_kernel, dim3((unsigned)nslabs, (unsigned)nb), dim3(512), sh, (hipStream_t)stream, a);
  return check_launch("wgrad_skip_wt");
}
