// Parallel-WaveNet student kernels (model.py:290-537): flow head + affine transform, the data gradient
// of the input conv, the STFT power loss, global-norm clipping.  gfx950 (MI355X) only.
// All of these are HBM-streaming or tiny; the MFMA work of a flow is the shared residual-layer kernels.
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

namespace {

constexpr int kFlowRows = 256;   // rows of [B*T] per block (one entropy / weight-gradient partial each)

template <typename T> struct Row8;
template <> struct Row8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)r[j];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x8*>(p) = r;
  }
};
template <> struct Row8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = v[j]; b[j] = v[4 + j]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};

// sum over the LPR consecutive lanes that share a row (LPR a power of two <= 8)
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int w = 1; w < LPR; w <<= 1) v += __shfl_xor(v, w, 64);
  return v;
}

// ------------------------------------------------------------------------------------------
// flow head + affine (model.py:451-452, 479-483):
//   prm = relu(h) @ W2 + b2 ([rows,2]);  scale = exp(prm0), mean = prm1;  x_out = x_in*scale + mean
//   ent_partials[block] = sum over the block's rows of prm0  (= log scale; entropy term model.py:356)
// LPR = R/8 lanes share a row, 8 channels (one 16 B load in bf16) each.
// ------------------------------------------------------------------------------------------
template <typename T, int R>
__global__ __launch_bounds__(256) void flow_affine_fwd_kernel(const T* __restrict__ h, const float* __restrict__ w2,
                                                              const float* __restrict__ b2,
                                                              const float* __restrict__ x_in,
                                                              float* __restrict__ prm, float* __restrict__ x_out,
                                                              float* __restrict__ ent_partials, int64_t rows) {
  constexpr int LPR = R / 8, RPI = 256 / LPR;
  __shared__ float red[4];
  const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
  float w0[8], w1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { w0[j] = w2[(8 * sub + j) * 2]; w1[j] = w2[(8 * sub + j) * 2 + 1]; }
  const float b0 = b2[0], b1 = b2[1];
  float ent = 0.0f;
  const int64_t base = (int64_t)blockIdx.x * kFlowRows;
#pragma unroll 2
  for (int it = 0; it < kFlowRows / RPI; ++it) {
    const int64_t row = base + it * RPI + rloc;
    const bool ok = row < rows;
    float v[8];
    Row8<T>::load(h + (ok ? row : 0) * R + 8 * sub, v);
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = fmaxf(v[j], 0.0f);
      p0 = fmaf(a, w0[j], p0);
      p1 = fmaf(a, w1[j], p1);
    }
    p0 = group_sum<LPR>(p0) + b0;
    p1 = group_sum<LPR>(p1) + b1;
    if (ok && sub == 0) {
      prm[2 * row] = p0;
      prm[2 * row + 1] = p1;
      x_out[row] = fmaf(x_in[row], expf(p0), p1);
      ent += p0;
    }
  }
  // block sum of the entropy partial (fixed order)
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) ent += __shfl_xor(ent, w, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ent;
  __syncthreads();
  if (threadIdx.x == 0) ent_partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------
// backward of the same (autodiff of model.py:451-452, 479-483 and of the entropy term model.py:356,376):
//   dprm0 = dx_out * x_in * scale + ent_grad;  dprm1 = dx_out;  dx_in = dx_out * scale
//   g[row][c] = (h > 0) * (dprm0*W2[c][0] + dprm1*W2[c][1])             (gradient wrt the last dense output)
//   w_partials[block][c*2+j] = sum_rows relu(h)[c]*dprm_j;  w_partials[block][2R+j] = sum_rows dprm_j
// ------------------------------------------------------------------------------------------
template <typename T, int R>
__global__ __launch_bounds__(256) void flow_affine_bwd_kernel(const T* __restrict__ h, const float* __restrict__ w2,
                                                              const float* __restrict__ prm,
                                                              const float* __restrict__ x_in,
                                                              const float* __restrict__ dx_out, float ent_grad,
                                                              T* __restrict__ g, float* __restrict__ dx_in,
                                                              float* __restrict__ w_partials, int64_t rows) {
  constexpr int LPR = R / 8, RPI = 256 / LPR;
  __shared__ float acc_s[256 * 16];
  __shared__ float bsum[256 * 2];
  const int sub = threadIdx.x % LPR, rloc = threadIdx.x / LPR;
  float w0[8], w1[8], a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    w0[j] = w2[(8 * sub + j) * 2]; w1[j] = w2[(8 * sub + j) * 2 + 1];
    a0[j] = 0.0f; a1[j] = 0.0f;
  }
  float s0 = 0.0f, s1 = 0.0f;
  const int64_t base = (int64_t)blockIdx.x * kFlowRows;
#pragma unroll 2
  for (int it = 0; it < kFlowRows / RPI; ++it) {
    const int64_t row = base + it * RPI + rloc;
    const bool ok = row < rows;
    const int64_t rc = ok ? row : 0;
    float v[8], gv[8];
    Row8<T>::load(h + rc * R + 8 * sub, v);
    const float dxo = ok ? dx_out[rc] : 0.0f;
    const float sc = expf(prm[2 * rc]);
    const float d0 = ok ? fmaf(dxo * x_in[rc], sc, ent_grad) : 0.0f;
    const float d1 = dxo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool pos = v[j] > 0.0f;
      gv[j] = pos ? fmaf(d0, w0[j], d1 * w1[j]) : 0.0f;
      const float a = pos ? v[j] : 0.0f;
      a0[j] = fmaf(a, d0, a0[j]);
      a1[j] = fmaf(a, d1, a1[j]);
    }
    if (ok) {
      Row8<T>::store(g + row * R + 8 * sub, gv);
      if (sub == 0) { dx_in[row] = dxo * sc; s0 += d0; s1 += d1; }
    }
  }
  // block reduction over the RPI row groups: thread (rloc, sub) holds channels 8*sub..8*sub+7
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc_s[threadIdx.x * 16 + 2 * j] = a0[j]; acc_s[threadIdx.x * 16 + 2 * j + 1] = a1[j]; }
  bsum[threadIdx.x * 2] = s0; bsum[threadIdx.x * 2 + 1] = s1;
  __syncthreads();
  float* out = w_partials + (size_t)blockIdx.x * (2 * R + 2);
  for (int o = threadIdx.x; o < 2 * R + 2; o += 256) {
    float s = 0.0f;
    if (o < 2 * R) {
      const int c = o >> 1, j = o & 1, sb = c >> 3, cj = c & 7;
      for (int r = 0; r < RPI; ++r) s += acc_s[(r * LPR + sb) * 16 + 2 * cj + j];
    } else {
      const int j = o - 2 * R;
      for (int r = 0; r < RPI; ++r) s += bsum[(r * LPR) * 2 + j];
    }
    out[o] = s;
  }
}

// ------------------------------------------------------------------------------------------
// data gradient of _DilatedCausalConv1d (ops.py:6-10) for narrow inputs (the 1-channel flow input):
//   dx[b,u,i] (+)= scale * sum_k sum_o w[k,i,o] * dy[b, u + shift + (K-1-k)*d, o]     (0 beyond the clip)
// `shift` folds the adjoint of RightShift (ops.py:78-80).  One thread per (b,u); dy rows read as 8-wide vectors.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                         float* __restrict__ dx, int B, int Tlen, int Cin, int Cout,
                                                         int K, int dil, int shift, int accumulate, float scale) {
  extern __shared__ float wl[];   // [K][Cin][Cout]
  for (int i = threadIdx.x; i < K * Cin * Cout; i += 256) wl[i] = w[i];
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= (int64_t)B * Tlen) return;
  const int b = (int)(row / Tlen), u = (int)(row - (int64_t)b * Tlen);
  for (int i = 0; i < Cin; ++i) {
    float s = 0.0f;
    for (int k = 0; k < K; ++k) {
      const int t = u + shift + (K - 1 - k) * dil;
      if (t >= Tlen) continue;
      const T* r = dy + ((int64_t)b * Tlen + t) * Cout;
      const float* wk = wl + (k * Cin + i) * Cout;
      for (int o = 0; o < Cout; o += 8) {
        float v[8];
        Row8<T>::load(r + o, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(v[j], wk[o + j], s);
      }
    }
    float* d = dx + row * Cin + i;
    *d = accumulate ? fmaf(s, scale, *d) : s * scale;
  }
}

// ------------------------------------------------------------------------------------------
// STFT power (model.py:360-371): tf.contrib.signal.stft(x, 512, 256) = frames without end padding, periodic
// Hann window, 512-point real DFT (257 bins); power[b,f] = mean over frames of |X|^2.  Direct DFT with an
// LDS twiddle table -- 0.13 MFLOP per frame, nothing to gain from an FFT at this size.
//   stage 1: block (frame, b): spec[b,n,f] = (Re, Im), fpow[b,n,f] = |X|^2
//   stage 2: block b: power[b,f] = (1/nf) sum_n fpow   (fixed order)
//   backward: dx[b,t] = sum_{frames n covering t} win[j] * (2/nf) * sum_f dpow[b,f]*(Re cos - Im sin)(2 pi f j/512)
// ------------------------------------------------------------------------------------------
constexpr int kFL = 512, kFS = 256, kNB = 257;

__device__ __forceinline__ void fill_twiddles(float* cs, float* sn) {
  for (int i = threadIdx.x; i < kFL; i += 256) {
    float s, c;
    sincospif(2.0f * (float)i / (float)kFL, &s, &c);
    cs[i] = c; sn[i] = s;
  }
}

__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ x, float* __restrict__ spec,
                                                          float* __restrict__ fpow, int Tlen, int nf) {
  __shared__ float fr[kFL], cs[kFL], sn[kFL];
  const int n = blockIdx.x, b = blockIdx.y;
  fill_twiddles(cs, sn);
  __syncthreads();
  for (int j = threadIdx.x; j < kFL; j += 256)
    fr[j] = x[(int64_t)b * Tlen + (int64_t)n * kFS + j] * (0.5f - 0.5f * cs[j]);   // periodic Hann
  __syncthreads();
  for (int f = threadIdx.x; f < kNB; f += 256) {
    float re = 0.0f, im = 0.0f;
    for (int j = 0; j < kFL; ++j) {
      const int idx = (f * j) & (kFL - 1);
      re = fmaf(fr[j], cs[idx], re);
      im = fmaf(-fr[j], sn[idx], im);
    }
    const int64_t o = ((int64_t)b * nf + n) * kNB + f;
    if (spec) { spec[2 * o] = re; spec[2 * o + 1] = im; }
    fpow[o] = re * re + im * im;
  }
}

__global__ __launch_bounds__(256) void stft_mean_kernel(const float* __restrict__ fpow, float* __restrict__ power,
                                                        int nf) {
  const int b = blockIdx.x;
  for (int f = threadIdx.x; f < kNB; f += 256) {
    float s = 0.0f;
    for (int n = 0; n < nf; ++n) s += fpow[((int64_t)b * nf + n) * kNB + f];
    power[(int64_t)b * kNB + f] = s / (float)nf;
  }
}

__global__ __launch_bounds__(256) void stft_bwd_kernel(const float* __restrict__ spec, const float* __restrict__ dpow,
                                                       float* __restrict__ dx, int Tlen, int nf, int accumulate) {
  __shared__ float gr[2][kNB], gi[2][kNB], cs[kFL], sn[kFL];
  const int m = blockIdx.x, b = blockIdx.y;   // segment of 256 samples: frame m (first half), frame m-1 (second half)
  fill_twiddles(cs, sn);
  for (int q = 0; q < 2; ++q) {
    const int n = m - q;
    const bool live = n >= 0 && n < nf;
    for (int f = threadIdx.x; f < kNB; f += 256) {
      float re = 0.0f, im = 0.0f;
      if (live) {
        const int64_t o = ((int64_t)b * nf + n) * kNB + f;
        const float d = dpow[(int64_t)b * kNB + f];
        re = d * spec[2 * o]; im = d * spec[2 * o + 1];
      }
      gr[q][f] = re; gi[q][f] = im;
    }
  }
  __syncthreads();
  const int t = m * kFS + threadIdx.x;
  if (t >= Tlen) return;
  float tot = 0.0f;
  for (int q = 0; q < 2; ++q) {
    const int n = m - q;
    if (n < 0 || n >= nf) continue;
    const int j = threadIdx.x + q * kFS;
    float s = 0.0f;
    for (int f = 0; f < kNB; ++f) {
      const int idx = (f * j) & (kFL - 1);
      s = fmaf(gr[q][f], cs[idx], s);
      s = fmaf(-gi[q][f], sn[idx], s);
    }
    tot = fmaf(s, (0.5f - 0.5f * cs[j]) * (2.0f / (float)nf), tot);
  }
  float* d = dx + (int64_t)b * Tlen + t;
  *d = accumulate ? *d + tot : tot;
}

// power_loss = gamma * sum (p_truth - p_out)^2  (tf.norm(.)**2, model.py:369-371);
// dpow = d(power_loss * gscale)/d p_out = -2 gamma gscale (p_truth - p_out)
__global__ __launch_bounds__(256) void power_loss_kernel(const float* __restrict__ pt, const float* __restrict__ po,
                                                         int64_t n, float gamma, float gscale,
                                                         float* __restrict__ dpow, float* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float d = pt[i] - po[i];
    s += (double)d * (double)d;
    if (dpow) dpow[i] = -2.0f * gamma * gscale * d;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] * (double)gamma);
}

// ------------------------------------------------------------------------------------------
// tf.clip_by_global_norm(grads, clip_norm) (model.py:385): sum of squares in fixed-size chunks, then
//   norm = pre_scale * sqrt(sum);  out[0] = pre_scale * clip_norm / max(norm, clip_norm);  out[1] = norm
// (pre_scale = 1/world under data parallelism: the flat gradients hold the all-reduced SUM)
// ------------------------------------------------------------------------------------------
constexpr int kSqChunk = 4096;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ parts) {
  __shared__ float red[4];
  const int64_t base = (int64_t)blockIdx.x * kSqChunk;
  float s = 0.0f;
  for (int i = threadIdx.x; i < kSqChunk; i += 256) {
    const int64_t k = base + i;
    const float v = k < n ? g[k] : 0.0f;
    s = fmaf(v, v, s);
  }
#pragma unroll
  for (int w = 32; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void clip_scale_kernel(const float* __restrict__ parts, int64_t n, float clip_norm,
                                                         float pre_scale, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)parts[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double norm = (double)pre_scale * sqrt(red[0]);
    out[0] = (float)((double)pre_scale * (double)clip_norm / fmax(norm, (double)clip_norm));
    out[1] = (float)norm;
  }
}

// tf.minimum(tf.maximum(x, lo), hi) (model.py:535) and its gradient (ties pass the gradient to x)
__global__ __launch_bounds__(256) void clamp_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                    float lo, float hi) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = fminf(fmaxf(x[i], lo), hi);
}
__global__ __launch_bounds__(256) void clamp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ dx, int64_t n, float lo, float hi) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] = (x[i] >= lo && x[i] <= hi) ? dy[i] : 0.0f;
}

// y += alpha * scale_dev[0] * x  (averaging per-row clipped gradients, model.py:614-618: the clip factor is on device)
__global__ __launch_bounds__(256) void axpy_dev_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                       const float* __restrict__ scale_dev, float alpha, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = fmaf(alpha * scale_dev[0], x[i], y[i]);
}

}  // namespace

extern "C" int srwn_axpy_dev(float* y, const float* x, const float* scale_dev, float alpha, int64_t n, void* stream) {
  if (n == 0) return 0;
  if (!y || !x || !scale_dev) return set_error(SRWN_E_NULL, "axpy_dev: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "axpy_dev: n=%lld", (long long)n);
  hipLaunchKernelGGL(axpy_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x,
                     scale_dev, alpha, n);
  return check_launch("axpy_dev");
}

extern "C" int srwn_clamp(const float* x, float* y, int64_t n, float lo, float hi, void* stream) {
  if (n == 0) return 0;
  if (!x || !y) return set_error(SRWN_E_NULL, "clamp: null pointer");
  if (n < 0 || !(lo <= hi)) return set_error(SRWN_E_SHAPE, "clamp: n=%lld lo=%g hi=%g", (long long)n, lo, hi);
  hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, lo, hi);
  return check_launch("clamp");
}

extern "C" int srwn_clamp_bwd(const float* x, const float* dy, float* dx, int64_t n, float lo, float hi, void* stream) {
  if (n == 0) return 0;
  if (!x || !dy || !dx) return set_error(SRWN_E_NULL, "clamp_bwd: null pointer");
  if (n < 0 || !(lo <= hi)) return set_error(SRWN_E_SHAPE, "clamp_bwd: n=%lld lo=%g hi=%g", (long long)n, lo, hi);
  hipLaunchKernelGGL(clamp_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx,
                     n, lo, hi);
  return check_launch("clamp_bwd");
}

extern "C" int64_t srwn_flow_partials(int64_t rows) { return rows <= 0 ? 0 : (rows + kFlowRows - 1) / kFlowRows; }

extern "C" int srwn_flow_affine_fwd(const void* h, const float* w2, const float* b2, const float* x_in, float* prm,
                                    float* x_out, float* ent_partials, int64_t rows, int32_t R, int32_t dtype,
                                    void* stream) {
  if (rows == 0) return 0;
  if (!h || !w2 || !b2 || !x_in || !prm || !x_out || !ent_partials)
    return set_error(SRWN_E_NULL, "flow_affine_fwd: null pointer");
  if (rows < 0) return set_error(SRWN_E_SHAPE, "flow_affine_fwd: rows=%lld", (long long)rows);
  if (R != 32 && R != 64) return set_error(SRWN_E_UNSUPPORTED, "flow_affine_fwd: dilation_channels %d (built: 32, 64)", R);
  dim3 grid((unsigned)srwn_flow_partials(rows)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define SRWN_FA(TT, RR)                                                                                     \
  hipLaunchKernelGGL((flow_affine_fwd_kernel<TT, RR>), grid, block, 0, st, (const TT*)h, w2, b2, x_in, prm, \
                     x_out, ent_partials, rows)
  if (dtype == SRWN_BF16) { if (R == 32) SRWN_FA(bf16_t, 32); else SRWN_FA(bf16_t, 64); }
  else if (dtype == SRWN_F32) { if (R == 32) SRWN_FA(float, 32); else SRWN_FA(float, 64); }
  else return set_error(SRWN_E_DTYPE, "flow_affine_fwd: dtype %d", dtype);
#undef SRWN_FA
  return check_launch("flow_affine_fwd");
}

extern "C" int srwn_flow_affine_bwd(const void* h, const float* w2, const float* prm, const float* x_in,
                                    const float* dx_out, float ent_grad, void* g, float* dx_in, float* w_partials,
                                    int64_t rows, int32_t R, int32_t dtype, void* stream) {
  if (rows == 0) return 0;
  if (!h || !w2 || !prm || !x_in || !dx_out || !g || !dx_in || !w_partials)
    return set_error(SRWN_E_NULL, "flow_affine_bwd: null pointer");
  if (rows < 0) return set_error(SRWN_E_SHAPE, "flow_affine_bwd: rows=%lld", (long long)rows);
  if (R != 32 && R != 64) return set_error(SRWN_E_UNSUPPORTED, "flow_affine_bwd: dilation_channels %d (built: 32, 64)", R);
  dim3 grid((unsigned)srwn_flow_partials(rows)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define SRWN_FB(TT, RR)                                                                                        \
  hipLaunchKernelGGL((flow_affine_bwd_kernel<TT, RR>), grid, block, 0, st, (const TT*)h, w2, prm, x_in, dx_out, \
                     ent_grad, (TT*)g, dx_in, w_partials, rows)
  if (dtype == SRWN_BF16) { if (R == 32) SRWN_FB(bf16_t, 32); else SRWN_FB(bf16_t, 64); }
  else if (dtype == SRWN_F32) { if (R == 32) SRWN_FB(float, 32); else SRWN_FB(float, 64); }
  else return set_error(SRWN_E_DTYPE, "flow_affine_bwd: dtype %d", dtype);
#undef SRWN_FB
  return check_launch("flow_affine_bwd");
}

extern "C" int srwn_causal_conv1d_dgrad(const void* dy, const float* w, float* dx, int32_t B, int32_t T, int32_t Cin,
                                        int32_t Cout, int32_t K, int32_t dilation, int32_t shift, int32_t accumulate,
                                        float scale, int32_t dtype, void* stream) {
  if (B == 0 || T == 0) return 0;
  if (!dy || !w || !dx) return set_error(SRWN_E_NULL, "causal_conv1d_dgrad: null pointer");
  if (B < 0 || T < 0 || Cin < 1 || Cout < 8 || Cout % 8 || K < 1 || dilation < 1 || shift < 0)
    return set_error(SRWN_E_SHAPE, "causal_conv1d_dgrad: B=%d T=%d Cin=%d Cout=%d K=%d d=%d shift=%d", B, T, Cin, Cout,
                     K, dilation, shift);
  const size_t sh = (size_t)K * Cin * Cout * sizeof(float);
  if (sh > 48 * 1024) return set_error(SRWN_E_UNSUPPORTED, "causal_conv1d_dgrad: kernel of %zu bytes (built for narrow inputs)", sh);
  const int64_t rows = (int64_t)B * T;
  dim3 grid((unsigned)((rows + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SRWN_BF16)
    hipLaunchKernelGGL(conv_dgrad_kernel<bf16_t>, grid, block, sh, st, (const bf16_t*)dy, w, dx, B, T, Cin, Cout, K,
                       dilation, shift, accumulate, scale);
  else if (dtype == SRWN_F32)
    hipLaunchKernelGGL(conv_dgrad_kernel<float>, grid, block, sh, st, (const float*)dy, w, dx, B, T, Cin, Cout, K,
                       dilation, shift, accumulate, scale);
  else
    return set_error(SRWN_E_DTYPE, "causal_conv1d_dgrad: dtype %d", dtype);
  return check_launch("causal_conv1d_dgrad");
}

extern "C" int32_t srwn_stft_frames(int32_t T) { return T < kFL ? 0 : 1 + (T - kFL) / kFS; }

extern "C" int srwn_stft_power(const float* x, float* spec, float* frame_power, float* power, int32_t B, int32_t T,
                               void* stream) {
  if (B == 0) return 0;
  if (!x || !frame_power || !power) return set_error(SRWN_E_NULL, "stft_power: null pointer");
  const int nf = srwn_stft_frames(T);
  if (B < 0 || nf < 1) return set_error(SRWN_E_SHAPE, "stft_power: B=%d T=%d (a clip must hold one 512-sample frame)", B, T);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stft_frames_kernel, dim3(nf, B), dim3(256), 0, st, x, spec, frame_power, T, nf);
  hipLaunchKernelGGL(stft_mean_kernel, dim3(B), dim3(256), 0, st, frame_power, power, nf);
  return check_launch("stft_power");
}

extern "C" int srwn_stft_power_bwd(const float* spec, const float* dpower, float* dx, int32_t B, int32_t T,
                                   int32_t accumulate, void* stream) {
  if (B == 0) return 0;
  if (!spec || !dpower || !dx) return set_error(SRWN_E_NULL, "stft_power_bwd: null pointer");
  const int nf = srwn_stft_frames(T);
  if (B < 0 || nf < 1) return set_error(SRWN_E_SHAPE, "stft_power_bwd: B=%d T=%d", B, T);
  hipLaunchKernelGGL(stft_bwd_kernel, dim3((T + kFS - 1) / kFS, B), dim3(256), 0, (hipStream_t)stream, spec, dpower,
                     dx, T, nf, accumulate);
  return check_launch("stft_power_bwd");
}

extern "C" int srwn_power_loss(const float* power_truth, const float* power_out, int64_t n, float gamma,
                               float grad_scale, float* dpower, float* loss, void* stream) {
  if (!power_truth || !power_out || !loss) return set_error(SRWN_E_NULL, "power_loss: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "power_loss: n=%lld", (long long)n);
  hipLaunchKernelGGL(power_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, power_truth, power_out, n, gamma,
                     grad_scale, dpower, loss);
  return check_launch("power_loss");
}

extern "C" int64_t srwn_sumsq_partials(int64_t n) { return n <= 0 ? 0 : (n + kSqChunk - 1) / kSqChunk; }

extern "C" int srwn_sumsq(const float* g, int64_t n, float* partials, void* stream) {
  if (n == 0) return 0;
  if (!g || !partials) return set_error(SRWN_E_NULL, "sumsq: null pointer");
  if (n < 0) return set_error(SRWN_E_SHAPE, "sumsq: n=%lld", (long long)n);
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)srwn_sumsq_partials(n)), dim3(256), 0, (hipStream_t)stream, g, n,
                     partials);
  return check_launch("sumsq");
}

extern "C" int srwn_clip_scale(const float* partials, int64_t n, float clip_norm, float pre_scale, float* out,
                               void* stream) {
  if (!partials || !out) return set_error(SRWN_E_NULL, "clip_scale: null pointer");
  if (n < 0 || !(clip_norm > 0.0f)) return set_error(SRWN_E_SHAPE, "clip_scale: n=%lld clip_norm=%g", (long long)n, clip_norm);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n, clip_norm, pre_scale,
                     out);
  return check_launch("clip_scale");
}
