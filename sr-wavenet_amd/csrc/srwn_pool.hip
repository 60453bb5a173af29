// Time-pooled softmax head of the WaveNet classifier (model.py:56-60, loss model.py:24-29).
// The average pool over the whole clip commutes with the last 1x1, so the pooled path is
//   mean_t r1[b,t,:]  ->  [B,S] @ W2 + b2  ->  softmax / CE with soft labels
// and its backward broadcasts one [B,S] gradient row over time through the relu mask.
// These are small VALU kernels (B rows); the heavy stack below them is shared with the teacher path.
#include "srwn_common.h"
#include "srwn_host.h"
#include "../../include/srwn.h"

using namespace srwn;

constexpr int kTmRows = 256;

// partials[b][slab][c] = sum over the slab's time rows of x[b,t,c]
template <typename T>
__global__ __launch_bounds__(256) void time_sum_kernel(const T* __restrict__ x, float* __restrict__ partials,
                                                       int Tlen, int C, int nslabs) {
  const int b = blockIdx.y, slab = blockIdx.x;
  const int t0 = slab * kTmRows, t1 = min(t0 + kTmRows, Tlen);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.0f;
    for (int t = t0; t < t1; ++t) s += (float)x[((int64_t)b * Tlen + t) * C + c];
    partials[((int64_t)b * nslabs + slab) * C + c] = s;
  }
}

extern "C" int32_t srwn_time_mean_slabs(int32_t T) { return (T + kTmRows - 1) / kTmRows; }

extern "C" int srwn_time_mean(const void* x, float* partials, float* out, int32_t B, int32_t T, int32_t C,
                              int32_t dtype, void* stream) {
  if (B == 0) return 0;
  if (!x || !partials || !out) return set_error(SRWN_E_NULL, "time_mean: null pointer");
  if (B < 0 || T < 1 || C < 1) return set_error(SRWN_E_SHAPE, "time_mean: B=%d T=%d C=%d", B, T, C);
  const int ns = srwn_time_mean_slabs(T);
  dim3 grid((unsigned)ns, (unsigned)B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SRWN_F32)
    hipLaunchKernelGGL(time_sum_kernel<float>, grid, block, 0, st, (const float*)x, partials, T, C, ns);
  else if (dtype == SRWN_BF16)
    hipLaunchKernelGGL(time_sum_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, partials, T, C, ns);
  else
    return set_error(SRWN_E_DTYPE, "time_mean: dtype %d", dtype);
  int rc = check_launch("time_mean");
  if (rc) return rc;
  // out[b][c] = (1/T) * sum_slab partials[b][slab][c]
  return srwn_reduce_partials(partials, ns, C, B, 1, 1.0f / (float)T, out, C, stream);
}

// One block.  mean [B,S] f32, w2 [S,ldw] f32 (columns >= C ignored), b2, labels [B,C] f32.
//   logits = mean @ w2 + b2; probs = softmax(logits); loss = mean_b( -sum_c labels*log_softmax )
//   dl = (probs*sum_c(labels) - labels) / B
//   gw2[s][c] = sum_b mean[b][s]*dl[b][c];  gb2[c] = sum_b dl[b][c];  dmean[b][s] = sum_c dl[b][c]*w2[s][c]
__global__ __launch_bounds__(256) void pooled_head_kernel(const float* __restrict__ mean, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, const float* __restrict__ labels,
                                                          float* __restrict__ probs, float* __restrict__ loss,
                                                          float* __restrict__ gw2, float* __restrict__ gb2,
                                                          float* __restrict__ dmean, int B, int S, int C, int ldw) {
  extern __shared__ float sh[];  // logits/dl [B*C], red[256]
  float* dl = sh;
  float* red = sh + B * C;
  for (int i = threadIdx.x; i < B * C; i += 256) {
    const int b = i / C, c = i % C;
    float acc = b2[c];
    for (int s = 0; s < S; ++s) acc = fmaf(mean[b * S + s], w2[(int64_t)s * ldw + c], acc);
    dl[i] = acc;
  }
  __syncthreads();
  float lsum = 0.0f;
  for (int b = 0; b < B; ++b) {  // rows are few; every thread walks every row, columns split over threads
    float m = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, dl[b * C + c]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + w]);
      __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float se = 0.0f, sl = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) {
      se += expf(dl[b * C + c] - m);
      sl += labels ? labels[b * C + c] : 0.0f;
    }
    red[threadIdx.x] = se;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    se = red[0];
    __syncthreads();
    red[threadIdx.x] = sl;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    sl = red[0];
    __syncthreads();
    const float lse = m + logf(se);
    float part = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float lg = dl[b * C + c];
      const float p = expf(lg - lse);
      const float y = labels ? labels[b * C + c] : 0.0f;
      if (probs) probs[b * C + c] = p;
      part += -y * (lg - lse);
      dl[b * C + c] = (p * sl - y) / (float)B;
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    lsum += red[0];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss) loss[0] = lsum / (float)B;
  if (!labels) return;
  for (int i = threadIdx.x; i < S * ldw; i += 256) {
    const int s = i / ldw, c = i % ldw;
    float acc = 0.0f;
    if (c < C)
      for (int b = 0; b < B; ++b) acc = fmaf(mean[b * S + s], dl[b * C + c], acc);
    gw2[i] = acc;
  }
  for (int c = threadIdx.x; c < ldw; c += 256) {
    float acc = 0.0f;
    if (c < C)
      for (int b = 0; b < B; ++b) acc += dl[b * C + c];
    gb2[c] = acc;
  }
  for (int i = threadIdx.x; i < B * S; i += 256) {
    const int b = i / S, s = i % S;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) acc = fmaf(dl[b * C + c], w2[(int64_t)s * ldw + c], acc);
    dmean[i] = acc;
  }
}

extern "C" int srwn_pooled_head(const float* mean, const float* w2, const float* b2, const float* labels,
                                float* probs, float* loss, float* gw2, float* gb2, float* dmean, int32_t B,
                                int32_t S, int32_t C, int32_t ldw, void* stream) {
  if (B == 0) return 0;
  if (!mean || !w2 || !b2) return set_error(SRWN_E_NULL, "pooled_head: null pointer");
  if (labels && (!gw2 || !gb2 || !dmean)) return set_error(SRWN_E_NULL, "pooled_head: labels given but gradient outputs missing");
  if (B < 0 || S < 1 || C < 1 || ldw < C) return set_error(SRWN_E_SHAPE, "pooled_head: B=%d S=%d C=%d ldw=%d", B, S, C, ldw);
  const size_t sh = ((size_t)B * C + 256) * sizeof(float);
  if (sh > 65536) return set_error(SRWN_E_SHAPE, "pooled_head: B*C=%d too large for one block", B * C);
  hipLaunchKernelGGL(pooled_head_kernel, dim3(1), dim3(256), sh, (hipStream_t)stream, mean, w2, b2, labels, probs,
                     loss, gw2, gb2, dmean, B, S, C, ldw);
  return check_launch("pooled_head");
}

// da1[b,t,s] = (r1[b,t,s] > 0) ? dmean[b,s] * scale : 0     (gradient of mean_t followed by relu mask)
template <typename T>
__global__ void bcast_mask_kernel(const float* __restrict__ dmean, const T* __restrict__ r1, T* __restrict__ out,
                                  int64_t rows, int Tlen, int S, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread = 4 channels
  const int sq = S / 4;
  if (i >= rows * sq) return;
  const int s0 = (int)(i % sq) * 4;
  const int64_t row = i / sq;
  const int b = (int)(row / Tlen);
  const f32x4 m = load4(r1 + row * S + s0);
  const f32x4 d = *reinterpret_cast<const f32x4*>(dmean + (int64_t)b * S + s0);
  store4(out + row * S + s0, m[0] > 0.f ? d[0] * scale : 0.f, m[1] > 0.f ? d[1] * scale : 0.f,
         m[2] > 0.f ? d[2] * scale : 0.f, m[3] > 0.f ? d[3] * scale : 0.f);
}

extern "C" int srwn_bcast_mask(const float* dmean, const void* r1, void* out, int32_t B, int32_t T, int32_t S,
                               float scale, int32_t dtype, void* stream) {
  if (B == 0) return 0;
  if (!dmean || !r1 || !out) return set_error(SRWN_E_NULL, "bcast_mask: null pointer");
  if (B < 0 || T < 1 || S < 4 || S % 4) return set_error(SRWN_E_SHAPE, "bcast_mask: B=%d T=%d S=%d", B, T, S);
  const int64_t rows = (int64_t)B * T, total = rows * (S / 4);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == SRWN_F32)
    hipLaunchKernelGGL(bcast_mask_kernel<float>, grid, block, 0, (hipStream_t)stream, dmean, (const float*)r1, (float*)out, rows, T, S, scale);
  else if (dtype == SRWN_BF16)
    hipLaunchKernelGGL(bcast_mask_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, dmean, (const bf16_t*)r1, (bf16_t*)out, rows, T, S, scale);
  else
    return set_error(SRWN_E_DTYPE, "bcast_mask: dtype %d", dtype);
  return check_launch("bcast_mask");
}

// ------------------------------------------------------------------------------------------
// discretised mixture-of-logistics NLL + gradient (ops.py:124-175), one thread per time step.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float mol_softplus(float v) { return (v > 20.0f) ? v : log1pf(expf(v)); }
__device__ __forceinline__ float mol_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

template <typename T>
__global__ __launch_bounds__(256) void mol_loss_kernel(const float* __restrict__ logits, int64_t ldl,
                                                       const float* __restrict__ x, int M,
                                                       float* __restrict__ loss_partials, T* __restrict__ dlogits,
                                                       int64_t ldd, int64_t rows, float grad_scale,
                                                       float* __restrict__ dx) {
  __shared__ float red[256];
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.0f;
  if (row < rows) {
    const float* l = logits + row * ldl;
    const float xv = x[row];
    float lp[16], dmean[16], dls[16], lg[16];
    float mx = -INFINITY;
    for (int m = 0; m < M; ++m) { lg[m] = l[m]; mx = fmaxf(mx, lg[m]); }
    float se = 0.0f;
    for (int m = 0; m < M; ++m) se += expf(lg[m] - mx);
    const float lse_logit = mx + logf(se);
    float mlp = -INFINITY;
    for (int m = 0; m < M; ++m) {
      const float mean = l[M + m];
      const float raw = l[2 * M + m];
      const float ls = fmaxf(raw, -7.0f);                                  // ops.py:137
      const float inv = expf(-ls);
      const float cen = xv - mean;
      const float plus_in = inv * (cen + (1.0f / 255.0f));
      const float min_in = inv * (cen - (1.0f / 255.0f));
      const float mid_in = inv * cen;
      const float sp = mol_sigmoid(plus_in), sm = mol_sigmoid(min_in), sd = mol_sigmoid(mid_in);
      const float cdf_delta = sp - sm;
      float comp, dm, ds;
      if (xv < -0.999f) {                                                   // ops.py:169, branch by branch
        comp = plus_in - mol_softplus(plus_in);
        dm = (1.0f - sp) * (-inv); ds = (1.0f - sp) * (-plus_in);
      } else if (xv > 0.999f) {
        comp = -mol_softplus(min_in);
        dm = sm * inv; ds = sm * min_in;
      } else if (cdf_delta > 1e-5f) {
        const float den = fmaxf(cdf_delta, 1e-12f);
        comp = logf(den);
        const float a = sp * (1.0f - sp), b = sm * (1.0f - sm);
        dm = (-(a - b) * inv) / den;
        ds = (-(a * plus_in - b * min_in)) / den;
      } else {
        comp = mid_in - ls - 2.0f * mol_softplus(mid_in) - 4.848116f;       // log(127.5)
        dm = (1.0f - 2.0f * sd) * (-inv); ds = (1.0f - 2.0f * sd) * (-mid_in) - 1.0f;
      }
      lp[m] = comp + (lg[m] - lse_logit);                                   // ops.py:171
      dmean[m] = dm;
      dls[m] = (raw > -7.0f) ? ds : 0.0f;
      mlp = fmaxf(mlp, lp[m]);
    }
    float sw = 0.0f;
    for (int m = 0; m < M; ++m) sw += expf(lp[m] - mlp);
    const float lse = mlp + logf(sw);
    loss = -lse;                                                            // ops.py:174
    if (dlogits) {
      T* dr = dlogits + row * ldd;
      for (int m = 0; m < M; ++m) {
        const float w = expf(lp[m] - lse);
        const float smx = expf(lg[m] - lse_logit);
        dr[m] = (T)(-(w - smx) * grad_scale);
        dr[M + m] = (T)(-w * dmean[m] * grad_scale);
        dr[2 * M + m] = (T)(-w * dls[m] * grad_scale);
        dr[3 * M + m] = (T)0.0f;                                            // coeffs never reach the loss
      }
      for (int64_t c = 4 * M; c < ldd; ++c) dr[c] = (T)0.0f;
    }
    if (dx) {   // x enters through centered = x - mean only (ops.py:147): d/dx = -sum_m d/dmean_m
      float g = 0.0f;
      for (int m = 0; m < M; ++m) g += expf(lp[m] - lse) * dmean[m];
      dx[row] = g * grad_scale;
    }
  }
  red[threadIdx.x] = loss;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = red[0];
}

extern "C" int srwn_mol_loss(const float* logits, int64_t ldl, const float* x, int32_t M, float* loss_partials,
                             void* dlogits, int64_t ldd, int64_t rows, float grad_scale, int32_t dtype,
                             void* stream) {
  if (rows == 0) return 0;
  if (!logits || !x || !loss_partials || !dlogits) return set_error(SRWN_E_NULL, "mol_loss: null pointer");
  if (rows < 0 || M < 1 || M > 16 || ldl < 4 * M || ldd < 4 * M)
    return set_error(SRWN_E_SHAPE, "mol_loss: rows=%lld M=%d ldl=%lld ldd=%lld", (long long)rows, M, (long long)ldl, (long long)ldd);
  dim3 grid((unsigned)((rows + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SRWN_F32)
    hipLaunchKernelGGL(mol_loss_kernel<float>, grid, block, 0, st, logits, ldl, x, M, loss_partials, (float*)dlogits, ldd, rows, grad_scale, (float*)nullptr);
  else if (dtype == SRWN_BF16)
    hipLaunchKernelGGL(mol_loss_kernel<bf16_t>, grid, block, 0, st, logits, ldl, x, M, loss_partials, (bf16_t*)dlogits, ldd, rows, grad_scale, (float*)nullptr);
  else
    return set_error(SRWN_E_DTYPE, "mol_loss: dtype %d", dtype);
  return check_launch("mol_loss");
}

extern "C" int srwn_mol_loss_dx(const float* logits, int64_t ldl, const float* x, int32_t M, float* loss_partials,
                                float* dx, int64_t rows, float grad_scale, void* stream) {
  if (rows == 0) return 0;
  if (!logits || !x || !loss_partials || !dx) return set_error(SRWN_E_NULL, "mol_loss_dx: null pointer");
  if (rows < 0 || M < 1 || M > 16 || ldl < 4 * M)
    return set_error(SRWN_E_SHAPE, "mol_loss_dx: rows=%lld M=%d ldl=%lld", (long long)rows, M, (long long)ldl);
  dim3 grid((unsigned)((rows + 255) / 256)), block(256);
  hipLaunchKernelGGL(mol_loss_kernel<float>, grid, block, 0, (hipStream_t)stream, logits, ldl, x, M, loss_partials,
                     (float*)nullptr, (int64_t)0, rows, grad_scale, dx);
  return check_launch("mol_loss_dx");
}
