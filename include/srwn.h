/* srwn.h -- C-ABI of libsrwn.so: MI355X (gfx950) kernels for the WaveNet residual-stack hot path.
 *
 * The reference (tachitachi/SR-WaveNet, TensorFlow 1.x Python) has NO native/FFI interface; its
 * seam is the Python call surface of ops.py / model.py.  Each entry point below names the
 * reference function (file:line under /root/reference) whose arithmetic it replaces; the host
 * mirror of that Python surface lives in sr-wavenet_amd/{ops,model}.py and binds this header
 * through the pybind11 module build.py generates from it (_srwn_pyb; ctypes on request:
 * SRWN_BINDING=ctypes) -- see INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; nothing is allocated or freed here;
 *  - `stream` is a hipStream_t passed as void*; launches are asynchronous on it; no call
 *    synchronises, so every entry point may be captured into a hipGraph;
 *  - return value: 0 = ok, >0 = hipError_t of the launch, <0 = argument error (SRWN_E_*);
 *    srwn_last_error() returns a thread-local message for the last non-zero return;
 *  - tensors are channels-last [B,T,C] / [rows,C] (ops.py:4); master weights are fp32 in the
 *    reference's own shapes ([K,Cin,Cout] conv kernels, ops.py:5; [1,Cin,Cout] tf.layers.conv1d);
 *  - `dtype` selects the activation/compute type: SRWN_F32 (exact fp32 MFMA, parity mode) or
 *    SRWN_BF16 (bf16 MFMA, fp32 accumulate, throughput mode);
 *  - "packed" weights are MFMA A-operand fragment images built by srwn_pack_a_index +
 *    srwn_pack_gather from the fp32 master weights (layout: csrc/srwn_common.h).
 */
#ifndef SRWN_H
#define SRWN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRWN_F32 0
#define SRWN_BF16 1

#define SRWN_E_DTYPE (-1)
#define SRWN_E_SHAPE (-2)
#define SRWN_E_NULL (-3)
#define SRWN_E_UNSUPPORTED (-4)

/* pw_linear prologue / epilogue selectors */
#define SRWN_PRO_NONE 0
#define SRWN_PRO_GATE 1 /* x -> x*sigmoid(x): rebuilds c = z*sigmoid(z) from stored z (ops.py:33,36) */
#define SRWN_EPI_NONE 0
#define SRWN_EPI_RELU 1
#define SRWN_EPI_MASK 2 /* y *= (aux > 0): relu backward against the saved activation */
#define SRWN_EPI_F32 4  /* y is fp32 regardless of dtype (mixture-of-logistics parameters need full precision) */

int srwn_version(void);
const char* srwn_last_error(void);

/* ---- mu-law companding: ops.py:82-93 (encode) and ops.py:96-104 (decode); bit-exact vs the
 *      oracle (log1p/pow evaluated in f64 and rounded once, every other step an IEEE f32 op). */
int srwn_mu_law_encode(const float* audio, int32_t* codes, int64_t n, int32_t quantization_channels, void* stream);
int srwn_mu_law_decode(const int32_t* codes, float* audio, int64_t n, int32_t quantization_channels, void* stream);

/* ---- weight packing (no reference counterpart: TF keeps [K,Cin,Cout] and lets cuDNN/Eigen relayout)
 * srwn_pack_a_index writes, for an A operand with `mt_count` 32-row tiles and k-steps
 * [ks_offset, ks_offset+ks_count) of a packed image whose k extent is `ks_total` steps, the int32
 * source index   src_offset + row*row_stride + k*k_stride   of every fragment element
 * (row = output channel, k = contraction index local to this call), or -1 where row >= rows_valid
 * or k >= k_valid (zero padding).  k-steps >= perm_from_ks (local) use the permuted k order.
 * idx image: [mt_count][ks_total][64 lanes][8] int32, written at dst_idx.
 * srwn_pack_gather then materialises dst[i] = (dtype) src[idx[i]] (0 where idx<0) in one launch. */
int srwn_pack_a_index(int32_t* dst_idx, int32_t src_offset, int32_t rows_valid, int32_t k_valid,
                      int32_t row_stride, int32_t k_stride, int32_t mt_count, int32_t ks_total,
                      int32_t ks_offset, int32_t ks_count, int32_t perm_from_ks, void* stream);
int srwn_pack_gather(const float* src, const int32_t* idx, void* dst, int64_t n, int32_t dtype, void* stream);
/* srwn_pack_gather and, in the same launch, the column sums of a [sum_rows, sum_cols] fp32 matrix (fp64 accumulate, rows
 * in order): sum_out[c] = sum_l sum_src[l*sum_cols + c].  The training step re-packs the weight images after every
 * optimizer step; the sum of the layers' skip biases -- the bias of the skip sum, model.py:50 -- rides along instead of
 * being a reduction launch in front of every forward pass.  sum_rows = 0: plain srwn_pack_gather. */
int srwn_pack_gather_rowsum(const float* src, const int32_t* idx, void* dst, int64_t n, int32_t dtype,
                            const float* sum_src, int32_t sum_rows, int32_t sum_cols, float* sum_out, void* stream);

/* ---- generic dilated causal conv: _DilatedCausalConv1d / DilatedCausalConv1d (ops.py:6-20)
 * y[b,t,o] = bias[o] + sum_k sum_i x[b, t-(K-1-k)*dilation - shift, i] * w[k,i,o]  (zero for t<0).
 * `shift` = 1 folds RightShift (ops.py:78-80) into the tap offsets.  x is fp32 (audio side);
 * y is `dtype_out`.  Plain-VALU kernel: used for the Cin=1 input conv (model.py:40,173) and the
 * ops-level API; the 64-channel hot conv lives inside srwn_residual_layer_fwd. */
int srwn_causal_conv1d_fwd(const float* x, const float* w, const float* bias, void* y, int32_t B, int32_t T,
                           int32_t Cin, int32_t Cout, int32_t K, int32_t dilation, int32_t shift,
                           int32_t dtype_out, void* stream);
/* gradient of the Cin=1 input conv wrt its kernel [K,1,R] and bias [R] (autodiff of model.py:40):
 * gw[k,o] = sum_{b,t} audio[b,t-(K-1-k)-shift] * g[b,t,o]; gb[o] = sum g.  `partials` is a
 * workspace of srwn_init_conv_wgrad_partials(B,T,R,K) floats; result written (not accumulated).
 * gw = gb = NULL: only the per-slab partials are written -- partials[slab][(K+1)*R] = [gw | gb] of each slab of rows,
 * srwn_init_conv_wgrad_partials / ((K+1)*R) slabs -- for the caller to sum (the training step does it as one more job
 * of its srwn_reduce_partials_multi launch instead of a launch of its own). */
int64_t srwn_init_conv_wgrad_partials(int32_t B, int32_t T, int32_t R, int32_t K);
int srwn_init_conv_wgrad(const float* audio, const void* g, float* partials, float* gw, float* gb, int32_t B,
                         int32_t T, int32_t R, int32_t K, int32_t shift, int32_t dtype, void* stream);

/* ---- fused residual layer forward: ResidualDilationLayer (ops.py:23-46) for K=2 taps, with the
 * decoder's conditioning add (model.py:180-183; NN upsample ops.py:64-74 as t/pool_stride) fused in.
 * x is the layer's COMPLETE input (the conditioning bias of this layer already added); the kernel adds the NEXT
 * layer's bias to what it stores, so no consumer (taps, residual base, weight gradients, the generator's rings)
 * ever re-adds it:
 *   z   = tanh(conv_K(x) + bias_f)                          -> z_out  (saved for skip GEMM and backward)
 *   c   = z * sigmoid(z)                                     (ops.py:33: the gate conv result is discarded)
 *   h   = (x + c @ Wr + bias_r) * sqrt(.5) + cond_next[b, t/pool_stride, :]   -> h_out   (cond_next may be NULL)
 * The first layer's bias is added to the input conv's output by srwn_add_frame_bias.
 * The skip 1x1 (ops.py:44) is deferred to srwn_pw_linear over the stored z of all layers.
 * wconv: packed [R/32][K*R/16] (last tap permuted k order), wres: packed [R/32][R/16] (permuted).
 * cond rows are cond_row_stride elements apart (one [B*frames, L*R] product serves every layer). */
int srwn_residual_layer_fwd(const void* x, const void* cond, const void* wconv, const void* wres,
                            const float* bias_f, const float* bias_r, void* h_out, void* z_out, int32_t B,
                            int32_t T, int32_t R, int32_t K, int32_t dilation, int32_t cond_frames,
                            int32_t pool_stride, int32_t cond_row_stride, int32_t dtype, void* stream);

/* ---- several consecutive residual layers per launch (the stacking loops model.py:42-47, 176-189, 428-453 around
 * ResidualDilationLayer, ops.py:23-46).  Same arithmetic, operands and outputs as `nlayers` calls of
 * srwn_residual_layer_fwd (bit-identical results), but the layer outputs travel between layers in LDS:
 * layer g (dilation dilations[g]) reads x_{g} and stores z_g at z_out + g*layer_stride and x_{g+1} at
 * x_out + g*layer_stride (elements; the engine's [L,B,T,R] stacks).  wconv/wres/bias_f/bias_r/cond_next are HOST
 * arrays of nlayers device pointers (cond_next[g] = the conditioning bias of the layer above layer g, or NULL).
 * Requirement: sum(dilations)/gcd(dilations) <= 31 and nlayers <= 8 (srwn_group_plan cuts a stack accordingly):
 * the kernel works on the residue classes t = j*gcd + r, where the group's dilations are small, and recomputes a
 * halo of that many steps per segment.  seg_rows = 0 lets the library choose the segment length. */
int srwn_residual_group_fwd(const void* x0, void* x_out, void* z_out, int64_t layer_stride,
                            const void* const* wconv, const void* const* wres, const float* const* bias_f,
                            const float* const* bias_r, const void* const* cond_next, int32_t cond_frames,
                            int32_t pool_stride, int32_t cond_row_stride, const int32_t* dilations, int32_t nlayers,
                            int32_t B, int32_t T, int32_t R, int32_t K, int32_t seg_rows, int32_t dtype, void* stream);
/* the backward chain of such a group (autodiff of the same lines; TF builds it in AdamOptimizer.minimize, model.py:31),
 * top layer first, in one launch: for g = nlayers-1 .. 0
 *   df_g = (Wr_g . (G_{g+1} sqrt(.5)) + dcs_g) * d(z sigmoid z)/df (z_g)        -> df_out + g*layer_stride
 *   G_g  = G_{g+1} sqrt(.5) + sum_k Wf_g[k] . df_g[t + (K-1-k)*dilations[g]]     -> g_out  + g*layer_stride
 * with G_{nlayers} = g_top (NULL = 0: the teacher's last dense output is unused, model.py:45-50) and dcs = Ws . dtotal
 * of every layer from srwn_skip_dgrad_all (NULL for the flows of ParallelWaveNet, model.py:440-449: no skip path).
 * Same values as nlayers + 1 calls of srwn_residual_layer_bwd (identical in fp32; in bf16 the gradient handed from
 * layer to layer is the stored, rounded one).  wconvT / wresT: HOST arrays of nlayers device pointers. */
int srwn_residual_group_bwd(const void* g_top, void* g_out, void* df_out, const void* z, const void* dcs,
                            int64_t layer_stride, const void* const* wconvT, const void* const* wresT,
                            const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R, int32_t K,
                            int32_t seg_rows, int32_t dtype, void* stream);
/* ---- layer weight gradients summed inside the backward group kernel, 8 waves, output-split ("wt" mode).
 * The two kernels of a group are given the SAME segment cut (srwn_group_wt_geometry).  srwn_residual_group_fwd_wt is
 * srwn_residual_group_fwd that also writes, per layer g of the group and per 32-step tile of every segment, the
 * TRANSPOSED layer input x_g and gate output c_g = z_g sigmoid(z_g) ("weight-gradient tiles", per 16 channels x 32 steps one MFMA fragment in lane order,
 * layer g at xT / cT + g*wt_layer_stride elements; steps a segment does not own are zero in xT).  With
 * store_inner_x = 0 only the group's top layer stores its output rows (x_out + (nlayers-1)*layer_stride): in this mode
 * nothing reads the inner layers' (their transposed copies feed the weight gradients).
 * srwn_residual_group_bwd_wt is the chain of srwn_residual_group_bwd which, per layer, additionally contracts over time
 *   part_r [g][slab][i][o]     = sum c_g[t,i] * G_{g+1}[t,o]      part_br[g][slab][o] = sum G_{g+1}[t,o]
 *   part_f [g][slab][k*R+i][o] = sum x_g[t-(1-k)*d_g,i] * df_g[t,o]   part_bf[g][slab][o] = sum df_g[t,o]
 * (the sums of srwn_wgrad_layers, in its layout; tf.gradients of ops.py:27,39) with the A operands loaded as MFMA
 * fragments from those tiles and df_g / G_{g+1} read from the kernel's own LDS image, the R x R outputs split into
 * 16 x 16 blocks over the eight waves.  df is not stored; G of the inner layers only with write_all_g (the conditioned
 * decoders sum it per frame, model.py:180); g_out always receives the group's bottom gradient.  One partial slab per
 * workgroup (`nslabs` from srwn_group_wt_geometry; slabs beyond it must stay zero); finish with srwn_reduce_partials,
 * sqrt(.5) on the residual pair.  Halo (sum(dilations)/gcd) <= 31.
 * part16 != 0 (dtype SRWN_BF16 only): part_f / part_r are written in the COMPUTE type instead of fp32 -- the same number
 * of elements per (layer, slab), as 16 x 16 blocks in lane order (layout SRWN_PARTIALS_BLK16 of SrwnReduceJob below):
 * half the bytes both ways for one more bf16 rounding per partial sum (the 256 x 30 partials of config 2 are 0.38 GB per
 * step in fp32, written here and read back by the reduction; measured cost in accuracy: DESIGN.md 4c).  The two bias
 * partials stay fp32.
 * ic_audio != NULL (the stack's FIRST group, dcs given; bf16 mode: with part16): the launch also leaves the partial sums
 * of the input conv's kernel and bias gradient (model.py:40; what srwn_init_conv_wgrad's first stage forms from g_out in a
 * launch of its own) -- ic_partials[slab][3 R] = [sum_t audio[t-1-ic_shift] G_0[t,:] | sum_t audio[t-ic_shift] G_0[t,:] |
 * sum_t G_0[t,:]] over rows the workgroup's segments own, fp32, 8 / (R/16) slabs per workgroup (the launch's waves split
 * the tiles between them): ic_partials holds nslabs * 8 / (R/16) slabs of 3 R floats, to be summed; the audio enters the
 * bf16 MFMA as a high and a low bf16 part (exact to 2^-17).  audio [B,T] fp32. */
int srwn_group_wt_geometry(const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R, int32_t dtype,
                           int32_t seg_rows_in, int32_t* seg_rows, int32_t* tiles_per_seg, int64_t* elems_per_layer,
                           int32_t* nslabs);
int srwn_residual_group_fwd_wt(const void* x0, void* x_out, void* z_out, int64_t layer_stride, void* xT, void* cT,
                               int64_t wt_layer_stride, int32_t store_inner_x, const void* const* wconv, const void* const* wres,
                               const float* const* bias_f, const float* const* bias_r, const void* const* cond_next,
                               int32_t cond_frames, int32_t pool_stride, int32_t cond_row_stride,
                               const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R, int32_t K,
                               int32_t seg_rows, int32_t dtype, void* stream);
/* the FIRST group of a stack with the stack's input conv fused in (model.py:40 / 172-173: DilatedCausalConv1d 1 -> R,
 * K = 2 taps, d = 1; RightShift, ops.py:78-80, as `shift` in {0, 1}): what srwn_causal_conv1d_fwd(audio, init_w, init_b)
 * would have written as the group's input is computed into the kernel's segment image in the same arithmetic (the
 * activations are bit-identical) and never reaches HBM -- one launch, a 2*R-byte-per-sample write and the read that
 * fetches it back less.  audio [B,T] fp32, init_w [2,1,R], init_b [R] fp32.  Otherwise srwn_residual_group_fwd_wt (xT / cT
 * required).  Not built for the conditioned decoders (their first layer's bias is added
 * to the input conv's output: srwn_add_frame_bias). */
int srwn_residual_group_fwd_ic(const float* audio, const float* init_w, const float* init_b, int32_t shift, void* x_out,
                               void* z_out, int64_t layer_stride, void* xT, void* cT, int64_t wt_layer_stride,
                               int32_t store_inner_x, const void* const* wconv, const void* const* wres,
                               const float* const* bias_f, const float* const* bias_r, const int32_t* dilations,
                               int32_t nlayers, int32_t B, int32_t T, int32_t R, int32_t K, int32_t seg_rows,
                               int32_t dtype, void* stream);
int srwn_residual_group_bwd_wt(const void* g_top, void* g_out, int32_t write_all_g, const void* z, const void* dcs,
                               int64_t layer_stride, const void* xT, const void* cT, int64_t wt_layer_stride,
                               const void* const* wconvT, const void* const* wresT, const int32_t* dilations,
                               int32_t nlayers, void* part_f, void* part_r, float* part_bf, float* part_br,
                               int32_t part16, const float* ic_audio, float* ic_partials, int32_t ic_shift,
                               int32_t nslabs, int32_t B, int32_t T, int32_t R, int32_t K, int32_t seg_rows,
                               int32_t dtype, void* stream);
/* the same cut chosen for a problem size (B clips of T steps, R channels, dtype): minimises the estimated run time of
 * the group kernels (tile rounds per layer + a fixed cost per launch) over all cuts into runs of <= max_layers layers. */
int32_t srwn_group_plan_auto(const int32_t* dilations, int32_t nlayers, int32_t B, int32_t T, int32_t R, int32_t dtype,
                             int32_t max_layers, int32_t* starts);
/* ---- the remaining free functions of ops.py (none on the timed path; fp32, plain VALU kernels):
 * log_prob_from_logits (ops.py:111-115) -> y [rows,C] and / or log_sum_exp (ops.py:117-122) -> lse [rows] (either NULL) */
int srwn_log_softmax(const float* x, float* y, float* lse, int64_t rows, int32_t C, void* stream);
/* categorical_sample (ops.py:106-109): one index per row drawn from softmax(logits) (tf.multinomial's stream is not
 * reproducible; this one is counter-based: the same seed gives the same draws) */
int srwn_categorical_sample(const float* logits, int32_t* out, int64_t rows, int32_t C, uint64_t seed, void* stream);
/* probs_logistic (ops.py:203-214): sigmoid((y-mu+h)/s) - sigmoid((y-mu-h)/s), h = 1/(num_classes-1),
 * s = max(scale, exp(log_scale_min)) */
int srwn_probs_logistic(const float* scale, const float* mu, const float* y, float* out, int64_t n,
                        int32_t num_classes, float log_scale_min, void* stream);
/* pieces of ResidualDilationLayer / ResidualDilationLayerNC for shapes outside the fused kernels (any filter_width, any
 * channel counts; ops.py:232-236 builds an 8-channel layer on a 1-channel input): z = tanh(f), c = z*sigmoid(z)
 * (ops.py:28,33,36); dense = (inputs + residual)*sqrt(.5) with a 1-channel input broadcast (ops.py:40); relu (ops.py:49,52) */
int srwn_tanh_gate(const float* f, float* z, float* c, int64_t n, void* stream);
/* the gated activation unit with SURVEY 8(b)'s gate_mode: z = tanh(f) and
 *   SRWN_GATE_REFERENCE: c = z * sigmoid(z)   -- the graph the reference RUNS (ops.py:33 overwrites the gate conv's result
 *                                                with sigmoid(filter_conv); g is not read and may be NULL) = srwn_tanh_gate
 *   SRWN_GATE_WAVENET:   c = z * sigmoid(g)   -- the canonical WaveNet unit ops.py:31-32 builds and then discards; g = the
 *                                                gate conv's output (srwn_causal_conv1d_fwd with the `_gate` kernel)
 * Forward, fp32, any shape: the ops-level ResidualDilationLayer(gate_mode="wavenet") of sr-wavenet_amd/ops.py runs on it.
 * The fused training kernels (srwn_residual_layer_* / srwn_residual_group_*) implement SRWN_GATE_REFERENCE only: parity is
 * judged on the graph the reference executes, and its `_gate` variables receive no gradient there. */
#define SRWN_GATE_REFERENCE 0
#define SRWN_GATE_WAVENET 1
int srwn_gated_activation(const float* f, const float* g, float* z, float* c, int64_t n, int32_t gate_mode, void* stream);
int srwn_residual_combine(const float* x, int32_t cin, const float* res, int32_t R, float* out, int64_t rows,
                          void* stream);
int srwn_relu(const float* x, float* y, int64_t n, void* stream);
/* discretized_mix_logistic_loss with sum_all=False (ops.py:174-175): out[row] = -log_sum_exp_m(log p_m(x) + log pi_m) */
int srwn_mol_nll_rows(const float* logits, int64_t ldl, const float* x, int32_t M, float* out, int64_t rows,
                      void* stream);

/* diagnostic hook (no reference counterpart).  Only the -DSRWN_DIAG build of this library (libsrwn_diag.so:
 * `python sr-wavenet_amd/build.py --diag`, loaded with SRWN_LIB_PATH) holds the stamped kernel instantiations: there,
 * while a device buffer of 1024 uint64 is registered, the bf16 group kernels, the skip sum and the one-launch head append
 * in-kernel clock stamps of workgroup 0 to it; NULL restores production code.  The shipped libsrwn.so accepts NULL and
 * returns SRWN_E_UNSUPPORTED for a buffer. */
int srwn_debug_stamp_buffer(void* device_buffer);
/* greedy cut of a stack's dilation list (model.py:9, teacher.py:57) into such groups: starts[0..n] (starts[n] = nlayers),
 * returns n.  `starts` needs nlayers + 1 entries. */
int32_t srwn_group_plan(const int32_t* dilations, int32_t nlayers, int32_t max_halo, int32_t max_layers,
                        int32_t* starts);

/* ---- pointwise linear ("channels GEMM"): tf.layers.conv1d kernel_size=1 (ops.py:39,44;
 * model.py:53,56,180) and the sum of all skip 1x1s (model.py:50) as one K = L*R contraction:
 *   y[row, n] = epi( bias[n] + sum_k pro(x[row, k]) * W[k, n] )
 * Input channel k lives at  x + (k / chunk_len)*x_chunk_stride + row*x_row_stride + k % chunk_len
 * (chunk = one layer's z tensor for the skip sum; chunk_len = Cin for an ordinary tensor).
 * wpack: packed [Cout_pad/32][Cin/16] natural k order.  Rows n >= cout_valid are not stored.
 * EPI_MASK multiplies by (aux[row, n] > 0) (aux row stride = aux_row_stride elements). */
int srwn_pw_linear(const void* x, int64_t x_row_stride, int64_t x_chunk_stride, int32_t chunk_len, int32_t Cin,
                   const void* wpack, const float* bias, void* y, int64_t y_row_stride, int32_t cout_pad,
                   int32_t cout_valid, int64_t rows, const void* aux, int64_t aux_row_stride, int32_t pro,
                   int32_t epi, int32_t dtype, void* stream);

/* the same product (no prologue / epilogue) with its outputs in chunks of y_chunk_len channels: channel n at
 * y + (n / y_chunk_len)*y_chunk_stride + row*y_row_stride + n % y_chunk_len -- the conditioning biases of all layers
 * (model.py:180: L*R outputs) stored layer by layer, [L][rows][R], so that a layer's rows are dense. */
int srwn_pw_linear_ychunks(const void* x, int64_t x_row_stride, int32_t Cin, const void* wpack, const float* bias, void* y,
                           int64_t y_row_stride, int32_t y_chunk_len, int64_t y_chunk_stride, int32_t cout_pad,
                           int32_t cout_valid, int64_t rows, int32_t dtype, void* stream);

/* the same product split over the contraction axis (few rows, long K): slice z of nsplit writes its fp32 partial to
 * y_partials + z*rows*y_row_stride (bias in slice 0); srwn_reduce_partials(nslabs = nsplit, n = rows*y_row_stride)
 * finishes it.  Used for the encoder's pooled skip sum (model.py:150: B*frames rows, K = L*encoder_channels). */
int srwn_pw_linear_ksplit(const void* x, int64_t x_row_stride, int64_t x_chunk_stride, int32_t chunk_len, int32_t Cin,
                          const void* wpack, const float* bias, float* y_partials, int64_t y_row_stride,
                          int32_t cout_pad, int32_t cout_valid, int64_t rows, int32_t nsplit, int32_t dtype,
                          void* stream);

/* ---- last 1x1 + softmax head: model.py:56 then the mu-law softmax-CE the reference carries at
 * model.py:100-112 (log-softmax as ops.py:111-115), per time step, fused in registers:
 *   logits = bias + x @ W;  loss_row = logsumexp(logits) - logits[target]
 *   dlogits = (softmax(logits) - onehot(target)) * grad_scale   -> dlogits (dtype), may be NULL
 * logits_out (fp32, [rows, cout_valid]) may be NULL.  Per-tile loss sums go to loss_partials
 * (srwn_softmax_ce_partials(rows) floats); srwn_reduce_loss sums them in a fixed order. */
int64_t srwn_softmax_ce_partials(int64_t rows);
int srwn_head_softmax_ce(const void* x, int64_t x_row_stride, int32_t Cin, const void* wpack, const float* bias,
                         const int32_t* targets, float* loss_partials, void* dlogits, float* logits_out,
                         int32_t cout_pad, int32_t cout_valid, int64_t rows, float grad_scale, int32_t dtype,
                         void* stream);
int srwn_reduce_loss(const float* loss_partials, int64_t n, float scale, float* loss_out, void* stream);

/* ---- the whole head of the softmax teacher, forward and backward, in one launch (bf16, S = cout_pad = 256):
 *   r1 = relu(r0 @ W1 + b1)                      model.py:53-54
 *   logits = r1 @ W2 + b2; loss_row / dlogits as srwn_head_softmax_ce      model.py:56, 100-112
 *   da1 = (dlogits @ W2^T) * (r1 > 0);  dtotal = (da1 @ W1^T) * (r0 > 0)   (autodiff of model.py:51-56)
 * A wave carries 32 rows through the four products in registers; r1, dlogits, da1, dtotal ([rows, 256] each) are
 * written for the weight-gradient passes.  w1 is the srwn_pack image of W1 in natural k order; w2_perm, w2T_perm,
 * w1T_perm are the images of W2, W2^T, W1^T in the accumulator's (permuted) k order.  loss_partials as
 * srwn_head_softmax_ce.  Other shapes/dtypes: SRWN_E_UNSUPPORTED (the four separate entry points remain). */
int srwn_head_chain(const void* r0, const void* w1, const void* w2_perm, const void* w2T_perm, const void* w1T_perm,
                    const float* b1, const float* b2, const int32_t* targets, float* loss_partials, void* r1,
                    void* dlogits, void* da1, void* dtotal, int32_t S, int32_t cout_pad, int32_t cout_valid,
                    int64_t rows, float grad_scale, int32_t dtype, void* stream);

/* ---- fused residual layer backward: autodiff of ResidualDilationLayer (ops.py:23-46); TF builds
 * these gradients in tf.train.AdamOptimizer.minimize (model.py:31).  One call per layer, top down:
 *   has_up  : G_{l+1}[t] = g_in[t]*sqrt(.5) + sum_k Wf_{l+1}[k] . df_up[t + (K-1-k)*dilation_up]  -> g_out
 *             (g_in = G_{l+2}, may be NULL = 0; wconvT_up packed [R/32][K*R/16], rows = in channel)
 *   has_down: dc = Wr_l . (G_{l+1} sqrt(.5)) + Ws_l . dtotal;  df_l = dc * d(z sigmoid z)/df      -> df_out
 *             (wresT packed permuted [R/32][R/16]; z = z_l; the skip term Ws_l . dtotal is either `dcs`
 *             = layer l's slice from srwn_skip_dgrad_all, or computed here from wskipT [R/32][S/16] + dtotal)
 * Layer L-1: has_up=0 (its dense output is unused, model.py:45-50).  Below layer 0: has_down=0.
 * Flow stacks of ParallelWaveNet (model.py:415-453) have no skip path: pass S=0 (dc = Wr_l . G sqrt(.5) only);
 * their layer L-1 runs with has_up=2: G_L (the flow head's gradient, srwn_flow_affine_bwd) is READ from g_out. */
int srwn_residual_layer_bwd(const void* g_in, const void* df_up, const void* wconvT_up, void* g_out,
                            const void* wresT, const void* wskipT, const void* dtotal, const void* dcs,
                            const void* z, void* df_out, int32_t B, int32_t T, int32_t R, int32_t S, int32_t K,
                            int32_t dilation_up, int32_t has_up, int32_t has_down, int32_t dtype, void* stream);

/* ---- weight gradients (the tf.gradients of every kernel on the path), batched over `nbatch` layers:
 *   partials[l][slab][i][o] = sum_{rows of slab} pro(in_l[row - shifts[l], i] + cond_l[b, (t-shift)/pool, i])
 *                                                 * dout_l[row, o]          (0 where t - shift < 0)
 *   bias_partials[l][slab][o] = sum_{rows of slab} dout_l[row, o]           (may be NULL)
 * in_l = in + l*in_batch_stride (elements), dout_l likewise (stride 0 = shared by all layers);
 * rows = B*T flattened, `T` delimits batch elements for the shift.  pro = SRWN_PRO_GATE rebuilds
 * c = z*sigmoid(z).  nslabs = srwn_wgrad_slabs(rows); partials need nbatch*nslabs*cin*cout floats.
 * srwn_reduce_partials then writes out[l*out_batch_stride + i] = scale * sum_slab partials[...]
 * (fixed order, f64 accumulate); partials_batched=0 re-reads batch entry 0 for every l. */
int32_t srwn_wgrad_slabs(int64_t rows);
int srwn_wgrad(const void* in, int64_t in_batch_stride, int32_t cin, const void* dout, int64_t dout_batch_stride,
               int32_t cout, const void* cond, int64_t cond_batch_stride, int32_t cond_frames, int32_t pool_stride,
               int32_t cond_row_stride, const int32_t* shifts /* host array [nbatch] or NULL */, int32_t nbatch, float* partials,
               float* bias_partials, int64_t rows, int32_t T, int32_t nslabs, int32_t pro, int32_t dtype,
               void* stream);
int srwn_reduce_partials(const float* partials, int32_t nslabs, int64_t n, int32_t nbatch, int32_t partials_batched,
                         float scale, float* out, int64_t out_batch_stride, void* stream);
/* Up to 16 such reductions as ONE launch (host array of jobs; each job's result is bit-identical to its own
 * srwn_reduce_partials call). */
#define SRWN_PARTIALS_F32 0   /* fp32 [batch][slab][n]: what srwn_reduce_partials takes */
#define SRWN_PARTIALS_BLK16 1 /* bf16 [batch][slab][n] where the n = rows*blk_cols elements of a [rows, blk_cols] matrix
                               * are stored as 16 x 16 blocks in MFMA-accumulator lane order: block (rb, cb) at
                               * (rb*(blk_cols/16) + cb)*256 elements, lane l's four values -- rows 16 rb + 4 (l >> 4) + 0..3
                               * of column 16 cb + (l & 15) -- at + 4 l (srwn_residual_group_bwd_wt with part16) */
#define SRWN_PARTIALS_SUM 2   /* fp32; ONE output: out[0] = scale * the sum of all nslabs*n values, in srwn_reduce_loss's
                               * order and precision (the loss partials of a training step: bit-equal to that launch) */
typedef struct SrwnReduceJob {
  const void* partials; int32_t nslabs; int64_t n; int32_t nbatch; int32_t partials_batched; float scale;
  float* out; int64_t out_batch_stride;
  int32_t layout; int32_t blk_cols;      /* SRWN_PARTIALS_*; blk_cols: BLK16 only (a multiple of 16) */
} SrwnReduceJob;
int srwn_reduce_partials_multi(const SrwnReduceJob* jobs, int32_t njobs, void* stream);

/* ---- adjoint of ResizeEmbeddingNearestNeighbor (ops.py:64-74): out[b,e,c] = sum_{t in frame e} g[b,t,c] */
/* x[b,t,c] += bias[b, t/pool_stride, c] in place (h = h + upsampled, model.py:181-183, for the first layer) */
int srwn_add_frame_bias(void* x, const void* bias, int64_t bias_row_stride, int32_t B, int32_t T, int32_t C,
                        int32_t frames, int32_t pool_stride, int32_t dtype, void* stream);
int srwn_frame_sum(const void* g, void* out, int32_t B, int32_t T, int32_t C, int32_t frames, int32_t pool_stride,
                   int32_t dtype, void* stream);
/* the same for `nbatch` layers in one launch: g + l*g_batch_stride -> out + l*out_batch_stride (elements), times
 * `scale` (1/pool_stride gives tf.nn.pool AVG, model.py:154) */
int srwn_frame_sum_batched(const void* g, int64_t g_batch_stride, void* out, int64_t out_batch_stride, int32_t nbatch,
                           int32_t B, int32_t T, int32_t C, int32_t frames, int32_t pool_stride, float scale,
                           int32_t dtype, void* stream);

/* ---- tf.train.AdamOptimizer update (model.py:31,117,382) on the flat fp32 parameter buffer:
 *   t = ++*step (device counter);  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  g = grads*grad_scale;
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  params -= lr_t * m / (sqrt(v) + eps) */
int srwn_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step, float lr,
                   float beta1, float beta2, float eps, float grad_scale, void* stream);

/* ---- time-pooled classifier head of class WaveNet (model.py:56-60; loss model.py:24-29).
 * tf.nn.pool AVG over the whole clip commutes with the last 1x1, so:
 *   srwn_time_mean   : out[b,c] = mean_t x[b,t,c]   (partials: B*srwn_time_mean_slabs(T)*C floats)
 *   srwn_pooled_head : logits = mean @ w2 + b2; probs = softmax (model.py:60); with labels also
 *                      loss = mean_b softmax_cross_entropy_with_logits_v2 (soft labels, model.py:29),
 *                      gw2/gb2 (written, [S,ldw]/[ldw]) and dmean = d loss / d mean  [B,S]
 *   srwn_bcast_mask  : da1[b,t,s] = (r1[b,t,s] > 0) ? dmean[b,s]*scale : 0  (scale = 1/T) */
int32_t srwn_time_mean_slabs(int32_t T);
int srwn_time_mean(const void* x, float* partials, float* out, int32_t B, int32_t T, int32_t C, int32_t dtype,
                   void* stream);
int srwn_pooled_head(const float* mean, const float* w2, const float* b2, const float* labels, float* probs,
                     float* loss, float* gw2, float* gb2, float* dmean, int32_t B, int32_t S, int32_t C,
                     int32_t ldw, void* stream);
int srwn_bcast_mask(const float* dmean, const void* r1, void* out, int32_t B, int32_t T, int32_t S, float scale,
                    int32_t dtype, void* stream);

/* ---- weight gradient of 256-wide products as ONE time-contraction GEMM (tf.gradients of ops.py:44,
 * model.py:53,56 kernels):  partials[slab][m][n] = sum_{rows of slab} pro(A[row][m]) * D[row][n], n < 256.
 * A is addressed in chunks of 64 channels: a + (m/64)*a_chunk_stride + row*a_row_stride + m%64
 * (the [L,rows,64] stack of z for all skip 1x1s at once, or a [rows,256] tensor with chunk stride 64).
 * bias_partials[slab][n] = column sums of D (may be NULL).  nslabs = srwn_wgrad256_slabs(rows, m_chunks);
 * finish with srwn_reduce_partials. */
int32_t srwn_wgrad256_slabs(int64_t rows, int32_t m_chunks);
/* the same GEMM for the reference scripts' widths: A chunks of `chunk_width` = 64 or 32 channels
 * (dilation_channels), D of `d_width` = 256 or 128 columns (skip_channels); m_chunks*chunk_width must be a
 * multiple of 64; partials [slab][m_chunks*chunk_width][d_width]. */
int32_t srwn_wgrad_wide_slabs(int64_t rows, int32_t m_chunks, int32_t chunk_width);
int srwn_wgrad_wide(const void* a, int64_t a_chunk_stride, int64_t a_row_stride, int32_t m_chunks, int32_t chunk_width,
                    const void* d, int64_t d_row_stride, int32_t d_width, float* partials, float* bias_partials,
                    int64_t rows, int32_t nslabs, int32_t pro, int32_t dtype, void* stream);
/* two such products of ONE shape as one launch (the head's two 1x1s, model.py:53,56: 46 us each alone -- 256 workgroups
 * of 4 MFMAs per barrier -- side by side their workgroups share the CUs and hide each other's waits) */
int srwn_wgrad_wide_pair(const void* a0, const void* d0, float* partials0, float* bias_partials0, const void* a1,
                         const void* d1, float* partials1, float* bias_partials1, int64_t a_chunk_stride,
                         int64_t a_row_stride, int32_t m_chunks, int32_t chunk_width, int64_t d_row_stride,
                         int32_t d_width, int64_t rows, int32_t nslabs, int32_t pro, int32_t dtype, void* stream);
int srwn_wgrad256(const void* a, int64_t a_chunk_stride, int64_t a_row_stride, int32_t m_chunks, const void* d,
                  int64_t d_row_stride, float* partials, float* bias_partials, int64_t rows, int32_t nslabs,
                  int32_t pro, int32_t dtype, void* stream);

/* ---- skip-path data gradient of every layer in one launch (autodiff of ops.py:44):
 *   dcs[l][row][n] = sum_s dtotal[row][s] * Ws_l[n][s],   dcs + l*dcs_layer_stride, rows of R elements.
 * wskipT_all: the nlayers packed images [R/32][S/16] (natural k order) laid out back to back.
 * srwn_residual_layer_bwd then takes dcs_l instead of recomputing Ws_l . dtotal (pass wskipT = dtotal = NULL). */
int srwn_skip_dgrad_all(const void* dtotal, const void* wskipT_all, void* dcs, int64_t dcs_layer_stride,
                        int32_t nlayers, int64_t rows, int32_t R, int32_t S, int32_t dtype, void* stream);

/* ---- all per-layer weight gradients of ResidualDilationLayer (ops.py:27,39) in one pass over the
 * saved tensors, batched over layers ([L][rows][64] stacks, `layer_stride` elements apart; R=64, K=2):
 *   part_f [l][slab][k*64+i][o] = sum x_l[t-(1-k)*d_l, i] * df_l[t, o]   (x_l + cond_l when cond != NULL)
 *   part_r [l][slab][n][m]      = sum c_l[t, n] * g_l[t, m]              (c = z sigmoid z; g_l = G_{l+1})
 *   part_bf[l][slab][o] = sum df_l[t,o];  part_br[l][slab][m] = sum g_l[t,m]
 * dilations: host array [nlayers].  Finish with srwn_reduce_partials (sqrt(.5) on the residual pair). */
int srwn_wgrad_layers(const void* x, const void* z, const void* df, const void* g, int64_t layer_stride,
                      const void* cond, int64_t cond_layer_stride, int32_t cond_frames, int32_t pool_stride,
                      int32_t cond_row_stride, const int32_t* dilations, int32_t nlayers, float* part_f,
                      float* part_r, float* part_bf, float* part_br, int64_t rows, int32_t T, int32_t nslabs,
                      int32_t R, int32_t K, int32_t dtype, void* stream);

/* ---- weight gradient of all skip 1x1s from the forward group kernel's transposed gate outputs (tf.gradients of
 * ops.py:44 summed as model.py:50):  out[l*64 + n][s] = sum_t c_l[t][n] * d[t][s].  cT / wt_layer_stride: the cT buffer
 * srwn_residual_group_fwd_wt wrote (layer l at cT + l * wt_layer_stride elements); st[l] / seg_rows[l]: the stride (gcd of
 * the dilations) and the segment length (srwn_group_wt_geometry) of the group layer l was run in -- they fix which
 * positions its tiles hold; d: dskip [B*T, d_row_stride] (256 columns used).  Partials in srwn_wgrad256's layout:
 * partials[slab][nlayers*64][256], bias_partials[slab][256] (column sums of d; may be NULL); nslabs from
 * srwn_wgrad_skip_wt_slabs.  part16 != 0: `partials` holds the same [nlayers*64, 256] matrix per slab in bf16, as 16 x 16
 * blocks in lane order (SRWN_PARTIALS_BLK16 with 256 columns: half the partial bytes both ways, one more bf16 rounding
 * per partial sum).  bf16, R = 64, S = 256 (csrc/srwn_wgradt.hip); other shapes: srwn_wgrad_wide on z. */
int32_t srwn_wgrad_skip_wt_slabs(const int32_t* st, const int32_t* seg_rows, int32_t nlayers, int32_t T);
int srwn_wgrad_skip_wt(const void* cT, int64_t wt_layer_stride, const int32_t* st, const int32_t* seg_rows,
                       int32_t nlayers, const void* d, int64_t d_row_stride, void* partials, float* bias_partials,
                       int32_t part16, int32_t nslabs, int32_t B, int32_t T, int32_t R, int32_t S, int32_t dtype,
                       void* stream);

/* ---- queue-cached incremental generation (BASELINE config 5; the reference only has the O(T^2 L) loop
 * of teacher.py:140-171).  Persistent workgroups generate `nsteps` samples, 32 utterances per workgroup,
 * with the arithmetic of the training graph (RightShift input conv model.py:172-173, layers ops.py:23-46,
 * head model.py:50-56, softmax over C mu-law classes, decode ops.py:96-104), keeping per layer a ring of
 * the last d_l+1 layer inputs (`ring`: ceil(B/32) * srwn_generate_ring_elems elements of `dtype`; one
 * workgroup per group of 32 utterances).
 * wcr: per layer, back to back, [conv image R/32 x 2R/16 (tap 0 natural, tap 1 permuted k order) |
 * residual image R/32 x R/16 (permuted)];  wskip: [S/32][L*R/16] in PERMUTED k order (its B operand is the
 * gate tile in registers); w1/w2/biases as for the training kernels.
 * mode 0 = argmax, 1 = categorical sample (counter-based RNG on seed, utterance, step).
 * forced != NULL: teacher forcing -- step t consumes forced[u, t-1] instead of its own sample (parity test).
 * audio_out/codes_out/forced are [B, Tout]; logits_out (may be NULL) [B, Tout, C] fp32. */
int64_t srwn_generate_ring_elems(const int32_t* dilations, int32_t nlayers, int32_t R);
int srwn_generate(const void* wcr, const void* wskip, const void* w1, const void* w2, const float* bias_f,
                  const float* bias_r, const float* bs_sum, const float* b1, const float* b2, const float* init_w,
                  const float* init_b, void* ring, float* audio_out, int32_t* codes_out, float* logits_out,
                  const float* forced, const int32_t* dilations, int32_t nlayers, int32_t B, int32_t Tout,
                  int32_t nsteps, int32_t R, int32_t S, int32_t C, int32_t K, int32_t mode, uint64_t seed,
                  int32_t dtype, void* stream);

/* The latency-optimised body of srwn_generate / srwn_generate_mol for bf16 stacks of R = 64 or 32 residual and S = 256 or
 * 128 skip channels, K = 2: same arithmetic, same rings (handed over ZERO-FILLED: a step before the first delayed tap
 * exists reads the slot nobody has written), same outputs, same RNG (teacher.py:140-171 / generator.py:150-170 are the
 * loops it replaces), but a layer's channels are split over the four waves on 16x16x32 tiles instead of every wave
 * running the whole chain, and the weights stream from L2 into registers one layer ahead (csrc/srwn_gen16.hip).
 * Fragment images of 64 lanes x 8 elements, lane l = row (l & 15), k = 8 (l >> 4) + j.  wl: per layer [4 waves][conv
 * k-steps 0..2R/32-1 (k < R: delayed tap, k >= R: current tap) | residual k-steps 0..R/32-1 | skip (S/64 row blocks) x
 * (R/32 k-steps)]; wave w owns conv / residual rows 16w.. (none for w >= R/16: zero fragments) and skip rows (S/4)w..;
 * wh1: [4 waves][S/64 row blocks][S/32 k-steps], wave w, block rb = rows (S/4) w + 16 rb; wh2: [4 waves][4 row
 * blocks][S/32 k-steps], rows 16 (4 rb + w) -- a head with few outputs still splits over the waves; rows beyond
 * ceil(C/32)*32 zero.  srwn_generate16_image_elems(nlayers, 0 | 1 | 2, R, S) = elements of wl | wh1 | wh2. */
int64_t srwn_generate16_image_elems(int32_t nlayers, int32_t which, int32_t R, int32_t S);
int srwn_generate16(const void* wl, const void* wh1, const void* wh2, const float* bias_f, const float* bias_r,
                    const float* bs_sum, const float* b1, const float* b2, const float* init_w, const float* init_b,
                    void* ring, float* audio_out, int32_t* codes_out, float* logits_out, const float* forced,
                    const int32_t* dilations, int32_t nlayers, int32_t B, int32_t Tout, int32_t nsteps, int32_t R,
                    int32_t S, int32_t C, int32_t mode, uint64_t seed, void* stream);
/* ... and for the conditioned mixture-of-logistics decoder (the model generator.py:150-170 samples from; arguments as
 * srwn_generate_mol): cond [B*cond_frames, cond_ld] bf16 (cond_ld a multiple of 4) or NULL. */
int srwn_generate16_mol(const void* wl, const void* wh1, const void* wh2, const float* bias_f, const float* bias_r,
                        const float* bs_sum, const float* b1, const float* b2, const float* init_w,
                        const float* init_b, void* ring, float* audio_out, int32_t* codes_out, float* logits_out,
                        const float* forced, const int32_t* dilations, int32_t nlayers, int32_t B, int32_t Tout,
                        int32_t nsteps, int32_t R, int32_t S, int32_t num_mixtures, const void* cond,
                        int32_t cond_frames, int32_t pool_stride, int64_t cond_ld, int32_t mode, uint64_t seed,
                        void* stream);

/* The same generator for the conditioned mixture-of-logistics decoder of WaveNetAutoEncoder (model.py:158-200; the
 * reference samples it with one whole-clip pass per sample, generator.py:150-170): cond [B*cond_frames, cond_ld] in
 * `dtype` holds the conditioning biases cb_l of every layer at columns [l*R, (l+1)*R) (model.py:180: one
 * srwn_pw_linear of encoding_w_condition); layer l adds row (u, t / pool_stride), rounded like the training kernel.
 * cond = NULL: unconditioned.  Head: 4*num_mixtures logits (w2 image / b2 padded to a multiple of 32 rows), sampled
 * as sample_from_discretized_mix_logistic (ops.py:178-201) with counter-based uniforms in (1e-5, 1-1e-5);
 * mode 0 returns the selected mixture's mean.  codes_out = selected mixture; logits_out [B, Tout, 4M] (may be NULL). */
int srwn_generate_mol(const void* wcr, const void* wskip, const void* w1, const void* w2, const float* bias_f,
                      const float* bias_r, const float* bs_sum, const float* b1, const float* b2, const float* init_w,
                      const float* init_b, void* ring, float* audio_out, int32_t* codes_out, float* logits_out,
                      const float* forced, const int32_t* dilations, int32_t nlayers, int32_t B, int32_t Tout,
                      int32_t nsteps, int32_t R, int32_t S, int32_t K, int32_t num_mixtures, const void* cond,
                      int32_t cond_frames, int32_t pool_stride, int64_t cond_ld, int32_t mode, uint64_t seed,
                      int32_t dtype, void* stream);

/* ---- discretised mixture-of-logistics loss of the reference's live teacher:
 * discretized_mix_logistic_loss (ops.py:124-175, sum_all=True) on logits [rows, ldl] fp32 whose first 4*M
 * columns are (logit_probs, means, log_scales, coeffs) and targets x [rows] in [-1,1]:
 *   loss = -sum_rows logsumexp_m( log p_m(x) + log_softmax(logit_probs)_m )   (bin half-width 1/255, log-scale
 *   floor -7, the four tf.where branches of ops.py:169; the coeffs never reach the loss)
 * loss_partials: one float per 256 rows (sum with srwn_reduce_loss);  dlogits [rows, ldd] in `dtype`,
 * = d loss / d logits * grad_scale, columns >= 4*M written as 0. */
int srwn_mol_loss(const float* logits, int64_t ldl, const float* x, int32_t M, float* loss_partials, void* dlogits,
                  int64_t ldd, int64_t rows, float grad_scale, int32_t dtype, void* stream);

/* Same loss on the student's output (model.py:374): the teacher logits are constants (stop_gradient,
 * model.py:334), the gradient wanted is dx[row] = d loss / d x[row] * grad_scale (fp32). */
int srwn_mol_loss_dx(const float* logits, int64_t ldl, const float* x, int32_t M, float* loss_partials, float* dx,
                     int64_t rows, float grad_scale, void* stream);

/* ---- Parallel-WaveNet student (class ParallelWaveNet, model.py:290-537).
 * A flow (createPartialFlow/createFlow, model.py:415-487) is the conditioned residual stack above WITHOUT the skip
 * path, then  prm = relu(h_L) @ W2[R,2] + b2;  scale = exp(prm0), mean = prm1;  x_out = x_in*scale + mean.
 *   srwn_flow_affine_fwd: prm [rows,2] and x_out [rows] (fp32); ent_partials[srwn_flow_partials(rows)] = block sums
 *                         of prm0 = log scale (entropy = sum over flows + 2*rows, model.py:356)
 *   srwn_flow_affine_bwd: dprm0 = dx_out*x_in*scale + ent_grad, dprm1 = dx_out;  dx_in = dx_out*scale;
 *                         g [rows,R] (dtype) = (h_L > 0) * (dprm @ W2^T) = G_L for srwn_residual_layer_bwd(has_up=2);
 *                         w_partials[block][2R+2] = (relu(h_L)^T dprm | column sums of dprm): srwn_reduce_partials
 * ent_grad carries d(-alpha*entropy/B)/d prm0 = -alpha/B (model.py:376-379). */
int64_t srwn_flow_partials(int64_t rows);
int srwn_flow_affine_fwd(const void* h, const float* w2, const float* b2, const float* x_in, float* prm, float* x_out,
                         float* ent_partials, int64_t rows, int32_t R, int32_t dtype, void* stream);
int srwn_flow_affine_bwd(const void* h, const float* w2, const float* prm, const float* x_in, const float* dx_out,
                         float ent_grad, void* g, float* dx_in, float* w_partials, int64_t rows, int32_t R,
                         int32_t dtype, void* stream);

/* ---- out = tf.minimum(tf.maximum(x, lo), hi) (model.py:535) and its gradient dx = dy where lo <= x <= hi, else 0
 * (dx may alias dy). */
int srwn_clamp(const float* x, float* y, int64_t n, float lo, float hi, void* stream);
int srwn_clamp_bwd(const float* x, const float* dy, float* dx, int64_t n, float lo, float hi, void* stream);

/* ---- data gradient of _DilatedCausalConv1d (ops.py:6-10) wrt a narrow input (the 1-channel flow input,
 * model.py:423-424); `shift` is the adjoint of RightShift (ops.py:78-80):
 *   dx[b,u,i] (+)= scale * sum_k sum_o w[k,i,o] * dy[b, u + shift + (K-1-k)*dilation, o]   (0 beyond the clip)
 * dy [B,T,Cout] in `dtype`, w [K,Cin,Cout] fp32, dx [B,T,Cin] fp32 (accumulate != 0 adds into it). */
int srwn_causal_conv1d_dgrad(const void* dy, const float* w, float* dx, int32_t B, int32_t T, int32_t Cin,
                             int32_t Cout, int32_t K, int32_t dilation, int32_t shift, int32_t accumulate, float scale,
                             int32_t dtype, void* stream);

/* ---- STFT power loss (model.py:360-371): tf.contrib.signal.stft(x, 512, 256) -> frames without end padding
 * (srwn_stft_frames(T) = 1 + (T-512)/256), periodic Hann window, 512-point real DFT (257 bins);
 *   srwn_stft_power    : power[b,f] = mean_frames |X[b,n,f]|^2;  spec [B,frames,257,2] keeps (Re,Im) for the
 *                        backward (may be NULL); frame_power [B,frames,257] is scratch
 *   srwn_power_loss    : loss = gamma * sum (power_truth - power_out)^2 (tf.norm(.)**2);  dpower (may be NULL)
 *                        = d(loss*grad_scale)/d power_out
 *   srwn_stft_power_bwd: dx[b,t] (+)= sum_f dpower[b,f] * d power[b,f] / d x[b,t] */
int32_t srwn_stft_frames(int32_t T);
int srwn_stft_power(const float* x, float* spec, float* frame_power, float* power, int32_t B, int32_t T, void* stream);
int srwn_power_loss(const float* power_truth, const float* power_out, int64_t n, float gamma, float grad_scale,
                    float* dpower, float* loss, void* stream);
int srwn_stft_power_bwd(const float* spec, const float* dpower, float* dx, int32_t B, int32_t T, int32_t accumulate,
                        void* stream);

/* ---- tf.clip_by_global_norm(grads, clip_norm) (model.py:385) + the Adam update with the clip factor on device:
 *   srwn_sumsq      : partials[srwn_sumsq_partials(n)] = chunk sums of g^2 (call once per gradient buffer, into
 *                     consecutive slices of one partials array)
 *   srwn_clip_scale : norm = pre_scale*sqrt(sum partials); out[0] = pre_scale*clip_norm/max(norm, clip_norm);
 *                     out[1] = norm   (pre_scale = 1/world when the buffers hold an all-reduced SUM)
 *   srwn_adam_step_scaled: srwn_adam_step with grad_scale read from device memory; tick=0 shares one step
 *                     counter between several parameter buffers (tick it on the first call of a step only) */
/* y += alpha * scale_dev[0] * x: accumulates per-row clipped gradients of the slow path ParallelWaveNet.train
 * (model.py:599-632) with the clip factor of srwn_clip_scale left on the device */
int srwn_axpy_dev(float* y, const float* x, const float* scale_dev, float alpha, int64_t n, void* stream);
int64_t srwn_sumsq_partials(int64_t n);
int srwn_sumsq(const float* g, int64_t n, float* partials, void* stream);
int srwn_clip_scale(const float* partials, int64_t n, float clip_norm, float pre_scale, float* out, void* stream);
int srwn_adam_step_scaled(float* params, const float* grads, float* m, float* v, int64_t n, int64_t* step, float lr,
                          float beta1, float beta2, float eps, const float* grad_scale_dev, int32_t tick,
                          void* stream);

/* ---- WaveNetAutoEncoder pieces (model.py:75-285).  The encoder (createEncoder, model.py:136-156) is a chain of
 * ResidualDilationLayerNC (ops.py:48-58): relu -> tf.layers.conv1d(K, SAME; the dilation argument is never passed
 * on) -> relu, then 1x1 residual and 1x1 skip; skips summed, 1x1 to latent_channels, average-pooled.
 *
 * srwn_tap_linear: time-tap GEMM on MFMA for 128 or 256 output channels --
 *   y[row][n] = epi( bias[n] + frame_add[clip*frames + t/pool_stride][n]*frame_add_scale
 *                    + sum_{tap<ntaps} sum_i x[row + tap*tap_step][i] * W[tap*Cin + i][n] )
 *   rows = clips*T; a tap that leaves its clip contributes 0 (SAME padding of the K=2 conv: tap_step=+1; its
 *   data gradient: tap_step=-1 with the transposed kernel); wpack = MFMA image [cout/32][ntaps*Cin/16] (natural k);
 *   epi = SRWN_EPI_NONE / _RELU / _MASK (aux > 0); frame_add (fp32, may be NULL) is the broadcast of a per-frame
 *   term (the pooled skip path's gradient, the adjoint of tf.nn.pool AVG model.py:154). */
int srwn_tap_linear(const void* x, int64_t x_row_stride, int32_t ntaps, int32_t tap_step, int32_t T, int32_t Cin,
                    const void* wpack, const float* bias, void* y, int64_t y_row_stride, int32_t cout, int64_t rows,
                    const void* aux, int64_t aux_row_stride, const float* frame_add, int64_t frame_add_ld,
                    int32_t frames, int32_t pool_stride, float frame_add_scale, int32_t epi, int32_t dtype,
                    void* stream);
/* One ResidualDilationLayerNC of the encoder as ONE launch (ops.py:48-58; bf16, 128 channels, K = 2 taps at t and
 * t+1; other shapes: SRWN_E_UNSUPPORTED -- use srwn_tap_linear):
 *   a_out[t] = relu(bias_c + sum_k r_in[t+k] . Wconv[k])     (a tap beyond the clip contributes 0)
 *   r_out[t] = relu(bias_r + a_out[t] . Wr)                   (the relu'd input of the next layer; NULL: skipped --
 *                                                              the last layer's residual output is unused, model.py:144-150)
 * wconv = MFMA image [4][16] natural k (k = tap*128 + in channel), wres = image [4][8] in permuted k order (its B
 * operand is the first product's accumulator tile).  a_bits / r_bits (may be NULL): srwn_nc_mask_words(B,T) 64-bit
 * words receiving the relu masks of a_out / r_out in the layout srwn_nc_layer_bwd reads (word [tile*64 + lane], one bit
 * per accumulator register of the lane: opaque outside srwn_nc.hip). */
int64_t srwn_nc_mask_words(int32_t B, int32_t T);
int srwn_nc_layer_fwd(const void* r_in, const void* wconv, const void* wres, const float* bias_c, const float* bias_r,
                      void* a_out, void* r_out, uint64_t* a_bits, uint64_t* r_bits, int32_t B, int32_t T, int32_t C,
                      int32_t K, int32_t dtype, void* stream);
/* the same mask words for a [B,T,128] bf16 tensor written by another kernel (x > 0) */
int srwn_nc_mask_bits(const void* x, uint64_t* bits, int32_t B, int32_t T, int32_t C, int32_t dtype, void* stream);
/* Its data gradient, pairing the conv of layer l with the 1x1 of the layer below (autodiff of ops.py:50-57):
 *   dh_out[t]   = [r[t] > 0] . sum_k dpre_up[t-k] . Wconv[k]^T                                      (= d loss / d r_l)
 *   dpre_out[t] = [a[t] > 0] . (dh_out[t] . Wr_below^T + frame_add[clip*frames + t/pool]*scale)       (NULL: skipped)
 * r_bits / a_bits = mask words of the layer's relu'd input and of the activation under the 1x1 below;
 * wconvT = image [4][16] natural k (rows = in channel, k = tap*128 + out channel), wresT = image [4][8] permuted k;
 * frame_add (fp32, may be NULL) is the pooled skip path's gradient broadcast over its frame (model.py:154). */
int srwn_nc_layer_bwd(const void* dpre_up, const void* wconvT, const uint64_t* r_bits, void* dh_out, const void* wresT,
                      const float* frame_add, int64_t frame_add_ld, int32_t frames, int32_t pool_stride,
                      float frame_add_scale, const uint64_t* a_bits, void* dpre_out, int32_t B, int32_t T, int32_t C,
                      int32_t K, int32_t dtype, void* stream);
/* weight gradients of the encoder's ResidualDilationLayerNC chain (ops.py:48-58), all layers in one pass over the saved
 * tensors ([L][rows][128] stacks, `layer_stride` elements apart; 128 channels, K = 2 taps at t and t+1):
 *   part_w [l][slab][k*128+i][o] = sum r_l[t+k, i] * dpre_l[t, o]   (taps beyond the clip contribute 0)
 *   part_r [l][slab][n][m]       = sum a_l[t, n] * dh_l[t, m];  part_b / part_br = column sums of dpre_l / dh_l
 * Finish with srwn_reduce_partials. */
int srwn_wgrad_nc_layers(const void* r, const void* a, const void* dpre, const void* dh, int64_t layer_stride,
                         int32_t nlayers, float* part_w, float* part_r, float* part_b, float* part_br, int64_t rows,
                         int32_t T, int32_t nslabs, int32_t C, int32_t K, int32_t dtype, void* stream);
/* first encoder layer on the raw clip (model.py:141-142): a[b,t,c] = relu(bias[c] + sum_k w[k][c]*relu(x[b,t+k])) */
int srwn_nc_input_fwd(const float* x, const float* w, const float* bias, void* a, int32_t B, int32_t T, int32_t C,
                      int32_t K, int32_t dtype, void* stream);
/* small products on the [B*frames] axis (latent 1x1 model.py:152, gradient wrt the encoding through model.py:180):
 *   C[m][n] = (accumulate ? C : 0) + bias[n] + sum_k A(m,k)*B(k,n), chunked addressing on both operands:
 *   A(m,k) = a[(k/a_chunk)*a_chunk_stride + m*lda + k%a_chunk];  B(k,n) = b[(k/b_chunk)*b_chunk_stride + (k%b_chunk)*ldb_k + n*ldb_n]
 *   srwn_small_wgrad: c[k][n] = scale*sum_m a[m][k]*d[m][n], bias_out[n] = scale*sum_m d[m][n] (may be NULL) */
int srwn_small_gemm(const void* a, int64_t lda, int32_t a_chunk, int64_t a_chunk_stride, int32_t a_dtype,
                    const float* b, int64_t ldb_k, int64_t ldb_n, int32_t b_chunk, int64_t b_chunk_stride,
                    const float* bias, void* c, int64_t ldc, int32_t c_dtype, int32_t M, int32_t N, int32_t K,
                    int32_t accumulate, void* stream);
int srwn_small_wgrad(const float* a, int64_t lda, const float* d, int64_t ldd, float* c, float* bias_out, int32_t M,
                     int32_t K, int32_t N, float scale, void* stream);
/* sample_from_discretized_mix_logistic (ops.py:178-201) given the uniform draws u1 [rows,M], u2 [rows] in
 * (1e-5, 1-1e-5): Gumbel-max mixture choice, logistic sample, clip to [-1,1] -> out [rows] */
int srwn_mol_sample(const float* logits, int64_t ldl, int32_t M, const float* u1, const float* u2, float* out,
                    int64_t rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRWN_H */
