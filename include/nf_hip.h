/* nf_hip.h - C ABI of libnf_hip.so: the MI355X (gfx950) kernels behind the flow-transform hot path.
 *
 * The reference (EricLiangOS/Normalizing-Flows) has no FFI: its boundary for this path is the
 * Python nn.Module surface.  This library is what the drop-in Python modules in
 * normalizing-flows_amd/ call through ctypes; each entry point below names the reference lines whose
 * ATen op sequence it replaces.
 *
 * Conventions (all entry points):
 *   - plain C types only: raw DEVICE pointers, element counts, strides in ELEMENTS, enums as int;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); work is asynchronous on it;
 *   - return value: 0 = ok; negative = argument error (NF_ERR_*), nothing was launched;
 *     positive = a hipError_t from the launch;
 *   - nothing is allocated or freed, nothing synchronises, nothing throws; thread-safe for distinct streams;
 *   - the caller owns every buffer, including the packed weights and the K-tile table.
 */
#ifndef NF_HIP_H
#define NF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NF_OK 0
#define NF_ERR_NULL (-1)    /* a required pointer is NULL */
#define NF_ERR_SHAPE (-2)   /* sizes inconsistent or outside the supported range */
#define NF_ERR_LAYOUT (-3)  /* stride / alignment not accepted */
#define NF_ERR_BINS (-4)    /* min_bin_width * K > 1 or min_bin_height * K > 1
                               (reference: ValueError, rational_quadratic.py:66-69) */
#define NF_ERR_DTYPE (-5)   /* unknown dtype / layout enum */
#define NF_ERR_SCRATCH (-6)

#define NF_DTYPE_F32 0      /* v_mfma_f32_32x32x2_f32: exact fp32 products (1e-5 contract) */
#define NF_DTYPE_BF16 1     /* v_mfma_f32_32x32x16_bf16 on bf16-rounded operands (2e-2 contract) */
#define NF_DTYPE_F16X2 3    /* fp32 operands as two fp16 terms (weights scaled by a power of two chosen at pack time), three
                             * fp16 MFMAs per product, hidden state scaled per launch likewise: same 1e-5 contract; fused entry points only.
                             * Its pack call needs ktab (the scale pair is kept in ktab[2*tiles .. 2*tiles+2)). */
#define NF_DTYPE_BF16X3 2   /* fp32 operands as three bf16 terms, six bf16 MFMAs per product, fp32 accumulate: fp32-level
                             * accuracy (1e-5 contract) on the bf16 matrix pipe; nf_made_rqs_* / nf_coupling_rqs_fwd only */

#define NF_AFFINE_INTERLEAVED 0 /* params[b,2d]=unconstrained scale, [b,2d+1]=shift  (transforms.py:561-566) */
#define NF_AFFINE_BLOCKED 1     /* params[b,d]=shift, params[b,D+d]=unconstrained scale (transforms.py:328-333) */

/* ABI version: major*100 + minor. */
int nf_version(void);

/* Static description of a return code (never NULL). */
const char* nf_strerror(int code);

/* ---- MADE masked linear (replaces models/made.py:150-152 `F.linear(x, W*mask, b)` and the
 *      activation / BatchNorm(eval) / skip ops around it, made.py:196-207, 268-284) ------------- */

/* Bytes of the packed-weight buffer / int32 entries of the K-tile table for an [out,in] layer. */
size_t nf_made_packed_bytes(int out, int in, int dtype);
int nf_made_ktab_len(int out);

/* Fold `mask` (NULL = dense layer) into row-major W[out,in], zero-pad to whole tiles, convert to
 * `dtype`, and record per 128-row tile the range of 32-wide K tiles holding unmasked weights.
 * Replaces the per-call `self.weight * self.mask` of made.py:152; run once per weight update.
 * ktab == NULL re-folds the weights only and leaves the caller's table alone (the mask, hence the
 * table, does not change between optimizer steps); the same holds for nf_made_pack_t / nf_made_rqs_pack. */
int nf_made_pack(const float* W, const float* mask, int out, int in, int dtype,
                 void* packed, int32_t* ktab, void* stream);

/* y[B,out] = act_out( act_in(x*in_scale + in_shift) @ Wm^T + bias (+ residual) ).
 *   relu_in / relu_out: 0 or 1 (F.relu, the reference default activation);
 *   in_scale/in_shift: NULL, or per-input-feature eval-mode BatchNorm folded to scale and shift;
 *   residual: NULL or [B,out] added before act_out (made.py:284 `inputs + temps`).
 * x and y are fp32 whatever `dtype` the matrix cores multiply in. */
int nf_masked_linear_fwd(const float* x, int64_t x_row_stride, const void* packed_w, const int32_t* ktab,
                         const float* bias, const float* residual, int64_t residual_row_stride,
                         const float* in_scale, const float* in_shift, float* y, int64_t y_row_stride,
                         int64_t B, int in, int out, int relu_in, int relu_out, int dtype, void* stream);

/* Same, restricted to the 128-column output tiles [n_tile_begin, n_tile_begin + n_tile_count): other
 * columns of y are left untouched.  Used by the incremental autoregressive inverse, where one pass only
 * changes the hidden units whose degree equals the feature that just became final (the reference
 * recomputes the whole network D times, transforms.py:493-500). */
int nf_masked_linear_fwd_tiles(const float* x, int64_t x_row_stride, const void* packed_w, const int32_t* ktab,
                               const float* bias, const float* residual, int64_t residual_row_stride,
                               const float* in_scale, const float* in_shift, float* y, int64_t y_row_stride,
                               int64_t B, int in, int out, int relu_in, int relu_out, int dtype,
                               int n_tile_begin, int n_tile_count, void* stream);

/* ---- rational-quadratic spline (replaces rational_quadratic.py:13-149, helper_functions.py
 *      searchsorted :507-512 and sum_except_batch :393-398; callers transforms.py:212-247, 819-860) -- */

/* For every sample b < B and feature d < D, with P = 3K-1 (tails=1) or 3K+1 (tails=0) parameters at
 * params[b*param_row_stride + d*P ...] = [K widths | K heights | derivatives]:
 *   tails=1: identity with zero logdet outside [-tail_bound, tail_bound], spline inside, boundary
 *            slopes pinned through the constant log(exp(1-min_derivative)-1);
 *   tails=0: spline on [0,1] (inputs outside are clamped to the first / last bin: the reference raises);
 *   pre_div: unnormalised widths and heights are divided by it first (sqrt(hidden_features) on the
 *            coupling path, transforms.py:217-219; 1.0 = off);
 *   element d reads x[b*x_row_stride + c] and writes y[b*y_row_stride + c], c = feat_index ? feat_index[d] : d;
 *   logabsdet[b] = (accumulate ? logabsdet[b] : 0) + sum_d log|dy/dx|   (negated sum for inverse);
 *   bin_idx (nullable, [B,D] contiguous): the bin each element fell in, -1 in the tails - the
 *            searchsorted result of helper_functions.py:507-512, the bit-exact parity target.
 * Requires param_row_stride == D*P (parameter rows contiguous) and K <= 32. */
int nf_rqs_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index,
               const float* params, int64_t param_row_stride, double pre_div,
               float* y, int64_t y_row_stride, float* logabsdet, int accumulate, int32_t* bin_idx,
               int64_t B, int D, int K, int tails, double tail_bound,
               double min_bin_width, double min_bin_height, double min_derivative,
               int inverse, void* stream);

/* ---- linear and quadratic spline families (same launch shape and argument meaning as
 *      nf_rqs_fwd) --------------------------------------------------------------------------------- */

/* Piecewise-linear CDF on K equal bins (replaces splines/linear.py:35-94 `linear_spline`, and with tails != 0
 * :9-33 `unconstrained_linear_spline`): params[b, d*K + k] = unnormalised pdf.  tails == 0: x in [0, 1]. */
int nf_linear_spline_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index, const float* params,
                         int64_t param_row_stride, float* y, int64_t y_row_stride, float* logabsdet, int accumulate,
                         int32_t* bin_idx, int64_t B, int D, int K, int tails, double tail_bound, int inverse,
                         void* stream);
/* Piecewise-quadratic CDF (replaces splines/quadratic.py:48-135 `quadratic_spline` / :11-46
 * `unconstrained_quadratic_spline`): per element K unnormalised widths then K+1 (tails == 0) or K-1
 * (linear tails) unnormalised heights.  NF_ERR_BINS as in nf_rqs_fwd. */
int nf_quadratic_spline_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index, const float* params,
                            int64_t param_row_stride, float* y, int64_t y_row_stride, float* logabsdet,
                            int accumulate, int32_t* bin_idx, int64_t B, int D, int K, int tails, double tail_bound,
                            double min_bin_width, double min_bin_height, int inverse, void* stream);

/* Monotone piecewise-cubic spline on [-bound, bound] (replaces splines/cubic.py:56-221 `cubic_spline`, whose
 * default bound is 10, and with tails != 0 :14-54 `unconstrained_cubic_spline`): per element K unnormalised
 * widths, K unnormalised heights, the unnormalised left and right boundary derivatives (P = 2K + 2).
 * pre_div: the coupling layers' division of the width / height logits by sqrt(hidden_features). */
int nf_cubic_spline_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index, const float* params,
                        int64_t param_row_stride, double pre_div, float* y, int64_t y_row_stride, float* logabsdet,
                        int accumulate, int32_t* bin_idx, int64_t B, int D, int K, int tails, double bound,
                        double min_bin_width, double min_bin_height, int inverse, void* stream);

/* ---- final MADE layer fused with the spline: `params` never reaches HBM ---------------------------
 *      (replaces made.py:358 `self.final_layer(outputs)` + transforms.py:819-860 `_elementwise` +
 *       rational_quadratic.py:13-149 in one launch; the forward AND the per-pass inverse of the
 *       autoregressive transform, transforms.py:488-500, go through it) ------------------------------ */

/* 1 if the fused kernel handles this spline shape (4 <= K <= 32); the hidden width must also be a
 * multiple of 8.  Otherwise callers use nf_masked_linear_fwd + nf_rqs_fwd. */
int nf_made_rqs_supported(int K, int tails);
/* Column tiles (256 packed weight rows = 256/P whole features each) for `features` features. */
int nf_made_rqs_tiles(int features, int K, int tails);
size_t nf_made_rqs_packed_bytes(int features, int in, int K, int tails, int dtype);
/* fp32 scratch for the per-tile logdet partial sums: tiles * B floats. */
size_t nf_made_rqs_workspace_bytes(int64_t B, int features, int K, int tails);
/* Same for a given dtype: the split-operand dtypes add a 256-byte scale slot (required) and room for the
 * hidden state split into term planes (terms * B * ceil16(in) * 2 bytes; without it, and in the windowed
 * launches, the kernel splits on the fly).  nf_made_rqs_workspace_bytes + 512 covers the slot alone. */
size_t nf_made_rqs_workspace_bytes_dtype(int64_t B, int features, int in, int K, int tails, int dtype);
/* Packs the final layer W[features*P, in] (mask folded, rows regrouped to whole features per tile),
 * its K-range table (2*tiles + 2 int32: the last two hold NF_DTYPE_F16X2's scale pair) and its bias
 * (tiles*256 floats). */
int nf_made_rqs_pack(const float* W, const float* mask, const float* bias, int features, int in, int K,
                     int tails, int dtype, void* packed, int32_t* ktab, float* bias_packed, void* stream);
/* Same with the width / height logits divided by `logit_div` first (the coupling layers' in-place
 * `/= sqrt(hidden_features)`, transforms.py:221-229); the division is folded into the packed rows. */
int nf_made_rqs_pack_scaled(const float* W, const float* mask, const float* bias, int features, int in, int K,
                            int tails, int dtype, double logit_div, void* packed, int32_t* ktab, float* bias_packed,
                            void* stream);
/* y, logabsdet (and optionally bin_idx) of the spline whose parameters are h @ Wm^T + bias; semantics of
 * x, y, logabsdet, accumulate, bin_idx, tails, tail_bound, min_* and inverse as in nf_rqs_fwd
 * (no feat_index, pre_div = 1: the autoregressive path). */
int nf_made_rqs_fwd(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                    const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                    int64_t y_row_stride, float* logabsdet, int accumulate, int32_t* bin_idx,
                    float* workspace, size_t workspace_bytes, int64_t B, int in, int features, int K,
                    int tails, double tail_bound, double min_bin_width, double min_bin_height,
                    double min_derivative, int inverse, int dtype, void* stream);
/* nf_made_rqs_fwd (forward direction, NF_DTYPE_F32 / NF_DTYPE_BF16) that also KEEPS the conditioner output: params_out
 * [B, features * P] as nf_masked_linear_fwd would have written it (made.py:358), logits unfolded with `logit_div` (the
 * value given to nf_made_rqs_pack_scaled, 1 for nf_made_rqs_pack).  The training forward: nf_rqs_bwd recomputes from
 * (x, params), and the parameters are not read back between the GEMM and the spline.  NF_DTYPE_F16X2 (round 3): the
 * same on the f16x2 "t4" kernel - whole layers only, with the workspace of nf_made_rqs_workspace_bytes_dtype (logdet
 * partials + scale slot + term planes), else NF_ERR_SCRATCH and the caller runs nf_masked_linear_split_fwd + nf_rqs_fwd. */
int nf_made_rqs_fwd_params_supported(int K, int tails, int dtype);   /* 1: nf_made_rqs_fwd_params takes such packs */
int nf_made_rqs_fwd_params(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                           const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                           int64_t y_row_stride, float* logabsdet, int accumulate, float* workspace,
                           size_t workspace_bytes, int64_t B, int in, int features, int K, int tails,
                           double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                           int dtype, double logit_div, float* params_out, int64_t params_row_stride, void* stream);

/* Same, restricted to column tiles [tile_begin, tile_begin + tile_count) and to features < feat_limit
 * (later features of the last tile are neither read nor written); the per-tile logdet partial sums stay
 * in `workspace` between calls and logabsdet is only produced when `reduce` != 0 (the last pass). */
int nf_made_rqs_fwd_tiles(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                          const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                          int64_t y_row_stride, float* logabsdet, int accumulate, int32_t* bin_idx,
                          float* workspace, size_t workspace_bytes, int64_t B, int in, int features, int K,
                          int tails, double tail_bound, double min_bin_width, double min_bin_height,
                          double min_derivative, int inverse, int dtype, int tile_begin, int tile_count,
                          int feat_limit, int reduce, void* stream);

/* Coupling layer (replaces the conditioner's final Linear, resnet.py:99, + transforms.py:156-168, :212-247
 * `_coupling_transform` / `_piecewise_cdf` + rational_quadratic.py:13-149): h [B, in] is the ResidualNet's last
 * hidden state, packed_w from nf_made_rqs_pack_scaled(mask = NULL); feature f of the spline reads
 * x[b, feat_index[f]] and writes y[b, feat_index[f]] (rows `row_width` columns wide; the identity columns of
 * y are the caller's).  The [B, features*P] parameter tensor is never written. */
int nf_coupling_rqs_fwd(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                        const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                        int64_t y_row_stride, const int32_t* feat_index, int row_width, float* logabsdet,
                        int accumulate, int32_t* bin_idx, float* workspace, size_t workspace_bytes, int64_t B,
                        int in, int features, int K, int tails, double tail_bound, double min_bin_width,
                        double min_bin_height, double min_derivative, int inverse, int dtype, void* stream);

/* nf_made_rqs_fwd with `h_max`: the slot a split-operand producer (nf_masked_linear_split_fwd's y_max) left
 * max |h| in, which saves the kernel's own pass over h (NULL: as nf_made_rqs_fwd), and `y_max` (nullable,
 * split-operand dtypes only): receives max |y| for the next layer's first GEMM. */
int nf_made_rqs_fwd_hmax(const float* h, int64_t h_row_stride, const uint32_t* h_max, const void* packed_w,
                         const int32_t* ktab, const float* bias_packed, const float* x, int64_t x_row_stride,
                         float* y, int64_t y_row_stride, uint32_t* y_max, float* logabsdet, int accumulate, int32_t* bin_idx,
                         float* workspace, size_t workspace_bytes, int64_t B, int in, int features, int K,
                         int tails, double tail_bound, double min_bin_width, double min_bin_height,
                         double min_derivative, int inverse, int dtype, void* stream);

/* nf_made_rqs_fwd_params with the conditioner output kept as bf16 [B, features * P] (16-byte aligned, row stride in
 * elements) and the spline backward that reads it (forward direction, no pre-divisor; gradients as bf16 like
 * nf_rqs_bwd_max_bf16; D * P % 8 == 0).  For the bf16 training mode: everything derived from the kept parameters is
 * rounded to bf16 by the backward GEMMs anyway; y / logabsdet of the forward come from the fp32 accumulators.
 * y_max (nullable): receives the bit pattern of max |y| - the operand scale the next layer's weight gradient wants of
 * its input, so no nf_maxabs pass over y.  No counterpart in the reference (autograd keeps fp32 throughout). */
int nf_made_rqs_fwd_params_bf16(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                                const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                                int64_t y_row_stride, uint32_t* y_max, float* logabsdet, int accumulate, float* workspace,
                                size_t workspace_bytes, int64_t B, int in, int features, int K, int tails,
                                double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                                int dtype, double logit_div, void* params_out_bf16, int64_t params_row_stride,
                                void* stream);
int nf_rqs_bwd_max_bf16p(const float* x, int64_t x_row_stride, const void* params_bf16, int64_t param_row_stride,
                         const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet, float* grad_x,
                         int64_t grad_x_row_stride, void* grad_params_bf16, uint32_t* grad_params_max, int64_t B, int D,
                         int K, int tails, double tail_bound, double min_bin_width, double min_bin_height,
                         double min_derivative, void* stream);

/* ---- one RQ-spline coupling layer in one launch ---------------------------------------------------------
 *      (replaces transforms.py:80-106 `CouplingTransform.forward` / :108-135 `inverse` with the conditioner of
 *      nn/resnet.py:90-100 and `_piecewise_cdf`, transforms.py:212-247: nf_resnet_hidden_fwd_copy +
 *      nf_coupling_rqs_fwd as a single kernel) --------------------------------------------------------------
 * A 256-thread workgroup owns 64 whole rows: rows in (coalesced), hidden stack in LDS, final Linear in chunks of
 * 4 features (P = 23) on bf16 MFMA with three-term split operands, spline evaluation with lane = sample, whole rows
 * out (identity columns copied, transformed columns from the spline).  The 1e-5 contract.
 * hidden_blob: nf_resnet_blob_floats layout; w_packed (nf_coupling_final_packed_bytes) / b_packed [features * P] from
 * nf_coupling_final_pack: the final Linear as three bf16 term planes per value in MFMA operand order (six bf16 MFMAs
 * per product = the fp32 product to 2^-22, the NF_DTYPE_BF16X3 arithmetic), logit rows folded with log2(e) /
 * logit_div (logit_div = sqrt(hidden) for ResidualNet, transforms.py:214-216).  in_index [n_in] / feat_index
 * [features]: int32 columns of the identity / transformed halves.  Limits (nf_coupling_layer_supported): row_width
 * <= 512, n_in <= 128, hidden 32 or 64, P <= 128, LDS <= 160 KB.  The pack depends on the chunk geometry the library
 * picks for (features, K, tails): pack and launch with the same library build and environment. */
int nf_coupling_layer_supported(int row_width, int n_in, int hidden, int features, int K, int tails);
size_t nf_coupling_final_packed_bytes(int features, int hidden, int K, int tails);
int nf_coupling_final_pack(const float* W, const float* bias, int features, int hidden, int K, int tails,
                           double logit_div, void* w_packed, float* b_packed, void* stream);
int nf_coupling_rqs_layer_fwd(const float* x, int64_t x_row_stride, float* y, int64_t y_row_stride, int row_width,
                              const int32_t* in_index, int n_in, const int32_t* feat_index, int features,
                              const float* hidden_blob, int hidden, int num_blocks, const void* w_packed,
                              const float* b_packed, float* logabsdet, int accumulate, int32_t* bin_idx, int64_t B,
                              int K, int tails, double tail_bound, double min_bin_width, double min_bin_height,
                              double min_derivative, int inverse, void* stream);

/* ---- the MADE hidden stack in one launch (NF_DTYPE_F16X2 arithmetic) -----------------------------------
 *      (replaces made.py:352-357 `initial_layer` + `blocks`: made.py:150-152 per layer, the residual blocks of
 *      :268-284 or the feed-forward blocks of :196-207, ReLU, no context, no BatchNorm) ------------------
 * A 512-thread workgroup carries 64 samples through every hidden layer: the layer input lives in LDS as two fp16
 * term planes scaled by a power of two taken from THAT 64-row tile, the hidden state stays in fp32 accumulator
 * registers, and only the final hidden state leaves the CU - as the tiled fp16 term planes nf_made_rqs_fwd_planes
 * reads, with max |h| of each 64-row group in row_scale.
 * Limits (nf_made_hidden_supported): 1..8 layers, hidden a multiple of 32 and <= 512, and no hidden unit of the
 * first layer may read an input column >= 512 (k_live0 = 1 + the last input column with an unmasked weight;
 * MADE's degree-ordered masks give k_live0 = hidden).
 * types[l]: 0 = initial layer (l == 0 only), 1 / 2 = first / second linear of a residual block, 3 = feed-forward
 * block.  W / mask / bias: host arrays of n_layers DEVICE pointers ([hidden, in_features[l]] row-major; mask or
 * mask[l] NULL = dense; bias or bias[l] NULL = 0); in_features[l] = hidden for l > 0. */
int nf_made_hidden_supported(int n_layers, int hidden, int k_live0);
size_t nf_made_hidden_packed_bytes(int n_layers, int hidden, const int* in_features);
int nf_made_hidden_pack(const float* const* W, const float* const* mask, const float* const* bias,
                        const int* in_features, const int* types, int n_layers, int hidden, int k_live0,
                        void* packed, void* stream);
/* bytes of `planes` for B rows (whole 128-row tiles); row_scale holds 2 * ceil(B / 128) uint32 */
size_t nf_made_hidden_planes_bytes(int64_t B, int hidden);
/* x [B, in0] -> planes + row_scale (+ h_out [B, hidden] fp32 when not NULL; 16-byte aligned rows). */
int nf_made_hidden_planes_fwd(const float* x, int64_t x_row_stride, const void* packed, int n_layers, int hidden,
                              int in0, void* planes, uint32_t* row_scale, float* h_out, int64_t h_row_stride,
                              int64_t B, void* stream);
/* nf_made_rqs_fwd_hmax on the output of nf_made_hidden_planes_fwd instead of an fp32 hidden state: no max pass, no
 * split pass (packed_w from nf_made_rqs_pack with NF_DTYPE_F16X2; workspace per nf_made_rqs_workspace_bytes + 512). */
int nf_made_rqs_fwd_planes(const void* planes, const uint32_t* row_scale, const void* packed_w, const int32_t* ktab,
                           const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                           int64_t y_row_stride, uint32_t* y_max, float* logabsdet, int accumulate, int32_t* bin_idx,
                           float* workspace, size_t workspace_bytes, int64_t B, int in, int features, int K, int tails,
                           double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                           int inverse, void* stream);

/* nf_made_rqs_fwd WITHOUT a workspace argument and without any allocation.  No counterpart in the reference: it is
 * for callers that cannot allocate - code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  The per-tile logdet partials (for the split-operand dtypes also the hidden state's scale slot and, when
 * they fit, its term planes) live in an 8 MiB device array inside the library, one copy per GPU, held until the
 * process ends.  Limit: nf_made_rqs_tiles(features, K, tails) * B * 4 bytes (+ 512 for NF_DTYPE_F16X2 / BF16X3) must
 * fit 8 MiB (cfg-3 shape, 72 tiles: B <= 29127), else NF_ERR_SCRATCH and nothing is launched.  Calls no hipMalloc /
 * hipFree / allocating routine; the first call on a device resolves that device's copy of the array (make it outside
 * graph capture; B = 0 does that and launches nothing).  Every call on a GPU shares that GPU's array: calls that
 * overlap on one GPU (different streams or threads) race - serialise them.  Nothing in the array needs to start at
 * zero (every value read is written first by the same call).  Results are those of nf_made_rqs_fwd called with an
 * 8 MiB workspace, bit for bit: for NF_DTYPE_F32 / NF_DTYPE_BF16 that is nf_made_rqs_fwd's result with any workspace;
 * for NF_DTYPE_F16X2 / BF16X3 the kernel is chosen from the workspace size - when the hidden state's pre-split term
 * planes (terms * B * kpad(in) * 2 bytes beside the partials; in = 512: B <= ~3900) fit the array the plane kernel
 * runs, past that the kernel that splits the hidden state inside its K loop (slower, another accumulation order:
 * both meet the same contract against the reference, they are not equal bit for bit). */
int nf_made_rqs_fwd_noalloc(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                            const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                            int64_t y_row_stride, float* logabsdet, int accumulate, int32_t* bin_idx, int64_t B, int in,
                            int features, int K, int tails, double tail_bound, double min_bin_width,
                            double min_bin_height, double min_derivative, int inverse, int dtype, void* stream);

/* ---- hidden-layer GEMMs on split operands (NF_DTYPE_F16X2 / NF_DTYPE_BF16X3; replaces made.py:150-152 like
 *      nf_masked_linear_fwd, same fp32 contract, fp16 / bf16 matrix pipe) --------------------------------- */

/* 256-column output tiles; packed bytes; pack (ktab: 2*tiles + 2 int32, bias_packed: tiles*256 floats). */
int nf_linear_split_tiles(int out);
size_t nf_linear_split_packed_bytes(int out, int in, int dtype);
int nf_linear_split_pack(const float* W, const float* mask, const float* bias, int out, int in, int dtype,
                         void* packed, int32_t* ktab, float* bias_packed, void* stream);
/* The same after a weight update with an unchanged mask: packed / bias_packed / the operand scale in ktab are
 * rewritten, the K-range table in ktab (a scan of the whole mask) is kept.  Training repacks every step. */
int nf_linear_split_repack(const float* W, const float* mask, const float* bias, int out, int in, int dtype,
                         void* packed, int32_t* ktab, float* bias_packed, void* stream);
/* The pack of the TRANSPOSED layer (weight operand of grad_x = grad_y @ (W * mask), autograd of made.py:150-152) read
 * from the layer's own [out, in] arrays: packed_t per nf_linear_split_packed_bytes(in, out, dtype), ktab_t
 * 2 * nf_linear_split_tiles(in) + 2 int32, bias_zero 256 * nf_linear_split_tiles(in) floats (zero filled here).
 * fresh = 0 keeps the K-range table (mask unchanged since the last call on these buffers).  Also accepts
 * NF_DTYPE_BF16 like the calls above: one bf16 term per operand, the 2e-2 mode. */
int nf_linear_split_pack_t(const float* W, const float* mask, int out, int in, int dtype, void* packed_t,
                           int32_t* ktab_t, float* bias_zero, int fresh, void* stream);
/* slot <- bit pattern of the largest finite |x[b, c]| (any width / alignment; 16-byte loads when cols % 4 == 0 and
 * the rows are 16-byte aligned). */
int nf_maxabs(const float* x, int64_t x_row_stride, int64_t B, int cols, uint32_t* slot, void* stream);
/* y = [relu](x) @ (W*mask)^T + bias (+ residual) [relu].  x_max: slot holding max |x| (nf_maxabs, or the y_max of
 * the launch that produced x; required for NF_DTYPE_F16X2); y_max (nullable) receives max |y|.  in % 8 == 0.
 * relu_out == 2: `residual` is a gate instead - y is zeroed where residual <= 0 (the ReLU derivative when this
 * launch is the input-gradient GEMM of a layer that applied ReLU to its input). */
int nf_masked_linear_split_fwd(const float* x, int64_t x_row_stride, const uint32_t* x_max, const void* packed_w,
                               const int32_t* ktab, const float* bias_packed, const float* residual,
                               int64_t res_row_stride, float* y, int64_t y_row_stride, uint32_t* y_max, int64_t B,
                               int in, int out, int relu_in, int relu_out, int dtype, void* stream);

/* Input-gradient GEMM in the split-operand modes: grad_x[B, in] = grad_y[B, out] @ (W * mask), zeroed where
 * gate[B, in] <= 0 (gate NULL: no gating).  packed_wt / ktab_t / bias_zero = nf_linear_split_pack of the TRANSPOSED layer
 * ([in, out], transposed mask, no bias).  Same kernel as nf_masked_linear_split_fwd; each tile's live K range is walked
 * from its end down, which lines the column tiles of one row tile up on the same grad_y columns (one HBM fetch, the
 * rest from L2).  grad_x_max (nullable): receives max |grad_x| like y_max.  (Replaces autograd through made.py:150-152.) */
int nf_masked_linear_split_bwd_input(const float* grad_y, int64_t grad_y_row_stride, const uint32_t* grad_y_max,
                                     const void* packed_wt, const int32_t* ktab_t, const float* bias_zero,
                                     const float* gate, int64_t gate_row_stride, float* grad_x,
                                     int64_t grad_x_row_stride, uint32_t* grad_x_max, int64_t B, int out, int in, int dtype,
                                     void* stream);
/* The same with a second gradient joining behind the gate: grad_x = gate(grad_y @ (W * mask)) + add[B, in] - the
 * gradient that reaches a residual block's input along its skip connection (made.py:284 `inputs + temps`; autograd
 * sums the two there with an add pass, and the next split-operand launch needs a max pass over the sum).  gate and add
 * are required; grad_x_max is max |grad_x| of the sum. */
int nf_masked_linear_split_bwd_input_add(const float* grad_y, int64_t grad_y_row_stride, const uint32_t* grad_y_max,
                                         const void* packed_wt, const int32_t* ktab_t, const float* bias_zero,
                                         const float* gate, int64_t gate_row_stride, const float* add,
                                         int64_t add_row_stride, float* grad_x, int64_t grad_x_row_stride,
                                         uint32_t* grad_x_max, int64_t B, int out, int in, int dtype, void* stream);

/* ---- coupling conditioner: the ResidualNet's hidden stack in one launch ------------------------------
 *      (replaces resnet.py:96-98 `initial_layer` + `blocks` and the `inputs[:, identity_features]` gather of
 *       transforms.py:88; BatchNorm in eval mode folded to scale/shift; no context, ReLU only) --------- */

/* Length in floats of the parameter blob: Kp = n_in rounded up to 8,
 *   W0[hidden][Kp] (zero padded) b0[hidden]
 *   per block: s0[hidden] t0[hidden] Wa[hidden][hidden] ba[hidden] s1[hidden] t1[hidden] Wb[hidden][hidden] bb[hidden]
 * with s/t the eval-BatchNorm scale/shift applied before each ReLU (1 / 0 without BatchNorm). */
size_t nf_resnet_blob_floats(int n_in, int hidden, int num_blocks);
/* h[b, :] = hidden state after the last residual block for inputs x[b, in_index[0..n_in)] (in_index NULL:
 * the first n_in columns); rows of x are `row_width` columns wide.  hidden % 32 == 0, n_in, hidden <= 128. */
int nf_resnet_hidden_fwd(const float* x, int64_t x_row_stride, const int32_t* in_index, int n_in, int row_width,
                         const float* blob, int hidden, int num_blocks, float* h, int64_t h_row_stride,
                         int64_t B, void* stream);
/* The same launch, also writing the gathered columns to copy_out[b, in_index[c]] (rows `copy_row_stride` apart): the
 * identity half of a coupling layer reaches the layer's output (transforms.py:98-106) without a copy of the whole
 * input; the transformed half is then written by nf_coupling_rqs_fwd into the same buffer. */
int nf_resnet_hidden_fwd_copy(const float* x, int64_t x_row_stride, const int32_t* in_index, int n_in, int row_width,
                              const float* blob, int hidden, int num_blocks, float* h, int64_t h_row_stride,
                              float* copy_out, int64_t copy_row_stride, int64_t B, void* stream);

/* ---- autoregressive inverse, row-resident (replaces the D conditioner passes of transforms.py:493-500
 *      for MaskedPiecewiseRationalQuadraticAutoregressiveTransform, made.py:352-359 + rational_quadratic.py) ----
 * Requires hidden units in degree order (unit j has MADE degree j+1 in every hidden layer: the reference's
 * non-random masks with hidden <= features - 1, made.py:142-146), ReLU, BatchNorm folded (eval mode), no context, and
 * P = 3K-1 (linear tails) or 3K+1 <= 32 parameters per feature.  One workgroup carries 32 samples through
 * features 0..hidden; total work about one forward pass (see nf_inverse.hip). */
int nf_made_rqs_inverse_rows_supported(int features, int hidden, int n_steps, int K, int tails);
/* Length in floats of the parameter blob, HS = hidden rounded up to 64, H = hidden:
 *   per hidden step s (initial layer, then each masked linear of the blocks, n_steps in all):
 *       WT[H][HS]: WT[j][k] = weight[k][j] * mask[k][j] (input j < H -> unit k), bias[HS]
 *   bf[H+1][32]: final-layer bias of feature f, parameters 0..P-1, zero padded
 *   wl[H+1][2][32]: wl[f][0][c] = weight[f*P + c][f-1] * mask, wl[f][1][c] = weight[f*P + c][f-2] * mask: the two
 *       newest hidden units feature f reads (zero where there is none)
 *   dg[H][32]: unit j of hidden step s: dg[j][s] = its weight on unit j of step s-1 (s = 0: on input j),
 *       dg[j][8 + s] = its weight on unit j-1 (zero for j = 0), mask applied; dg[j][16 + s], dg[j][24 + s] = scale /
 *       shift of the eval-mode BatchNorm on the input of step s (made.py:271-277), 1 / 0 without BatchNorm
 *   wf[T][64][4], features 0..H back to back (feature f starts at super-step sum over g < f of ceil(g/8)):
 *       for feature f, K super-step S < ceil(f/8), lane (kh = lane>>5, col = lane&31):
 *       weight[f*P + col][8S + 4kh + 0..3] * mask (zero for col >= P and for the units f-2, f-1 kept in wl) */
size_t nf_made_inverse_blob_floats(int hidden, int n_steps);
/* x[b, 0..hidden] = inverse of z[b, 0..hidden] feature by feature; hidden_out[b, :] = final hidden state;
 * workspace = the [tiles][B] logdet partials of nf_made_rqs_fwd_tiles (tiles wholly below feature hidden+1
 * are written).  The caller finishes features hidden+1.. with nf_made_rqs_fwd_tiles(tile_begin =
 * (hidden+1) / features_per_tile, reduce = 1).  step_flags[s]: 1 = ReLU on the step's input, 2 = add the
 * output of step s-2 (residual block), 4 = ReLU on the step's output (feed-forward block).
 * Context-conditioned MADE (made.py:354-355, :279-283): context_term[b, :] = context_layer(context) is added to the
 * initial layer's output, gate0 / gate1 [b, :] = sigmoid(block.context_layer(context)) scale the second layer of
 * residual block 0 / 1 before the skip connection; all [B, hidden] with row stride context_row_stride, NULL without a
 * context. */
int nf_made_rqs_inverse_rows(const float* blob, int hidden, int n_steps, const int* step_flags, const float* z,
                             int64_t z_row_stride, float* x, int64_t x_row_stride, float* hidden_out,
                             int64_t h_row_stride, const float* context_term, const float* gate0, const float* gate1,
                             int64_t context_row_stride, float* workspace, size_t workspace_bytes, int64_t B, int features,
                             int K, int tails, double tail_bound, double min_bin_width, double min_bin_height,
                             double min_derivative, void* stream);

/* The same for MaskedAffineAutoregressiveTransform (transforms.py:511-566): blob with 2 parameters per feature
 * (unconstrained scale, shift interleaved, P = 2 in the layout above).  x[b, 0..hidden] and the final hidden state
 * are written; the caller finishes with the final layer on hidden_out (nf_masked_linear_fwd) and one nf_affine_fwd
 * (inverse = 1) over all features, which also gives logabsdet. */
int nf_made_affine_inverse_rows_supported(int features, int hidden, int n_steps);
int nf_made_affine_inverse_rows(const float* blob, int hidden, int n_steps, const int* step_flags, const float* z,
                                int64_t z_row_stride, float* x, int64_t x_row_stride, float* hidden_out,
                                int64_t h_row_stride, const float* context_term, const float* gate0, const float* gate1,
                                int64_t context_row_stride, int64_t B, int features, void* stream);

/* The same for the MaskedPiecewise{Linear,Quadratic,Cubic}AutoregressiveTransform classes (transforms.py:568-769,
 * splines/linear.py, quadratic.py, cubic.py): family 1 = linear (P = K parameters per feature), 2 = quadratic
 * (P = 2K+1, or 2K-1 with linear tails), 3 = cubic (P = 2K+2 on [-tail_bound, tail_bound], widths / heights divided
 * by pre_div); P <= 32.  As with the affine variant the caller finishes with the final layer on hidden_out and one
 * nf_{linear,quadratic,cubic}_spline_fwd(inverse = 1) over all features. */
int nf_made_family_inverse_rows_supported(int family, int features, int hidden, int n_steps, int K, int tails);
int nf_made_family_inverse_rows(int family, const float* blob, int hidden, int n_steps, const int* step_flags,
                                const float* z, int64_t z_row_stride, float* x, int64_t x_row_stride,
                                float* hidden_out, int64_t h_row_stride, const float* context_term, const float* gate0,
                                const float* gate1, int64_t context_row_stride, int64_t B, int features, int K,
                                int tails, double tail_bound, double min_bin_width, double min_bin_height,
                                double pre_div, void* stream);

/* ---- affine transform (replaces transforms.py:328-347 coupling, :540-566 autoregressive) -------- */

/* scale = sigmoid(u + 2) + 1e-3; forward y = scale*x + shift, inverse y = (x - shift)/scale;
 * logabsdet[b] (+)= +/- sum_d log(scale).  `layout` is NF_AFFINE_INTERLEAVED or NF_AFFINE_BLOCKED. */
int nf_affine_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index,
                  const float* params, int64_t param_row_stride, int layout,
                  float* y, int64_t y_row_stride, float* logabsdet, int accumulate,
                  int64_t B, int D, int inverse, void* stream);

/* ---- base density joined with the flow's logdet (replaces distributions.py:105-107 `SimpleNormal.log_prob` + the
 *      `log_prob + logabsdet` of distributions.py:154-157 and the `log_prob - logabsdet` of :173-186) ----------------- */

/* out[b] = lad_scale * logabsdet[b] + sum_d (-z[b,d]^2/2 - log(2 pi)/2); logabsdet NULL = the density alone.  `out`
 * may alias `logabsdet`.  Fixed-order reduction (deterministic). */
int nf_std_normal_log_prob(const float* z, int64_t z_row_stride, const float* logabsdet, double lad_scale, float* out,
                           int64_t B, int D, void* stream);

/* ---- backward kernels (maximum-likelihood training; the reference differentiates the same op sequences
 *      with autograd: test2.py:127-135, model.py:233-244) ---------------------------------------------- */

/* Transposed pack of (W*mask): the weight operand of grad_x = grad_y @ (W*mask).  Buffer sizes:
 * nf_made_packed_bytes(in, out, dtype) and nf_made_ktab_len(in). */
int nf_made_pack_t(const float* W, const float* mask, int out, int in, int dtype, void* packed_t,
                   int32_t* ktab_t, void* stream);
/* grad_x[B,in] = grad_y[B,out] @ (W*mask), zeroed where gate <= 0; gate = the layer's forward input when it
 * applied ReLU to it (relu_in), else NULL. */
int nf_masked_linear_bwd_input(const float* grad_y, int64_t grad_y_row_stride, const void* packed_wt,
                               const int32_t* ktab_t, const float* gate, int64_t gate_row_stride,
                               float* grad_x, int64_t grad_x_row_stride, int64_t B, int in, int out,
                               int dtype, void* stream);
/* grad_w[out,in] += mask * (grad_y^T @ act(x)); grad_b[out] += column sums of grad_y (either may be NULL).
 * ACCUMULATING: zero the buffers before the first call.  fp32 atomics are used when the batch is split
 * over workgroups (last bits depend on arrival order).  ktab (nullable) = the layer's forward K-tile table.
 * Any widths and alignment: 16-byte operand loads when in % 4 == 0, out % 4 == 0 and both row sets are 16-byte
 * aligned, bounded scalar loads otherwise (D = 2, D = 6 with K = 4, odd identity halves ...). */
int nf_masked_linear_bwd_weight(const float* x, int64_t x_row_stride, const float* grad_y,
                                int64_t grad_y_row_stride, const float* mask, const int32_t* ktab,
                                float* grad_w, float* grad_b, int64_t B, int in, int out, int relu_in,
                                int dtype, void* stream);
/* The same on split operands (two fp16 terms per value, fp32 accumulation; the NF_DTYPE_F16X2 contract of
 * nf_masked_linear_split_fwd).  x_max / grad_y_max: slots holding max |x| and max |grad_y| (nf_maxabs or the
 * y_max of the producing launch).  ACCUMULATING like nf_masked_linear_bwd_weight; no alignment requirement. */
int nf_masked_linear_split_bwd_weight(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                      const float* grad_y, int64_t grad_y_row_stride, const uint32_t* grad_y_max,
                                      const float* mask, const int32_t* ktab, float* grad_w, float* grad_b,
                                      int64_t B, int in, int out, int relu_in, void* stream);
/* The same with a caller-owned `workspace` of at least 2 * B * in * 2 bytes (16-byte aligned; in % 8 == 0, 16-byte
 * aligned rows): act(x) is written there first as two fp16 planes and fetched by LDS-DMA into the transposed-read
 * kernel (row-wise 16-byte fetches of grad_y, ds_read_b64_tr_b16 operand reads). */
int nf_masked_linear_split_bwd_weight_ws(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                         const float* grad_y, int64_t grad_y_row_stride, const uint32_t* grad_y_max,
                                         const float* mask, const int32_t* ktab, float* grad_w, float* grad_b,
                                         int64_t B, int in, int out, int relu_in, void* workspace,
                                         size_t workspace_bytes, void* stream);

/* The same weight gradient with ONE scaled fp16 term per operand (11 significant bits relative to each tensor's largest
 * element, fp32 accumulation, a third of the MFMAs): what the bf16 training mode uses (BASELINE config 5 "bf16 MADE
 * MFMA backward"; 2e-2 tolerance class).  Same arguments, plus an optional caller-owned `workspace` (NULL = none) of at
 * least B * in * 2 bytes, 16-byte aligned: wide layers (more than 128 output tiles of 128 x 256) then carry act(x) as an
 * fp16 plane written there first and fetched by LDS-DMA (half the bytes of the operand every workgroup re-reads). */
int nf_masked_linear_half_bwd_weight(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                     const float* grad_y, int64_t grad_y_row_stride, const uint32_t* grad_y_max,
                                     const float* mask, const int32_t* ktab, float* grad_w, float* grad_b,
                                     int64_t B, int in, int out, int relu_in, void* stream);
/* The bf16 training mode's gradient plane.  The parameter gradients of an autoregressive RQ layer have exactly two
 * readers, the final layer's backward GEMMs, which round them to bf16 / fp16: nf_rqs_bwd_max_bf16 writes them as a bf16
 * matrix [B, D*P] (8-byte aligned; everything else as nf_rqs_bwd_max, the slot gets max |gradient| before rounding),
 * nf_masked_linear_split_bwd_input_bf16 is nf_masked_linear_split_bwd_input (NF_DTYPE_BF16 pack) with that matrix as
 * its A operand (row stride in elements, out % 16 == 0), nf_masked_linear_half_bwd_weight_bf16 is
 * nf_masked_linear_half_bwd_weight with it as grad_y (wide layers with a workspace only, else NF_ERR_LAYOUT).
 * No reference counterpart beyond autograd through made.py:358 + transforms.py:819-860. */
int nf_rqs_bwd_max_bf16(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                        double pre_div, const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet,
                        float* grad_x, int64_t grad_x_row_stride, void* grad_params_bf16, uint32_t* grad_params_max,
                        int64_t B, int D, int K, int tails, double tail_bound, double min_bin_width,
                        double min_bin_height, double min_derivative, int inverse, void* stream);
int nf_masked_linear_split_bwd_input_bf16(const void* grad_y_bf16, int64_t grad_y_row_stride, const void* packed_wt,
                                          const int32_t* ktab_t, const float* bias_zero, const float* gate,
                                          int64_t gate_row_stride, float* grad_x, int64_t grad_x_row_stride,
                                          uint32_t* grad_x_max, int64_t B, int out, int in, void* stream);
int nf_masked_linear_half_bwd_weight_bf16(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                          const void* grad_y_bf16, int64_t grad_y_row_stride, const uint32_t* grad_y_max,
                                          const float* mask, const int32_t* ktab, float* grad_w, float* grad_b,
                                          int64_t B, int in, int out, int relu_in, void* workspace,
                                          size_t workspace_bytes, void* stream);
/* Backward of nf_affine_fwd (either direction; replaces autograd through transforms.py:540-566 / :323-347): from x,
 * params (scale recomputed), grad_y[B, x columns] and grad_logabsdet[B] (nullable = 0) to grad_x (written at the D
 * transformed columns only; with feat_index the caller owns the identity columns) and grad_params[B, 2D] in the
 * layout of params. */
int nf_affine_bwd(const float* x, int64_t x_row_stride, const int32_t* feat_index, const float* params,
                  int64_t param_row_stride, int layout, const float* grad_y, int64_t grad_y_row_stride,
                  const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride, float* grad_params,
                  int64_t grad_param_row_stride, int64_t B, int D, int inverse, void* stream);
/* Backward of nf_rqs_fwd(inverse=0, feat_index=NULL): from x, params (recomputed, nothing saved from the
 * forward), grad_y[B,D] and grad_logabsdet[B] to grad_x[B,D] (the spline's direct part) and
 * grad_params[B, D*P] (contiguous rows). */
int nf_rqs_bwd(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
               double pre_div, const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet,
               float* grad_x, int64_t grad_x_row_stride, float* grad_params, int64_t B, int D, int K,
               int tails, double tail_bound, double min_bin_width, double min_bin_height,
               double min_derivative, void* stream);

/* nf_rqs_bwd / nf_rqs_inverse_bwd (inverse = 0 / 1) that also leaves, in *grad_params_max, the bit pattern of the
 * largest finite |grad_params| element (the slot is zeroed on the stream first): the fp16 operand scale that the f16x2
 * GEMMs of the conditioner's backward (nf_masked_linear_split_fwd on the transposed layer, nf_masked_linear_split_bwd_weight)
 * take, so that no separate nf_maxabs pass over the [B, D*P] gradient is needed.  No counterpart in the reference
 * (autograd keeps fp32 throughout). */
int nf_rqs_bwd_max(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                   double pre_div, const float* grad_y, int64_t grad_y_row_stride,
                   const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride, float* grad_params,
                   uint32_t* grad_params_max, int64_t B, int D, int K, int tails, double tail_bound,
                   double min_bin_width, double min_bin_height, double min_derivative, int inverse,
                   void* stream);

/* Backward of nf_rqs_fwd(inverse = 1, feat_index = NULL) - the sampling direction the reference's default reverse-KL
 * loss differentiates (model.py:135-139 through :78-85): x_out is that pass's OUTPUT (nothing else is saved), and by
 * implicit differentiation of y_in = f(x_out; params) the kernel turns grad_x_out [B,D] and grad_logabsdet [B] into
 * grad_y_in [B,D] (the spline's direct part) and grad_params [B, D*P]. */
int nf_rqs_inverse_bwd(const float* x_out, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                       double pre_div, const float* grad_x_out, int64_t grad_x_out_row_stride,
                       const float* grad_logabsdet, float* grad_y_in, int64_t grad_y_in_row_stride, float* grad_params,
                       int64_t B, int D, int K, int tails, double tail_bound, double min_bin_width,
                       double min_bin_height, double min_derivative, void* stream);

/* Backward of nf_linear_spline_fwd / nf_quadratic_spline_fwd / nf_cubic_spline_fwd with feat_index = NULL (what
 * autograd computes through models/splines/linear.py:35-94, quadratic.py:48-135, cubic.py:56-221 when the reference
 * trains a flow of these families).  inverse = 0: x is the forward pass's input; inverse = 1: x is the inverse pass's
 * OUTPUT and grad_y the gradient wrt it - the inverse is differentiated implicitly through the forward map at that
 * point (for the cubic: through the quadratic the reference re-solves when |a| < 1e-3, cubic.py:191-198).
 * grad_x [B,D] is the spline's direct part, grad_params [B, D*P] has contiguous rows.  torch.clamp(outputs, 0, 1) of
 * the linear / quadratic forward passes no gradient to an output outside [0, 1], as in the reference.  LDS holds the
 * parameters and their gradients of 256 elements: P <= 79. */
int nf_linear_spline_bwd(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                         const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet,
                         float* grad_x, int64_t grad_x_row_stride, float* grad_params, int64_t B, int D,
                         int K, int tails, double tail_bound, int inverse, void* stream);
int nf_quadratic_spline_bwd(const float* x, int64_t x_row_stride, const float* params,
                            int64_t param_row_stride, const float* grad_y, int64_t grad_y_row_stride,
                            const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride,
                            float* grad_params, int64_t B, int D, int K, int tails, double tail_bound,
                            double min_bin_width, double min_bin_height, int inverse, void* stream);
int nf_cubic_spline_bwd(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                        double pre_div, const float* grad_y, int64_t grad_y_row_stride,
                        const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride,
                        float* grad_params, int64_t B, int D, int K, int tails, double bound,
                        double min_bin_width, double min_bin_height, int inverse, void* stream);

/* ---- legacy Gaussian MAF layer, float64 (replaces models/flows.py:239-263 MaskedAutoregressiveFlow on
 *      models/made.py:11-89 SimpleMADE; the reference runs this class in float64 only) ---------------- */

/* folded = mask ? mask*W : W, elementwise over [out,in] (made.py:21 `self.mask * self.weight`, once). */
int nf_fold_f64(const double* W, const double* mask, int out, int in, double* folded, void* stream);
/* y = act(x @ folded^T + bias), act = ReLU when relu_out (made.py:20-21, :40-46). */
int nf_linear_f64_fwd(const double* x, int64_t x_row_stride, const double* folded_w, const double* bias,
                      double* y, int64_t y_row_stride, int64_t B, int in, int out, int relu_out, void* stream);
/* made_out[b] = [mu (D) | logp (D)]; x = (z - mu)*exp(logp/2), features flipped when reverse_input;
 * logJ[b] = sum(logp)/2 (flows.py:242-247). */
int nf_maf_fwd(const double* z, int64_t z_row_stride, const double* made_out, int64_t out_row_stride,
               double* x, int64_t x_row_stride, double* logJ, int64_t B, int D, int reverse_input, void* stream);
/* One pass of reverse() (flows.py:255-261): z[:,dim] = mu[:,dim] + x'[:,dim]*exp(min(-logp[:,dim]/2, 10)), x' = x
 * with features flipped back when reverse_input; when logJ != NULL also logJ[b] = sum_d min(-logp_d/2, 10)
 * (the reference keeps the value of the LAST pass). */
int nf_maf_reverse_step(const double* x, int64_t x_row_stride, const double* made_out, int64_t out_row_stride,
                        double* z, int64_t z_row_stride, double* logJ, int64_t B, int D, int dim,
                        int reverse_input, void* stream);

/* Backward of the three above (the reference trains this class by autograd through flows.py:239-263 and made.py:20-21:
 * test2.py:127-135, model.py:233-244; cfg-1's "short training").
 * nf_linear_f64_bwd: g = grad_y zeroed where relu_out clipped the forward output y (y: the forward result, needed only
 * when relu_out); grad_x = g @ folded_w, grad_w = mask * (g^T @ x) (mask NULL = dense; WRITTEN, fixed-order sums, no
 * atomics), grad_b = column sums of g.  grad_x / grad_w / grad_b may each be NULL. */
int nf_linear_f64_bwd(const double* x, int64_t x_row_stride, const double* y, int64_t y_row_stride,
                      const double* folded_w, const double* mask, const double* grad_y, int64_t grad_y_row_stride,
                      double* grad_x, int64_t grad_x_row_stride, double* grad_w, double* grad_b, int64_t B, int in,
                      int out, int relu_out, void* stream);
/* Backward of nf_maf_fwd: from grad_x [B,D] and grad_logJ [B] (either may be NULL = 0) to grad_z (the direct part)
 * and grad_made_out [B,2D] = [grad_mu | grad_logp]. */
int nf_maf_bwd(const double* z, int64_t z_row_stride, const double* made_out, int64_t out_row_stride,
               const double* grad_x, int64_t grad_x_row_stride, const double* grad_logJ, double* grad_z,
               int64_t grad_z_row_stride, double* grad_made_out, int64_t grad_out_row_stride, int64_t B, int D,
               int reverse_input, void* stream);
/* Backward of one nf_maf_reverse_step pass taken out of place (z_new = z_old with column `dim` rebuilt): from
 * grad_z_new [B,D] and, for the pass that produced logJ, grad_logJ [B] (NULL otherwise) to grad_x [B,D] (zero except
 * the column the pass read), grad_made_out [B,2D] and grad_z_old [B,D] (grad_z_new with column `dim` zeroed; the part
 * through made_out comes from the conditioner's backward).  The clamp at 10 passes gradient where -logp/2 <= 10. */
int nf_maf_reverse_step_bwd(const double* x, int64_t x_row_stride, const double* made_out, int64_t out_row_stride,
                            const double* grad_z_new, int64_t grad_z_new_row_stride, const double* grad_logJ,
                            double* grad_x, int64_t grad_x_row_stride, double* grad_made_out,
                            int64_t grad_out_row_stride, double* grad_z_old, int64_t grad_z_old_row_stride, int64_t B,
                            int D, int dim, int reverse_input, void* stream);

/* ---- diagnostics (no reference counterpart): what this device sustains, printed by bench.py beside the
 *      nominal roofline peaks ------------------------------------------------------------------------- */

/* blocks x 4 waves each issue iters*4 back-to-back MFMAs on 4 accumulators: dtype NF_DTYPE_F32 =
 * v_mfma_f32_32x32x2_f32 (4096 FLOP per wave instruction), NF_DTYPE_BF16 = 32x32x16 (32768 FLOP) - both on constant
 * operands (the datapath barely switches: an upper bound); NF_DTYPE_F16X2 = v_mfma_f32_32x32x16_f16 on pseudo-random
 * operands that change from one instruction to the next (what the pipe sustains under the power cap on real data). */
int nf_diag_mfma_loop(float* scratch, int blocks, int iters, int dtype, void* stream);
/* 16-byte-per-lane device copy of `bytes` (a multiple of 16): the achievable HBM stream rate. */
int nf_diag_stream_copy(const void* src, void* dst, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NF_HIP_H */
