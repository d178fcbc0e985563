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
 * Replaces the per-call `self.weight * self.mask` of made.py:152; run once per weight update. */
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

/* ---- affine transform (replaces transforms.py:328-347 coupling, :540-566 autoregressive) -------- */

/* scale = sigmoid(u + 2) + 1e-3; forward y = scale*x + shift, inverse y = (x - shift)/scale;
 * logabsdet[b] (+)= +/- sum_d log(scale).  `layout` is NF_AFFINE_INTERLEAVED or NF_AFFINE_BLOCKED. */
int nf_affine_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index,
                  const float* params, int64_t param_row_stride, int layout,
                  float* y, int64_t y_row_stride, float* logabsdet, int accumulate,
                  int64_t B, int D, int inverse, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NF_HIP_H */
