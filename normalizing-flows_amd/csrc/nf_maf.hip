// Legacy Gaussian MAF layer in float64 (config 1 of BASELINE.json; reference models/flows.py:230-263
// on top of SimpleMADE, models/made.py:11-89).  The reference runs this class in float64 only
// (flows.py:240 casts the input to double), so these kernels are fp64 end to end.
//
//   linear_f64_kernel : y = act(x @ Wm^T + b), Wm = mask*W folded once on the host side of the ABI
//                       (made.py:20-21).  D and the hidden widths of this class are tiny (2..64), so this
//                       is a plain LDS-tiled fp64 VALU kernel: 64x64 outputs per block, 4x4 per thread.
//   maf_fwd_kernel    : x = (z - mu) * exp(logp/2), optional feature flip, logJ = sum(logp)/2
//                       (flows.py:242-247), one thread per sample.
//   maf_rev_kernel    : one pass of the D-pass reverse loop (flows.py:255-261): column `dim` of z is
//                       rebuilt from mu/logp of the current MADE output; the last pass also writes
//                       logJ = sum_d min(-logp_d/2, 10).
#include "nf_common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

__global__ __launch_bounds__(256) void linear_f64_kernel(const double* __restrict__ x, int64_t ldx,
                                                         const double* __restrict__ w, const double* __restrict__ b,
                                                         double* __restrict__ y, int64_t ldy, int64_t M, int K, int N,
                                                         int relu_out) {
    __shared__ double xs[TK][TM + 1];
    __shared__ double ws[TK][TN + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;          // 16 x 16 threads, 4 x 4 outputs each
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = 0; k0 < K; k0 += TK) {
        for (int i = tid; i < TM * TK; i += 256) {
            const int r = i / TK, k = i - r * TK;
            const int64_t m = m0 + r;
            xs[k][r] = (m < M && k0 + k < K) ? x[m * ldx + k0 + k] : 0.0;
            const int n = n0 + r;
            ws[k][r] = (n < N && k0 + k < K) ? w[(int64_t)n * K + k0 + k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            double a[4], c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = xs[k][ty * 4 + i]; c[i] = ws[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], c[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            double v = acc[i][j] + (b ? b[n] : 0.0);
            if (relu_out) v = v > 0.0 ? v : 0.0;
            y[m * ldy + n] = v;
        }
    }
}

__global__ __launch_bounds__(256) void fold_f64_kernel(const double* __restrict__ w, const double* __restrict__ mask,
                                                       int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = mask ? mask[i] * w[i] : w[i];
}

__global__ __launch_bounds__(256) void maf_fwd_kernel(const double* __restrict__ z, int64_t ldz,
                                                      const double* __restrict__ out, int64_t ldo,
                                                      double* __restrict__ x, int64_t ldxo, double* __restrict__ logJ,
                                                      int64_t B, int D, int flip) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        const double mu = out[r * ldo + d], logp = out[r * ldo + D + d];
        s += logp;
        x[r * ldxo + (flip ? D - 1 - d : d)] = (z[r * ldz + d] - mu) * exp(0.5 * logp);
    }
    logJ[r] = 0.5 * s;
}

__global__ __launch_bounds__(256) void maf_rev_kernel(const double* __restrict__ xin, int64_t ldx,
                                                      const double* __restrict__ out, int64_t ldo,
                                                      double* __restrict__ z, int64_t ldz, double* __restrict__ logJ,
                                                      int64_t B, int D, int dim, int flip) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    const double xv = xin[r * ldx + (flip ? D - 1 - dim : dim)];        // reverse() un-flips its input first
    const double m = fmin(-0.5 * out[r * ldo + D + dim], 10.0);
    z[r * ldz + dim] = out[r * ldo + dim] + xv * exp(m);
    if (logJ) {
        double s = 0.0;
        for (int d = 0; d < D; ++d) s += fmin(-0.5 * out[r * ldo + D + d], 10.0);
        logJ[r] = s;
    }
}

// ---- backward (cfg-1 "short training": the reference differentiates these op sequences with autograd,
//      test2.py:127-135 / model.py:233-244) -------------------------------------------------------------------
// grad_x[b,k] = sum_n g[b,n] * w[n,k], g = grad_y zeroed where the forward output was clipped by the ReLU
__global__ __launch_bounds__(256) void linear_f64_bwd_input_kernel(const double* __restrict__ gy, int64_t ldgy,
                                                                   const double* __restrict__ y, int64_t ldy,
                                                                   const double* __restrict__ w,
                                                                   double* __restrict__ gx, int64_t ldgx, int64_t M,
                                                                   int K, int N, int relu_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * K) return;
    const int64_t b = i / K;
    const int k = (int)(i - b * K);
    double s = 0.0;
    for (int n = 0; n < N; ++n) {
        double g = gy[b * ldgy + n];
        if (relu_out && !(y[b * ldy + n] > 0.0)) g = 0.0;
        s = fma(g, w[(int64_t)n * K + k], s);
    }
    gx[b * ldgx + k] = s;
}

// blocks 0 .. N*K-1: grad_w[n,k] = mask[n,k] * sum_b g[b,n] * x[b,k]; blocks N*K .. N*K+N-1: grad_b[n] = sum_b g[b,n].
// One block per output element, fixed-order tree reduction over the batch (no atomics: results are reproducible).
__global__ __launch_bounds__(256) void linear_f64_bwd_weight_kernel(const double* __restrict__ x, int64_t ldx,
                                                                    const double* __restrict__ gy, int64_t ldgy,
                                                                    const double* __restrict__ y, int64_t ldy,
                                                                    const double* __restrict__ mask,
                                                                    double* __restrict__ gw, double* __restrict__ gb,
                                                                    int64_t M, int K, int N, int relu_out) {
    __shared__ double part[256];
    const int e = blockIdx.x;
    const bool is_bias = e >= N * K;
    const int n = is_bias ? e - N * K : e / K;
    const int k = is_bias ? 0 : e - n * K;
    if (is_bias ? gb == nullptr : gw == nullptr) return;
    double s = 0.0;
    if (is_bias || !mask || mask[(int64_t)n * K + k] != 0.0) {
        for (int64_t b = threadIdx.x; b < M; b += 256) {
            double g = gy[b * ldgy + n];
            if (relu_out && !(y[b * ldy + n] > 0.0)) g = 0.0;
            s += is_bias ? g : g * x[b * ldx + k];
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (is_bias) gb[n] = part[0];
        else gw[(int64_t)n * K + k] = mask ? mask[(int64_t)n * K + k] * part[0] : part[0];
    }
}

// x_pre = (z - mu) e, e = exp(logp/2), x = flip?(x_pre), logJ = sum(logp)/2:
//   grad_z = g e, grad_mu = -g e, grad_logp = g (z - mu) e / 2 + grad_logJ / 2, g = grad_x un-flipped
__global__ __launch_bounds__(256) void maf_bwd_kernel(const double* __restrict__ z, int64_t ldz,
                                                      const double* __restrict__ out, int64_t ldo,
                                                      const double* __restrict__ gx, int64_t ldgx,
                                                      const double* __restrict__ glj, double* __restrict__ gz,
                                                      int64_t ldgz, double* __restrict__ gout, int64_t ldgo, int64_t B,
                                                      int D, int flip) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    const double gl = glj ? glj[r] : 0.0;
    for (int d = 0; d < D; ++d) {
        const double mu = out[r * ldo + d], logp = out[r * ldo + D + d];
        const double e = exp(0.5 * logp);
        const double g = gx ? gx[r * ldgx + (flip ? D - 1 - d : d)] : 0.0;
        gz[r * ldgz + d] = g * e;
        gout[r * ldgo + d] = -g * e;
        gout[r * ldgo + D + d] = 0.5 * g * (z[r * ldz + d] - mu) * e + 0.5 * gl;
    }
}

// one pass of reverse(): z_new = z_old except z_new[dim] = mu[dim] + x'[dim] exp(m[dim]), m = min(-logp/2, 10);
// the last pass also gives logJ = sum_d m[d].  G = grad wrt z_new:
//   grad_x'[dim] = G[dim] exp(m[dim]); grad_mu[dim] = G[dim];
//   grad_logp[d] = -1/2 [ -logp[d]/2 <= 10 ] ( [d == dim] G[dim] x'[dim] exp(m[dim]) + grad_logJ );  grad_z_old = G, column dim zeroed
__global__ __launch_bounds__(256) void maf_rev_bwd_kernel(const double* __restrict__ xin, int64_t ldx,
                                                          const double* __restrict__ out, int64_t ldo,
                                                          const double* __restrict__ gzn, int64_t ldgzn,
                                                          const double* __restrict__ glj, double* __restrict__ gx,
                                                          int64_t ldgx, double* __restrict__ gout, int64_t ldgo,
                                                          double* __restrict__ gzo, int64_t ldgzo, int64_t B, int D,
                                                          int dim, int flip) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= B) return;
    const int xc = flip ? D - 1 - dim : dim;
    const double gl = glj ? glj[r] : 0.0;
    const double G = gzn ? gzn[r * ldgzn + dim] : 0.0;
    for (int d = 0; d < D; ++d) {
        const double h = -0.5 * out[r * ldo + D + d];
        const double open = h <= 10.0 ? 1.0 : 0.0;                  // torch.clamp(max=10) passes the gradient at equality
        double glp = gl;
        if (d == dim) {
            const double em = exp(fmin(h, 10.0));
            glp += G * xin[r * ldx + xc] * em;
            gx[r * ldgx + xc] = G * em;
        }
        gout[r * ldgo + d] = d == dim ? G : 0.0;
        gout[r * ldgo + D + d] = -0.5 * open * glp;
        gzo[r * ldgzo + d] = d == dim ? 0.0 : (gzn ? gzn[r * ldgzn + d] : 0.0);
        if (d != dim) gx[r * ldgx + (flip ? D - 1 - d : d)] = 0.0;
    }
}

}  // namespace

extern "C" int nf_fold_f64(const double* W, const double* mask, int out, int in, double* folded, void* stream) {
    if (out <= 0 || in <= 0) return NF_ERR_SHAPE;
    if (!W || !folded) return NF_ERR_NULL;
    const int64_t n = (int64_t)out * in;
    hipLaunchKernelGGL(fold_f64_kernel, dim3(nf_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, W, mask, n, folded);
    return nf_launch_status();
}

extern "C" int nf_linear_f64_fwd(const double* x, int64_t x_row_stride, const double* folded_w, const double* bias,
                                 double* y, int64_t y_row_stride, int64_t B, int in, int out, int relu_out,
                                 void* stream) {
    if (B < 0 || in <= 0 || out <= 0) return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!x || !folded_w || !y) return NF_ERR_NULL;
    if (x_row_stride < in || y_row_stride < out) return NF_ERR_LAYOUT;
    const int64_t gx = (B + TM - 1) / TM;
    if (gx > 0x7fffffff) return NF_ERR_SHAPE;
    hipLaunchKernelGGL(linear_f64_kernel, dim3((int)gx, (out + TN - 1) / TN), dim3(256), 0, (hipStream_t)stream, x,
                       x_row_stride, folded_w, bias, y, y_row_stride, B, in, out, relu_out);
    return nf_launch_status();
}

extern "C" int nf_maf_fwd(const double* z, int64_t z_row_stride, const double* made_out, int64_t out_row_stride,
                          double* x, int64_t x_row_stride, double* logJ, int64_t B, int D, int reverse_input,
                          void* stream) {
    if (B < 0 || D <= 0) return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!z || !made_out || !x || !logJ) return NF_ERR_NULL;
    if (z_row_stride < D || x_row_stride < D || out_row_stride < 2 * (int64_t)D) return NF_ERR_LAYOUT;
    hipLaunchKernelGGL(maf_fwd_kernel, dim3(nf_ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, z, z_row_stride,
                       made_out, out_row_stride, x, x_row_stride, logJ, B, D, reverse_input);
    return nf_launch_status();
}

extern "C" int nf_maf_reverse_step(const double* x, int64_t x_row_stride, const double* made_out,
                                   int64_t out_row_stride, double* z, int64_t z_row_stride, double* logJ, int64_t B,
                                   int D, int dim, int reverse_input, void* stream) {
    if (B < 0 || D <= 0 || dim < 0 || dim >= D) return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!x || !made_out || !z) return NF_ERR_NULL;
    if (z_row_stride < D || x_row_stride < D || out_row_stride < 2 * (int64_t)D) return NF_ERR_LAYOUT;
    hipLaunchKernelGGL(maf_rev_kernel, dim3(nf_ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, x, x_row_stride,
                       made_out, out_row_stride, z, z_row_stride, logJ, B, D, dim, reverse_input);
    return nf_launch_status();
}

extern "C" int nf_linear_f64_bwd(const double* x, int64_t x_row_stride, const double* y, int64_t y_row_stride,
                                 const double* folded_w, const double* mask, const double* grad_y,
                                 int64_t grad_y_row_stride, double* grad_x, int64_t grad_x_row_stride, double* grad_w,
                                 double* grad_b, int64_t B, int in, int out, int relu_out, void* stream) {
    if (B < 0 || in <= 0 || out <= 0) return NF_ERR_SHAPE;
    if (!grad_y && B > 0) return NF_ERR_NULL;
    if ((grad_w || grad_b) && B > 0 && !x) return NF_ERR_NULL;
    if (grad_x && B > 0 && !folded_w) return NF_ERR_NULL;
    if (relu_out && B > 0 && !y) return NF_ERR_NULL;
    if (B > 0 && (grad_y_row_stride < out || (x && x_row_stride < in) || (relu_out && y_row_stride < out) ||
                  (grad_x && grad_x_row_stride < in)))
        return NF_ERR_LAYOUT;
    hipStream_t s = (hipStream_t)stream;
    if (grad_x && B > 0) {
        const int64_t blocks = (B * in + 255) / 256;
        if (blocks > 0x7fffffff) return NF_ERR_SHAPE;
        hipLaunchKernelGGL(linear_f64_bwd_input_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grad_y,
                           grad_y_row_stride, y, y_row_stride, folded_w, grad_x, grad_x_row_stride, B, in, out, relu_out);
    }
    if (grad_w || grad_b)           // (B == 0: the sums are empty, the kernel writes zeros)
        hipLaunchKernelGGL(linear_f64_bwd_weight_kernel, dim3(out * in + out), dim3(256), 0, s, x, x_row_stride, grad_y,
                           grad_y_row_stride, y, y_row_stride, mask, grad_w, grad_b, B, in, out, relu_out);
    return nf_launch_status();
}

extern "C" int nf_maf_bwd(const double* z, int64_t z_row_stride, const double* made_out, int64_t out_row_stride,
                          const double* grad_x, int64_t grad_x_row_stride, const double* grad_logJ, double* grad_z,
                          int64_t grad_z_row_stride, double* grad_made_out, int64_t grad_out_row_stride, int64_t B, int D,
                          int reverse_input, void* stream) {
    if (B < 0 || D <= 0) return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!z || !made_out || !grad_z || !grad_made_out) return NF_ERR_NULL;
    if (z_row_stride < D || out_row_stride < 2 * (int64_t)D || grad_z_row_stride < D ||
        grad_out_row_stride < 2 * (int64_t)D || (grad_x && grad_x_row_stride < D))
        return NF_ERR_LAYOUT;
    hipLaunchKernelGGL(maf_bwd_kernel, dim3(nf_ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, z, z_row_stride,
                       made_out, out_row_stride, grad_x, grad_x_row_stride, grad_logJ, grad_z, grad_z_row_stride,
                       grad_made_out, grad_out_row_stride, B, D, reverse_input);
    return nf_launch_status();
}

extern "C" int nf_maf_reverse_step_bwd(const double* x, int64_t x_row_stride, const double* made_out,
                                       int64_t out_row_stride, const double* grad_z_new, int64_t grad_z_new_row_stride,
                                       const double* grad_logJ, double* grad_x, int64_t grad_x_row_stride,
                                       double* grad_made_out, int64_t grad_out_row_stride, double* grad_z_old,
                                       int64_t grad_z_old_row_stride, int64_t B, int D, int dim, int reverse_input,
                                       void* stream) {
    if (B < 0 || D <= 0 || dim < 0 || dim >= D) return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!x || !made_out || !grad_x || !grad_made_out || !grad_z_old) return NF_ERR_NULL;
    if (x_row_stride < D || out_row_stride < 2 * (int64_t)D || grad_x_row_stride < D ||
        grad_out_row_stride < 2 * (int64_t)D || grad_z_old_row_stride < D || (grad_z_new && grad_z_new_row_stride < D))
        return NF_ERR_LAYOUT;
    hipLaunchKernelGGL(maf_rev_bwd_kernel, dim3(nf_ceil_div(B, 256)), dim3(256), 0, (hipStream_t)stream, x, x_row_stride,
                       made_out, out_row_stride, grad_z_new, grad_z_new_row_stride, grad_logJ, grad_x, grad_x_row_stride,
                       grad_made_out, grad_out_row_stride, grad_z_old, grad_z_old_row_stride, B, D, dim, reverse_input);
    return nf_launch_status();
}
