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
