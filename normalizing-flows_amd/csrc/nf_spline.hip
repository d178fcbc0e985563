// Fused elementwise transforms of the flow hot path for gfx950 (MI355X):
//   rqs_kernel    - rational-quadratic spline forward / inverse with linear tails or on [0,1]
//                   (reference: models/splines/rational_quadratic.py:13-149,
//                    helper_functions.py:507-512, :393-398)
//   affine_kernel - sigmoid-scale affine forward / inverse
//                   (reference: models/transforms.py:328-347, :540-566)
//
// One pass over HBM: the (3K-1)-float parameter rows of 256 consecutive (sample, feature)
// elements are one contiguous span; the block copies it with 16-byte coalesced loads into LDS
// (next span prefetched into registers while the current one is evaluated), each lane then walks
// its own P floats from LDS (odd stride => conflict free), builds the knots on the fly, searches,
// evaluates, and the per-sample log|det J| is reduced by a wavefront shuffle in a fixed order
// (deterministic: no float atomics).
//
// This translation unit is compiled with -ffp-contract=off: every product and sum below rounds
// once, like the separate ATen ops of the reference, so that knots (and hence bin indices) track
// the reference as closely as a different expf allows.
#include "nf_common.h"
#include <math.h>

namespace {

constexpr int kThreads = 256;

struct RqsArgs {
    const float* x;
    int64_t x_stride;
    const int32_t* feat;       // nullable: column of element d in x / y rows (coupling)
    const float* params;       // [B, D*P] contiguous
    int64_t n_param_floats;    // B*D*P
    float* y;
    int64_t y_stride;
    float* lad;                // [B]
    int accumulate;
    int32_t* bins;             // nullable [B, D]
    int64_t B;
    int D, K, P;
    int tails;                 // 1 = linear tails (P = 3K-1), 0 = none (P = 3K+1)
    float left, right;         // same interval for heights (bottom = left, top = right)
    float span;                // (float)(right - left) computed in double on the host
    float min_w, min_h, min_d;
    float scale_w, scale_h;    // (float)(1 - min*K)
    float pre_div;             // widths/heights are divided by this first (1 = skip)
    float tail_const;          // (float)log(exp(1 - min_d) - 1)
    int rows_per_block;
    int vec_ok;                // params base 16-byte aligned
};

__device__ __forceinline__ float softplus_f(float u) {
    // F.softplus(beta=1, threshold=20)
    return u > 20.0f ? u : log1pf(expf(u));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Evaluates one element.  p points at its P parameters in LDS.
template <int KT, bool INV>
__device__ __forceinline__ void rqs_eval(const float* __restrict__ p, float x, const RqsArgs& a,
                                         float& y, float& ld, int& bin) {
    const int K = KT > 0 ? KT : a.K;
    if (a.tails && !(x >= a.left && x <= a.right)) {   // linear tails: identity, logdet 0 (NaN lands here too)
        y = x;
        ld = 0.0f;
        bin = -1;
        return;
    }
    const float* uw = p;
    const float* uh = p + K;
    const float* ud = p + 2 * K;
    const bool scaled = a.pre_div != 1.0f;

    // softmax denominators: max, then sum of exp(u - max) left to right
    auto raw = [&](const float* u, int k) {
        float v = u[k];
        return scaled ? v / a.pre_div : v;
    };
    float mw = -INFINITY, mh = -INFINITY;
    constexpr int KA = KT > 0 ? KT : 1;
    float ew[KA], eh[KA];              // exp(u - max) kept in registers when K is a template constant
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float w = raw(uw, k), h = raw(uh, k);
        if constexpr (KT > 0) { ew[k] = w; eh[k] = h; }
        mw = fmaxf(mw, w);
        mh = fmaxf(mh, h);
    }
    float sw = 0.0f, sh = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float w, h;
        if constexpr (KT > 0) { w = ew[k]; h = eh[k]; }
        else { w = raw(uw, k); h = raw(uh, k); }
        w = expf(w - mw);
        h = expf(h - mh);
        if constexpr (KT > 0) { ew[k] = w; eh[k] = h; }
        sw += w;
        sh += h;
    }
    const float rw = 1.0f / sw, rh = 1.0f / sh;

    // knots on the fly: floor + rescale, prefix sum carried in fp64 and rounded per prefix (what
    // ATen's fp32 cumsum produces), affine map to [left,right], last knot pinned.
    double accw = 0.0, acch = 0.0;
    float kw0 = a.left, kh0 = a.left;          // left knot of the current bin
    float x0 = a.left, x1 = a.left, y0 = a.left, y1 = a.left;
    bin = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float w, h;
        if constexpr (KT > 0) { w = ew[k]; h = eh[k]; }
        else { w = expf(raw(uw, k) - mw); h = expf(raw(uh, k) - mh); }
        w = a.min_w + a.scale_w * (w * rw);
        h = a.min_h + a.scale_h * (h * rh);
        accw += (double)w;
        acch += (double)h;
        float kw1 = a.span * (float)accw + a.left;
        float kh1 = a.span * (float)acch + a.left;
        if (k == K - 1) { kw1 = a.right; kh1 = a.right; }
        // reference: bin = #(x >= knot) - 1 with the last knot raised by 1e-6; knots increase, so the
        // last k whose left knot is <= x is that count (clamped to K-1 where the reference would index
        // past the end).
        const bool ge = INV ? (x >= kh0) : (x >= kw0);
        if (ge) { bin = k; x0 = kw0; x1 = kw1; y0 = kh0; y1 = kh1; }
        kw0 = kw1;
        kh0 = kh1;
    }

    // slopes at the two knots of the bin
    float u0, u1;
    if (a.tails) {
        u0 = bin == 0 ? a.tail_const : ud[bin - 1];
        u1 = bin == K - 1 ? a.tail_const : ud[bin];
    } else {
        u0 = ud[bin];
        u1 = ud[bin + 1];
    }
    const float d0 = a.min_d + softplus_f(u0);
    const float d1 = a.min_d + softplus_f(u1);

    const float bw = x1 - x0;
    const float bh = y1 - y0;
    const float s = bh / bw;
    const float curv = d0 + d1 - 2.0f * s;
    float t, den;
    if (INV) {
        const float dy = x - y0;
        const float qa = dy * curv + bh * (s - d0);
        const float qb = bh * d0 - dy * curv;
        const float qc = -s * dy;
        const float disc = qb * qb - 4.0f * qa * qc;
        t = (2.0f * qc) / (-qb - sqrtf(disc));       // disc < 0 yields NaN (reference asserts)
        y = t * bw + x0;
    } else {
        t = (x - x0) / bw;
    }
    const float t1mt = t * (1.0f - t);
    den = s + curv * t1mt;
    if (!INV) {
        const float num = bh * (s * (t * t) + d0 * t1mt);
        y = y0 + num / den;
    }
    const float omt = 1.0f - t;
    const float dnum = (s * s) * (d1 * (t * t) + 2.0f * s * t1mt + d0 * (omt * omt));
    const float l = logf(dnum) - 2.0f * logf(den);
    ld = INV ? -l : l;
}

// Copies params floats [a0, a0 + 4*nvec) into registers (16-byte loads where the whole vector is in range).
__device__ __forceinline__ float4 load_vec(const float* __restrict__ params, int64_t f, int64_t total, int vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < 0) return v;
    if (vec_ok && f + 3 < total) {
        v = *reinterpret_cast<const float4*>(params + f);
    } else {
        if (f < total) v.x = params[f];
        if (f + 1 < total) v.y = params[f + 1];
        if (f + 2 < total) v.z = params[f + 2];
        if (f + 3 < total) v.w = params[f + 3];
    }
    return v;
}

template <int KT, bool INV, int MAXV>
__global__ __launch_bounds__(kThreads) void rqs_kernel(const RqsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int P = a.P;
    const int D = a.D;
    const int64_t row0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int nrows = (int)min((int64_t)a.rows_per_block, a.B - row0);
    const int E = nrows * D;
    const int stage_floats = (kThreads * P + 4 + 3) & ~3;
    float* stage = smem;
    float* ldbuf = smem + stage_floats;

    float4 pre[MAXV];
    // span of iteration `it0`: floats [g0, g0 + n*P) of params; staged from the 4-float boundary below g0
    auto issue = [&](int it0) {
        const int n = min(kThreads, E - it0);
        const int64_t g0 = (row0 * D + it0) * (int64_t)P;
        const int64_t a0 = g0 & ~(int64_t)3;
        const int nvec = (int)((g0 - a0 + (int64_t)n * P + 3) >> 2);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = tid + i * kThreads;
            pre[i] = v < nvec ? load_vec(a.params, a0 + 4 * (int64_t)v, a.n_param_floats, a.vec_ok)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int it0) {
        const int n = min(kThreads, E - it0);
        const int64_t g0 = (row0 * D + it0) * (int64_t)P;
        const int64_t a0 = g0 & ~(int64_t)3;
        const int nvec = (int)((g0 - a0 + (int64_t)n * P + 3) >> 2);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = tid + i * kThreads;
            if (v < nvec) *reinterpret_cast<float4*>(stage + 4 * v) = pre[i];
        }
    };

    if (E > 0) issue(0);
    for (int it0 = 0; it0 < E; it0 += kThreads) {
        __syncthreads();               // previous span fully consumed
        commit(it0);
        __syncthreads();
        if (it0 + kThreads < E) issue(it0 + kThreads);   // in flight while this span is evaluated
        const int n = min(kThreads, E - it0);
        if (tid < n) {
            const int e = it0 + tid;
            const int r = e / D;
            const int d = e - r * D;
            const int col = a.feat ? a.feat[d] : d;
            const int shift = (int)(((row0 * D + it0) * (int64_t)P) & 3);
            const float xv = a.x[(row0 + r) * a.x_stride + col];
            float yv, ld;
            int bin;
            rqs_eval<KT, INV>(stage + shift + tid * P, xv, a, yv, ld, bin);
            a.y[(row0 + r) * a.y_stride + col] = yv;
            if (a.bins) a.bins[(row0 + r) * (int64_t)D + d] = bin;
            ldbuf[e] = ld;
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < nrows; r += kThreads / 64) {
        float s = 0.0f;
        for (int i = lane; i < D; i += 64) s += ldbuf[r * D + i];
        s = wave_sum(s);
        if (lane == 0) {
            float* o = a.lad + row0 + r;
            *o = a.accumulate ? *o + s : s;
        }
    }
}

struct AffineArgs {
    const float* x;
    int64_t x_stride;
    const int32_t* feat;
    const float* params;
    int64_t p_stride;
    float* y;
    int64_t y_stride;
    float* lad;
    int accumulate;
    int64_t B;
    int D;
    int layout;
    int rows_per_block;
};

template <bool INV>
__global__ __launch_bounds__(kThreads) void affine_kernel(const AffineArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int D = a.D;
    const int64_t row0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int nrows = (int)min((int64_t)a.rows_per_block, a.B - row0);
    const int E = nrows * D;
    for (int e = tid; e < E; e += kThreads) {
        const int r = e / D;
        const int d = e - r * D;
        const int col = a.feat ? a.feat[d] : d;
        const float* prow = a.params + (row0 + r) * a.p_stride;
        float us, shift;
        if (a.layout == NF_AFFINE_INTERLEAVED) {
            const float2 v = *reinterpret_cast<const float2*>(prow + 2 * d);
            us = v.x;
            shift = v.y;
        } else {
            shift = prow[d];
            us = prow[D + d];
        }
        const float z = us + 2.0f;
        const float scale = 1.0f / (1.0f + expf(-z)) + 1e-3f;       // sigmoid(u + 2) + 1e-3
        const float ls = logf(scale);
        const float xv = a.x[(row0 + r) * a.x_stride + col];
        a.y[(row0 + r) * a.y_stride + col] = INV ? (xv - shift) / scale : scale * xv + shift;
        smem[e] = INV ? -ls : ls;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < nrows; r += kThreads / 64) {
        float s = 0.0f;
        for (int i = lane; i < D; i += 64) s += smem[r * D + i];
        s = wave_sum(s);
        if (lane == 0) {
            float* o = a.lad + row0 + r;
            *o = a.accumulate ? *o + s : s;
        }
    }
}

int pick_rows_per_block(int D, int64_t B) {
    // ~3k elements per block, whole rows, at most what LDS holds for the logdet buffer
    int r = 3072 / D;
    if (r < 1) r = 1;
    if ((int64_t)r > B) r = (int)B;
    return r;
}

template <int KT, bool INV>
int launch_rqs_k(const RqsArgs& a, int grid, size_t lds, hipStream_t s) {
    const int nv = (kThreads * a.P + 4 + 3) / 4;
    const int per = (nv + kThreads - 1) / kThreads;
    if (per <= 6) hipLaunchKernelGGL((rqs_kernel<KT, INV, 6>), dim3(grid), dim3(kThreads), lds, s, a);
    else if (per <= 12) hipLaunchKernelGGL((rqs_kernel<KT, INV, 12>), dim3(grid), dim3(kThreads), lds, s, a);
    else if (per <= 25) hipLaunchKernelGGL((rqs_kernel<KT, INV, 25>), dim3(grid), dim3(kThreads), lds, s, a);
    else return NF_ERR_SHAPE;
    return nf_launch_status();
}

template <bool INV>
int launch_rqs(const RqsArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (a.K) {
        case 4: return launch_rqs_k<4, INV>(a, grid, lds, s);
        case 8: return launch_rqs_k<8, INV>(a, grid, lds, s);
        case 10: return launch_rqs_k<10, INV>(a, grid, lds, s);
        case 16: return launch_rqs_k<16, INV>(a, grid, lds, s);
        default: return launch_rqs_k<0, INV>(a, grid, lds, s);
    }
}

}  // namespace

extern "C" int nf_rqs_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index,
                          const float* params, int64_t param_row_stride, double pre_div,
                          float* y, int64_t y_row_stride, float* logabsdet, int accumulate,
                          int32_t* bin_idx, int64_t B, int D, int K, int tails, double tail_bound,
                          double min_bin_width, double min_bin_height, double min_derivative,
                          int inverse, void* stream) {
    if (B < 0 || D <= 0 || K <= 0 || K > 32) return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    if (B == 0) return NF_OK;          // empty batch: nothing to read, pointers may be NULL
    if (!x || !params || !y || !logabsdet) return NF_ERR_NULL;
    const int P = tails ? 3 * K - 1 : 3 * K + 1;
    if (param_row_stride != (int64_t)D * P) return NF_ERR_LAYOUT;
    if (x_row_stride < D || y_row_stride < D) return NF_ERR_LAYOUT;

    RqsArgs a;
    a.x = x; a.x_stride = x_row_stride; a.feat = feat_index;
    a.params = params; a.n_param_floats = B * (int64_t)D * P;
    a.y = y; a.y_stride = y_row_stride; a.lad = logabsdet; a.accumulate = accumulate; a.bins = bin_idx;
    a.B = B; a.D = D; a.K = K; a.P = P; a.tails = tails ? 1 : 0;
    const double lo = tails ? -tail_bound : 0.0, hi = tails ? tail_bound : 1.0;
    a.left = (float)lo; a.right = (float)hi; a.span = (float)(hi - lo);
    a.min_w = (float)min_bin_width; a.min_h = (float)min_bin_height; a.min_d = (float)min_derivative;
    a.scale_w = (float)(1.0 - min_bin_width * K);
    a.scale_h = (float)(1.0 - min_bin_height * K);
    a.pre_div = (float)pre_div;
    a.tail_const = (float)log(exp(1.0 - min_derivative) - 1.0);
    a.rows_per_block = pick_rows_per_block(D, B);
    a.vec_ok = ((uintptr_t)params & 15) == 0;
    const int stage_floats = (kThreads * P + 4 + 3) & ~3;
    size_t lds = sizeof(float) * ((size_t)stage_floats + (size_t)a.rows_per_block * D);
    if (lds > 160 * 1024) return NF_ERR_SHAPE;
    const int64_t grid = (B + a.rows_per_block - 1) / a.rows_per_block;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    return inverse ? launch_rqs<true>(a, (int)grid, lds, s) : launch_rqs<false>(a, (int)grid, lds, s);
}

extern "C" int nf_affine_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index,
                             const float* params, int64_t param_row_stride, int layout,
                             float* y, int64_t y_row_stride, float* logabsdet, int accumulate,
                             int64_t B, int D, int inverse, void* stream) {
    if (B < 0 || D <= 0) return NF_ERR_SHAPE;
    if (layout != NF_AFFINE_INTERLEAVED && layout != NF_AFFINE_BLOCKED) return NF_ERR_DTYPE;
    if (B == 0) return NF_OK;
    if (!x || !params || !y || !logabsdet) return NF_ERR_NULL;
    if (param_row_stride < 2 * (int64_t)D || x_row_stride < D || y_row_stride < D) return NF_ERR_LAYOUT;
    if (layout == NF_AFFINE_INTERLEAVED && ((((uintptr_t)params) & 7) || (param_row_stride & 1))) return NF_ERR_LAYOUT;
    if (B == 0) return NF_OK;
    AffineArgs a;
    a.x = x; a.x_stride = x_row_stride; a.feat = feat_index; a.params = params; a.p_stride = param_row_stride;
    a.y = y; a.y_stride = y_row_stride; a.lad = logabsdet; a.accumulate = accumulate; a.B = B; a.D = D;
    a.layout = layout;
    a.rows_per_block = pick_rows_per_block(D, B);
    size_t lds = sizeof(float) * (size_t)a.rows_per_block * D;
    if (lds > 160 * 1024) return NF_ERR_SHAPE;
    const int64_t grid = (B + a.rows_per_block - 1) / a.rows_per_block;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (inverse) hipLaunchKernelGGL(affine_kernel<true>, dim3((int)grid), dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL(affine_kernel<false>, dim3((int)grid), dim3(kThreads), lds, s, a);
    return nf_launch_status();
}
