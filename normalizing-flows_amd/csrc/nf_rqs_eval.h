// Per-element rational-quadratic spline evaluation shared by the standalone kernel (nf_spline.hip)
// and the fused MADE-final-layer + spline kernel (nf_fused.hip).  Both translation units are built
// with -ffp-contract=off: every product and sum rounds once, like the reference's separate ATen ops
// (models/splines/rational_quadratic.py:54-149; search: helper_functions.py:507-512).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

struct RqsConst {
    int K, P;
    int tails;                 // 1 = linear tails (P = 3K-1), 0 = none (P = 3K+1)
    float left, right;         // same interval for heights (bottom = left, top = right)
    float span;                // (float)(right - left) computed in double on the host
    float min_w, min_h, min_d;
    float scale_w, scale_h;    // (float)(1 - min*K)
    float pre_div;             // widths/heights are divided by this first (1 = skip)
    float tail_const;          // (float)log(exp(1 - min_d) - 1)
};

static inline RqsConst nf_make_rqs_const(int K, int tails, double tail_bound, double min_w, double min_h,
                                         double min_d, double pre_div) {
    RqsConst c;
    c.K = K;
    c.P = tails ? 3 * K - 1 : 3 * K + 1;
    c.tails = tails ? 1 : 0;
    const double lo = tails ? -tail_bound : 0.0, hi = tails ? tail_bound : 1.0;
    c.left = (float)lo;
    c.right = (float)hi;
    c.span = (float)(hi - lo);
    c.min_w = (float)min_w;
    c.min_h = (float)min_h;
    c.min_d = (float)min_d;
    c.scale_w = (float)(1.0 - min_w * K);
    c.scale_h = (float)(1.0 - min_h * K);
    c.pre_div = (float)pre_div;
    c.tail_const = (float)log(exp(1.0 - min_d) - 1.0);
    return c;
}

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
__device__ __forceinline__ void rqs_eval(const float* __restrict__ p, float x, const RqsConst& a,
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

