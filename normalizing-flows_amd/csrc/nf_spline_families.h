// Per-element evaluation of the linear and quadratic spline families (reference models/splines/linear.py,
// models/splines/quadratic.py), for the same kernel skeleton as the rational-quadratic spline
// (nf_spline.hip: parameter spans staged through LDS, one element per lane, per-row logdet reduction).
// Built with -ffp-contract=off like nf_rqs_eval.h; prefix sums are carried in fp64 and rounded per prefix
// (what ATen's fp32 cumsum produces), searches are `#(x >= knot) - 1` with the last knot raised by 1e-6.
#pragma once
#include "nf_rqs_eval.h"

// torch.linspace(0, 1, K + 1)[i] in fp32: start + step * i in the lower half, end - step * (K - i) above
__device__ __forceinline__ float nf_linspace01(int i, int K) {
    const float step = 1.0f / (float)K;
    return i < (K + 1) / 2 ? step * (float)i : 1.0f - step * (float)(K - i);
}

// Linear spline: p[0..K) = unnormalised pdf of K equal bins (linear.py:35-94; tails :9-33).
template <bool INV>
__device__ __forceinline__ void lin_eval(const float* __restrict__ p, float x, const RqsConst& a, float& y, float& ld,
                                         int& bin) {
    const int K = a.K;
    if (a.tails && !(x >= a.left && x <= a.right)) {
        y = x;
        ld = 0.0f;
        bin = -1;
        return;
    }
    const float t = (x - a.left) / a.span;
    const float lc = (float)log((double)a.span);
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, p[k]);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += nf_exp(p[k] - m);
    double acc = 0.0;
    float c0 = 0.0f;                       // cdf at the left edge of bin k
    if (!INV) {
        const float pos = t * (float)K;
        int b = (int)floorf(pos);
        b = b >= K ? K - 1 : b;
        const float alpha = pos - (float)b;
        float pb = 0.0f, cb = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float pk = nf_exp(p[k] - m) / s;
            if (k == b) { pb = pk; cb = c0; }
            acc += (double)pk;
            c0 = (float)acc;
        }
        float o = cb + alpha * pb;
        o = fminf(fmaxf(o, 0.0f), 1.0f);
        y = o * a.span + a.left;
        ld = nf_log(pb) - (float)log(1.0 / (double)K);
        ld = (ld + lc) - lc;               // + log(top - bottom) - log(right - left), two roundings as in :91-92
        bin = b;
    } else {
        float lo = 0.0f, hi = 0.0f;        // cdf at the two edges of the selected bin
        int b = 0;
        for (int k = 0; k < K; ++k) {
            const float pk = nf_exp(p[k] - m) / s;
            acc += (double)pk;
            float c1 = (float)acc;
            if (k == K - 1) c1 = 1.0f + 1e-6f;      // pinned to 1, then raised in place by the search
            if (t >= c0) { b = k; lo = c0; hi = c1; }
            c0 = c1;
        }
        const float e0 = nf_linspace01(b, K), e1 = nf_linspace01(b + 1, K);
        const float slope = (hi - lo) / (e1 - e0);
        const float off = hi - slope * e1;
        float o = (t - off) / slope;
        o = fminf(fmaxf(o, 0.0f), 1.0f);
        y = o * a.span + a.left;
        ld = -nf_log(slope);
        ld = (ld - lc) + lc;
        bin = b;
    }
}

// Quadratic spline: p[0..K) unnormalised widths, p[K..P) unnormalised heights of the piecewise-linear density
// (K+1 knots; with linear tails only the K-1 interior ones, the boundary pair set so that it normalises to 1)
// (quadratic.py:48-135; tails :11-46).
template <bool INV>
__device__ __forceinline__ void quad_eval(const float* __restrict__ p, float x, const RqsConst& a, float& y, float& ld,
                                          int& bin) {
    const int K = a.K;
    if (a.tails && !(x >= a.left && x <= a.right)) {
        y = x;
        ld = 0.0f;
        bin = -1;
        return;
    }
    const float* uw = p;
    const float* uh = p + K;
    const float t = (x - a.left) / a.span;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, uw[k]);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += nf_exp(uw[k] - m);
    auto width = [&](int k) { return a.min_w + a.scale_w * (nf_exp(uw[k] - m) / s); };
    // unnormalised knot heights H_0 .. H_K
    float cst = 0.0f;
    if (a.tails) {
        const float first = 0.5f * width(0), last = 0.5f * width(K - 1);
        float mid = 0.0f;
        for (int j = 0; j + 1 < K - 1; ++j) mid += ((nf_exp(uh[j]) + nf_exp(uh[j + 1])) / 2.0f) * width(j + 1);
        const float num = 0.5f * first * nf_exp(uh[0]) + 0.5f * last * nf_exp(uh[K - 2]) + mid;
        cst = num / (1.0f - 0.5f * first - 0.5f * last);
    }
    auto H = [&](int k) {
        if (!a.tails) return nf_exp(uh[k]);
        return (k == 0 || k == K) ? cst : nf_exp(uh[k - 1]);
    };
    float area = 0.0f;
    for (int k = 0; k < K; ++k) area += ((H(k) + H(k + 1)) / 2.0f) * width(k);
    const float omh = 1.0f - a.min_h;
    double accw = 0.0, accc = 0.0;
    float loc0 = 0.0f, cdf0 = 0.0f;                  // left edge of bin k: location and cdf
    float loc = 0.0f, w = 1.0f, c = 0.0f, hl = 1.0f, hr = 1.0f;
    int b = 0;
    float hk = a.min_h + omh * (H(0) / area);
    for (int k = 0; k < K; ++k) {
        const float wk = width(k);
        const float hk1 = a.min_h + omh * (H(k + 1) / area);
        accw += (double)wk;
        accc += (double)(((hk + hk1) / 2.0f) * wk);
        float loc1 = (float)accw, cdf1 = (float)accc;
        if (k == K - 1) {                            // both pinned to 1; the searched one is then raised by 1e-6
            loc1 = INV ? 1.0f : 1.0f + 1e-6f;
            cdf1 = INV ? 1.0f + 1e-6f : 1.0f;
        }
        const bool ge = INV ? (t >= cdf0) : (t >= loc0);
        if (ge) { b = k; loc = loc0; w = wk; c = cdf0; hl = hk; hr = hk1; }
        loc0 = loc1;
        cdf0 = cdf1;
        hk = hk1;
    }
    const float qa = 0.5f * (hr - hl) * w;
    const float qb = hl * w;
    float o, alpha;
    if (INV) {
        const float c_ = c - t;
        // the reference's root (-b + sqrt(b^2 - 4 a c')) / (2a) (:120) in its cancellation-free form: for a nearly
        // flat density (a -> 0) the quotient above loses all its digits in fp32, this one none
        alpha = (2.0f * c_) / (-qb - sqrtf(qb * qb - 4.0f * qa * c_));
        o = alpha * w + loc;
    } else {
        alpha = (t - loc) / w;
        o = qa * (alpha * alpha) + qb * alpha + c;
    }
    o = fminf(fmaxf(o, 0.0f), 1.0f);
    y = o * a.span + a.left;
    const float l = nf_log(alpha * (hr - hl) + hl);
    const float lc = (float)log((double)a.span);
    ld = INV ? (-l - lc) + lc : (l + lc) - lc;      // the reference's + log(top-bottom) - log(right-left), :131-135
    bin = b;
}
