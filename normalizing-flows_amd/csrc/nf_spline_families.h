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
template <bool INV, class PT = const float*>
__device__ __forceinline__ void lin_eval(PT p, float x, const RqsConst& a, float& y, float& ld,
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
template <bool INV, class PT = const float*>
__device__ __forceinline__ void quad_eval(PT p, float x, const RqsConst& a, float& y, float& ld,
                                          int& bin) {
    const int K = a.K;
    if (a.tails && !(x >= a.left && x <= a.right)) {
        y = x;
        ld = 0.0f;
        bin = -1;
        return;
    }
    const PT uw = p;
    const PT uh = p + K;
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

// Cubic spline (cubic.py:56-221; tails :14-54): p[0..K) unnormalised widths, p[K..2K) unnormalised heights,
// p[2K], p[2K+1] unnormalised left / right boundary derivatives.  The interval is [a.left, a.right] in both
// directions (the reference's default is [-10, 10]); a.pre_div divides the width / height logits (coupling).
// eps and the near-quadratic threshold are the reference's defaults (1e-5, 1e-3).
template <bool INV, class PT = const float*>
__device__ __forceinline__ void cubic_eval(PT p, float x, const RqsConst& a, float& y, float& ld,
                                           int& bin) {
    const int K = a.K;
    if (a.tails && !(x >= a.left && x <= a.right)) {
        y = x;
        ld = 0.0f;
        bin = -1;
        return;
    }
    const PT uw = p;
    const PT uh = p + K;
    const bool scaled = a.pre_div != 1.0f;
    auto raw = [&](PT u, int k) { return scaled ? u[k] / a.pre_div : u[k]; };
    const float t = (x - a.left) / a.span;
    float mw = -INFINITY, mh = -INFINITY;
    for (int k = 0; k < K; ++k) {
        mw = fmaxf(mw, raw(uw, k));
        mh = fmaxf(mh, raw(uh, k));
    }
    float sw = 0.0f, sh_ = 0.0f;
    for (int k = 0; k < K; ++k) {
        sw += nf_exp(raw(uw, k) - mw);
        sh_ += nf_exp(raw(uh, k) - mh);
    }
    auto width = [&](int k) { return a.min_w + a.scale_w * (nf_exp(raw(uw, k) - mw) / sw); };
    auto height = [&](int k) { return a.min_h + a.scale_h * (nf_exp(raw(uh, k) - mh) / sh_); };
    // bin search over the running sums
    double accw = 0.0, acch = 0.0;
    float cw0 = 0.0f, ch0 = 0.0f, lw = 0.0f, rw = 1.0f, d = 0.0f;
    int b = 0;
    for (int k = 0; k < K; ++k) {
        accw += (double)width(k);
        acch += (double)height(k);
        float cw1 = (float)accw, ch1 = (float)acch;
        if (k == K - 1) {                              // pinned to 1; the searched one is then raised by 1e-6
            cw1 = INV ? 1.0f : 1.0f + 1e-6f;
            ch1 = INV ? 1.0f + 1e-6f : 1.0f;
        }
        const bool ge = INV ? (t >= ch0) : (t >= cw0);
        if (ge) { b = k; lw = cw0; rw = cw1; d = ch0; }
        cw0 = cw1;
        ch0 = ch1;
    }
    // knot derivatives at the two ends of bin b
    const float wb = width(b), sb = height(b) / wb;
    auto interior = [&](float w0, float s0, float w1, float s1) {      // knot between bins (w0, s0) and (w1, s1)
        const float m1 = fminf(fabsf(s0), fabsf(s1));
        const float m2 = 0.5f * (w1 * s0 + w0 * s1) / (w0 + w1);
        const float sg0 = (s0 > 0.f) - (s0 < 0.f), sg1 = (s1 > 0.f) - (s1 < 0.f);
        return fminf(m1, m2) * (sg0 + sg1);
    };
    float dl, dr;
    if (b == 0) {
        dl = (1.0f / (1.0f + nf_exp(-p[2 * K]))) * 3.0f * sb;
    } else {
        const float w0 = width(b - 1);
        dl = interior(w0, height(b - 1) / w0, wb, sb);
    }
    if (b == K - 1) {
        dr = (1.0f / (1.0f + nf_exp(-p[2 * K + 1]))) * 3.0f * sb;
    } else {
        const float w1 = width(b + 1);
        dr = interior(wb, sb, w1, height(b + 1) / w1);
    }
    const float ca = (dl + dr - 2.0f * sb) / (wb * wb);
    const float cb = (3.0f * sb - 2.0f * dl - dr) / wb;
    const float cc = dl;
    const float lc = (float)log((double)a.span);
    if (!INV) {
        const float s = t - lw;
        const float o = ca * (s * s * s) + cb * (s * s) + cc * s + d;
        y = o * a.span + a.left;
        ld = nf_log(3.0f * ca * (s * s) + 2.0f * cb * s + cc);
        ld = (ld + lc) - lc;
    } else {
        const float b_ = (cb / ca) / 3.0f;
        const float c_ = (cc / ca) / 3.0f;
        const float d_ = (d - t) / ca;
        const float delta_1 = -(b_ * b_) + c_;
        const float delta_2 = -c_ * b_ + d_;
        const float delta_3 = b_ * d_ - c_ * c_;
        const float disc = 4.0f * delta_1 * delta_3 - delta_2 * delta_2;
        const float dep_1 = -2.0f * b_ * delta_1 + delta_2;
        float o;
        if (disc < 0.0f) {                                            // one real root (Cardano)
            const float r = sqrtf(-disc);
            const float u = (-dep_1 + r) / 2.0f, v = (-dep_1 - r) / 2.0f;
            // (expf, not nf_exp: u or v can be exactly 0, and exp(-inf) must be 0 as in helper_functions.py:514-516)
            const float cu = ((u > 0.f) - (u < 0.f)) * expf(logf(fabsf(u)) / 3.0f);
            const float cv = ((v > 0.f) - (v < 0.f)) * expf(logf(fabsf(v)) / 3.0f);
            o = (cu + cv) - b_ + lw;
        } else {                                                      // three real roots: the one inside the bin
            const float th = atan2f(sqrtf(disc), -dep_1) / 3.0f;
            const float c1 = cosf(th), s1 = sinf(th);
            const float scale = 2.0f * sqrtf(-delta_1), shift = -b_ + lw;
            const float hs3 = 0.5f * 1.7320508075688772f;
            const float r1 = c1 * scale + shift;
            const float r2 = (-0.5f * c1 - hs3 * s1) * scale + shift;
            const float r3 = (-0.5f * c1 + hs3 * s1) * scale + shift;
            const float lo = lw - 1e-5f, hi = rw + 1e-5f;
            o = (lo < r1 && r1 < hi) ? r1 : ((lo < r2 && r2 < hi) ? r2 : ((lo < r3 && r3 < hi) ? r3 : r1));
        }
        if (fabsf(ca) < 1e-3f) {                                      // almost quadratic: re-solve (:186-192)
            const float qc = d - t;
            o = (-cc + sqrtf(cc * cc - 4.0f * cb * qc)) / (2.0f * cb) + lw;
        }
        const float s = o - lw;
        y = o * a.span + a.left;
        ld = -nf_log(3.0f * ca * (s * s) + 2.0f * cb * s + cc);
        ld = (ld - lc) + lc;
    }
    bin = b;
}
