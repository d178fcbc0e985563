// Per-element evaluation of the linear and quadratic spline families (reference models/splines/linear.py,
// models/splines/quadratic.py), for the same kernel skeleton as the rational-quadratic spline
// (nf_spline.hip: parameter spans staged through LDS, one element per lane, per-row logdet reduction).
// Built with -ffp-contract=off like nf_rqs_eval.h; prefix sums are carried in fp64 and rounded per prefix
// (what ATen's fp32 cumsum produces), searches are `#(x >= knot) - 1` with the last knot raised by 1e-6.
#pragma once
#include "nf_rqs_eval.h"

// KT > 0: the bin count as a template constant.
// CACHE (forward evaluators): p is this lane's own writable copy of the parameters (the standalone kernel's LDS
// stage), so exp(u - max) / the widths / exp(heights) are written over the logits once and re-read instead of being
// recomputed by every loop below - same operations on the same values, so the same bits, at a third of the
// transcendentals.  The backward evaluators always work this way.
// torch.linspace(0, 1, K + 1)[i] in fp32: start + step * i in the lower half, end - step * (K - i) above
__device__ __forceinline__ float nf_linspace01(int i, int K) {
    const float step = 1.0f / (float)K;
    return i < (K + 1) / 2 ? step * (float)i : 1.0f - step * (float)(K - i);
}

// Linear spline: p[0..K) = unnormalised pdf of K equal bins (linear.py:35-94; tails :9-33).
template <bool INV, class PT = const float*, bool CACHE = false, int KT = 0>
__device__ __forceinline__ void lin_eval(PT p, float x, const RqsConst& a, float& y, float& ld,
                                         int& bin) {
    const int K = KT > 0 ? KT : a.K;          // KT: compile-time bin count (loops unroll, values stay in registers)
    if (a.tails && !(x >= a.left && x <= a.right)) {
        y = x;
        ld = 0.0f;
        bin = -1;
        return;
    }
    const float t = (x - a.left) / a.span;
    const float lc = (float)log((double)a.span);
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) m = fmaxf(m, p[k]);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float e = nf_exp(p[k] - m);
        if constexpr (CACHE) p[k] = e;
        s += e;
    }
    auto ex = [&](int k) {
        if constexpr (CACHE) return p[k];
        else return nf_exp(p[k] - m);
    };
    const NfDivisor ds = nf_divisor(s);
    double acc = 0.0;
    float c0 = 0.0f;                       // cdf at the left edge of bin k
    if (!INV) {
        const float pos = t * (float)K;
        int b = (int)floorf(pos);
        b = b >= K ? K - 1 : b;
        const float alpha = pos - (float)b;
        float pb = 0.0f, cb = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float pk = nf_div_ieee(ex(k), ds);
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
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float pk = nf_div_ieee(ex(k), ds);
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
template <bool INV, class PT = const float*, bool CACHE = false, int KT = 0>
__device__ __forceinline__ void quad_eval(PT p, float x, const RqsConst& a, float& y, float& ld,
                                          int& bin) {
    const int K = KT > 0 ? KT : a.K;          // KT: compile-time bin count (loops unroll, values stay in registers)
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
#pragma unroll
    for (int k = 0; k < K; ++k) m = fmaxf(m, uw[k]);
    // KT > 0: widths and exp(heights) live in registers (every index below is static once the loops unroll);
    // otherwise CACHE keeps them over the logits in LDS
    constexpr int KA = KT > 0 ? KT : 1;
    float wv[KA], ev[KA + 1];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float e = nf_exp(uw[k] - m);
        if constexpr (KT > 0) wv[k] = e;
        else if constexpr (CACHE) uw[k] = e;
        s += e;
    }
    const NfDivisor ds = nf_divisor(s);
    const int nh = a.tails ? K - 1 : K + 1;
    if constexpr (KT > 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) wv[k] = a.min_w + a.scale_w * nf_div_ieee(wv[k], ds);
#pragma unroll
        for (int j = 0; j < K + 1; ++j) ev[j] = j < K - 1 || j < nh ? nf_exp(uh[j]) : 0.0f;
    } else if constexpr (CACHE) {
#pragma unroll
        for (int k = 0; k < K; ++k) uw[k] = a.min_w + a.scale_w * nf_div_ieee(uw[k], ds);
#pragma unroll
        for (int j = 0; j < nh; ++j) uh[j] = nf_exp(uh[j]);
    }
    auto width = [&](int k) {
        if constexpr (KT > 0) return wv[k];
        else if constexpr (CACHE) return uw[k];
        else return a.min_w + a.scale_w * nf_div_ieee(nf_exp(uw[k] - m), ds);
    };
    auto eh = [&](int j) {
        if constexpr (KT > 0) return ev[j];
        else if constexpr (CACHE) return uh[j];
        else return nf_exp(uh[j]);
    };
    // unnormalised knot heights H_0 .. H_K
    float cst = 0.0f;
    if (a.tails) {
        const float first = 0.5f * width(0), last = 0.5f * width(K - 1);
        float mid = 0.0f;
#pragma unroll
        for (int j = 0; j + 1 < K - 1; ++j) mid += ((eh(j) + eh(j + 1)) / 2.0f) * width(j + 1);
        const float num = 0.5f * first * eh(0) + 0.5f * last * eh(K - 2) + mid;
        cst = num / (1.0f - 0.5f * first - 0.5f * last);
    }
    float hv[KA + 1];                                // KT > 0: the K + 1 knot heights, resolved once
    if constexpr (KT > 0) {
#pragma unroll
        for (int k = 0; k <= K; ++k) hv[k] = !a.tails ? ev[k] : ((k == 0 || k == K) ? cst : ev[k - 1]);
    }
    auto H = [&](int k) {
        if constexpr (KT > 0) return hv[k];
        else {
            if (!a.tails) return eh(k);
            return (k == 0 || k == K) ? cst : eh(k - 1);
        }
    };
    float area = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) area += ((H(k) + H(k + 1)) / 2.0f) * width(k);
    const float omh = 1.0f - a.min_h;
    const NfDivisor da = nf_divisor(area);
    double accw = 0.0, accc = 0.0;
    float loc0 = 0.0f, cdf0 = 0.0f;                  // left edge of bin k: location and cdf
    float loc = 0.0f, w = 1.0f, c = 0.0f, hl = 1.0f, hr = 1.0f;
    int b = 0;
    float hk = a.min_h + omh * nf_div_ieee(H(0), da);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float wk = width(k);
        const float hk1 = a.min_h + omh * nf_div_ieee(H(k + 1), da);
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
template <bool INV, class PT = const float*, bool CACHE = false, int KT = 0, int PREDIV = -1>
__device__ __forceinline__ void cubic_eval(PT p, float x, const RqsConst& a, float& y, float& ld,
                                           int& bin) {
    const int K = KT > 0 ? KT : a.K;          // KT: compile-time bin count (loops unroll, values stay in registers)
    if (a.tails && !(x >= a.left && x <= a.right)) {
        y = x;
        ld = 0.0f;
        bin = -1;
        return;
    }
    const PT uw = p;
    const PT uh = p + K;
    // PREDIV: 1 / 0 = the caller has branched on pre_div != 1 (no per-logit select), -1 = decide here
    const bool scaled = PREDIV < 0 ? a.pre_div != 1.0f : PREDIV != 0;
    const NfDivisor dpre = nf_divisor(a.pre_div);
    if constexpr (CACHE && PREDIV != 0) {                 // divide once, in place
        if (scaled) {
#pragma unroll
            for (int k = 0; k < 2 * K; ++k) uw[k] = nf_div_ieee(uw[k], dpre);
        }
    }
    auto raw = [&](PT u, int k) {
        if constexpr (CACHE || PREDIV == 0) return u[k];
        else return scaled ? nf_div_ieee(u[k], dpre) : u[k];
    };
    const float t = (x - a.left) / a.span;
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        mw = fmaxf(mw, raw(uw, k));
        mh = fmaxf(mh, raw(uh, k));
    }
    // KT > 0: widths / heights in registers (static indices only: the search loop below also picks the bin's and its
    // two neighbours' values); otherwise CACHE keeps them over the logits in LDS
    constexpr int KA = KT > 0 ? KT : 1;
    float wv[KA], hv[KA];
    float sw = 0.0f, sh_ = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float ew = nf_exp(raw(uw, k) - mw), eh = nf_exp(raw(uh, k) - mh);
        if constexpr (KT > 0) { wv[k] = ew; hv[k] = eh; }
        else if constexpr (CACHE) { uw[k] = ew; uh[k] = eh; }
        sw += ew;
        sh_ += eh;
    }
    const NfDivisor dw = nf_divisor(sw), dh = nf_divisor(sh_);
    if constexpr (KT > 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            wv[k] = a.min_w + a.scale_w * nf_div_ieee(wv[k], dw);
            hv[k] = a.min_h + a.scale_h * nf_div_ieee(hv[k], dh);
        }
    } else if constexpr (CACHE) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            uw[k] = a.min_w + a.scale_w * nf_div_ieee(uw[k], dw);
            uh[k] = a.min_h + a.scale_h * nf_div_ieee(uh[k], dh);
        }
    }
    auto width = [&](int k) {
        if constexpr (KT > 0) return wv[k];
        else if constexpr (CACHE) return uw[k];
        else return a.min_w + a.scale_w * nf_div_ieee(nf_exp(raw(uw, k) - mw), dw);
    };
    auto height = [&](int k) {
        if constexpr (KT > 0) return hv[k];
        else if constexpr (CACHE) return uh[k];
        else return a.min_h + a.scale_h * nf_div_ieee(nf_exp(raw(uh, k) - mh), dh);
    };
    // bin search over the running sums
    double accw = 0.0, acch = 0.0;
    float cw0 = 0.0f, ch0 = 0.0f, lw = 0.0f, rw = 1.0f, d = 0.0f;
    int b = 0;
    float wb = 1.0f, hb = 1.0f, w0 = 1.0f, h0 = 1.0f, w1 = 1.0f, h1 = 1.0f;      // bin b and its neighbours (KT > 0)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        accw += (double)width(k);
        acch += (double)height(k);
        float cw1 = (float)accw, ch1 = (float)acch;
        if (k == K - 1) {                              // pinned to 1; the searched one is then raised by 1e-6
            cw1 = INV ? 1.0f : 1.0f + 1e-6f;
            ch1 = INV ? 1.0f + 1e-6f : 1.0f;
        }
        const bool ge = INV ? (t >= ch0) : (t >= cw0);
        if (ge) {
            b = k; lw = cw0; rw = cw1; d = ch0;
            if constexpr (KT > 0) {
                wb = wv[k]; hb = hv[k];
                if (k > 0) { w0 = wv[k - 1]; h0 = hv[k - 1]; }
                if (k < KT - 1) { w1 = wv[k + 1]; h1 = hv[k + 1]; }
            }
        }
        cw0 = cw1;
        ch0 = ch1;
    }
    if constexpr (KT == 0) {
        wb = width(b); hb = height(b);
        if (b > 0) { w0 = width(b - 1); h0 = height(b - 1); }
        if (b < K - 1) { w1 = width(b + 1); h1 = height(b + 1); }
    }
    // knot derivatives at the two ends of bin b
    const float sb = hb / wb;
    auto interior = [&](float wl, float sl, float wr, float sr) {      // knot between bins (wl, sl) and (wr, sr)
        const float m1 = fminf(fabsf(sl), fabsf(sr));
        const float m2 = 0.5f * (wr * sl + wl * sr) / (wl + wr);
        const float sg0 = (sl > 0.f) - (sl < 0.f), sg1 = (sr > 0.f) - (sr < 0.f);
        return fminf(m1, m2) * (sg0 + sg1);
    };
    float dl, dr;
    if (b == 0) {
        dl = (1.0f / (1.0f + nf_exp(-p[2 * K]))) * 3.0f * sb;
    } else {
        dl = interior(w0, h0 / w0, wb, sb);
    }
    if (b == K - 1) {
        dr = (1.0f / (1.0f + nf_exp(-p[2 * K + 1]))) * 3.0f * sb;
    } else {
        dr = interior(wb, sb, w1, h1 / w1);
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

// ===================================================================================================================
// Backward of the three families (training: the reference differentiates linear.py / quadratic.py / cubic.py with
// autograd).  p = the element's parameters in LDS (read only), g = its P-float gradient slot in LDS (written, and used
// as the accumulator arrays of the reverse sweep).  Returns dL/dx.  Forward direction: x is the input, (gy, gl) the
// gradients wrt y and wrt this element's log|dy/dx|.  Inverse direction (inv): x is the inverse pass's OUTPUT, (gy, gl)
// the gradients wrt that output and wrt the inverse's logabsdet (= -log f'(x)); the result follows from the forward
// map f by implicit differentiation - with A = (gy - gl * d(-log f')/dx ... ) / f' the gradient wrt the inverse's input
// is A and the parameter gradient is that of the forward sweep run with upstream (-A, -gl) - which is what autograd
// through the reference's closed-form roots evaluates, up to rounding.
// Everything below is in the normalised coordinates t = (x - left) / span, o = (y - left) / span.
// ===================================================================================================================

__device__ __forceinline__ float lin_bwd_eval(float* p, float* g, float x, float gy, float gl, const RqsConst& a, bool inv) {
    const int K = a.K;
    if (a.tails && !(x >= a.left && x <= a.right)) {
        for (int k = 0; k < K; ++k) g[k] = 0.0f;
        return gy;
    }
    const float t = (x - a.left) / a.span;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, p[k]);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float e = nf_exp(p[k] - m);
        p[k] = e;                                         // cached: the loops below re-read it
        s += e;
    }
    const float pos = t * (float)K;
    int b = (int)floorf(pos);
    b = b >= K ? K - 1 : (b < 0 ? 0 : b);
    const float alpha = pos - (float)b;
    float cb = 0.0f, pb = 0.0f;
    for (int k = 0; k <= b; ++k) {
        const float pk = p[k] / s;
        if (k < b) cb += pk; else pb = pk;
    }
    const float o = cb + alpha * pb;
    const float fp = (float)K * pb;                       // dy/dx (the two spans cancel); d log f' / dx = 0 inside a bin
    float A = 0.0f;
    if (inv) {
        A = gy / fp;
        gy = -A;
        gl = -gl;
    } else if (!(o >= 0.0f && o <= 1.0f)) {
        gy = 0.0f;                                        // torch.clamp(outputs, 0, 1) passes no gradient outside
    }
    const float go = gy * a.span;
    const float S = go * o + gl;                          // sum_k g_pdf[k] * pdf[k]
    for (int k = 0; k < K; ++k) {
        const float pk = p[k] / s;
        const float gp = k < b ? go : (k == b ? go * alpha + gl / pb : 0.0f);
        g[k] = pk * (gp - S);
    }
    return inv ? A : gy * fp;
}

__device__ __forceinline__ float quad_bwd_eval(float* p, float* g, float x, float gy, float gl, const RqsConst& a, bool inv) {
    const int K = a.K, P = a.P;
    if (a.tails && !(x >= a.left && x <= a.right)) {
        for (int k = 0; k < P; ++k) g[k] = 0.0f;
        return gy;
    }
    float* uw = p;
    float* uh = p + K;
    float* gw = g;                 // [K]   dL/d width_k, then dL/d uw_k
    float* gE = g + K;             // [K-1] (tails) dL/d exp(uh_j)  |  [K+1] dL/d H_k
    const float t = (x - a.left) / a.span;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, uw[k]);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float e = nf_exp(uw[k] - m);
        uw[k] = e;
        s += e;
    }
    // cached over the logits (p is this lane's own copy): the exact forward widths, and exp(uh)
    for (int k = 0; k < K; ++k) uw[k] = a.min_w + a.scale_w * (uw[k] / s);
    {
        const int nh = a.tails ? K - 1 : K + 1;
        for (int j = 0; j < nh; ++j) uh[j] = nf_exp(uh[j]);
    }
    const float isw = 1.0f / a.scale_w;
    auto sm = [&](int k) { return (uw[k] - a.min_w) * isw; };
    auto width = [&](int k) { return (k < 0 || k >= K) ? 0.0f : uw[k]; };
    auto eh = [&](int j) { return uh[j]; };
    float cst = 0.0f, num = 0.0f, den = 1.0f;
    if (a.tails) {
        const float first = 0.5f * width(0), last = 0.5f * width(K - 1);
        float mid = 0.0f;
        for (int j = 0; j + 1 < K - 1; ++j) mid += ((eh(j) + eh(j + 1)) / 2.0f) * width(j + 1);
        num = 0.5f * first * eh(0) + 0.5f * last * eh(K - 2) + mid;
        den = 1.0f - 0.5f * first - 0.5f * last;
        cst = num / den;
    }
    auto H = [&](int k) {
        if (!a.tails) return eh(k);
        return (k == 0 || k == K) ? cst : eh(k - 1);
    };
    float area = 0.0f;
    for (int k = 0; k < K; ++k) area += ((H(k) + H(k + 1)) / 2.0f) * width(k);
    const float omh = 1.0f - a.min_h;
    auto hgt = [&](int k) { return a.min_h + omh * (H(k) / area); };
    // the forward-direction search at x (bin edges = the running sum of the widths, as in quad_eval<false>)
    double accw = 0.0, accc = 0.0;
    float loc0 = 0.0f, cdf0 = 0.0f, loc = 0.0f, w = 1.0f, c = 0.0f, hl = 1.0f, hr = 1.0f;
    int b = 0;
    float hk = hgt(0);
    for (int k = 0; k < K; ++k) {
        const float wk = width(k);
        const float hk1 = hgt(k + 1);
        accw += (double)wk;
        accc += (double)(((hk + hk1) / 2.0f) * wk);
        if (t >= loc0) { b = k; loc = loc0; w = wk; c = cdf0; hl = hk; hr = hk1; }
        loc0 = (float)accw;
        cdf0 = (float)accc;
        hk = hk1;
    }
    const float alpha = (t - loc) / w;
    const float dens = alpha * (hr - hl) + hl;
    const float o = (0.5f * (hr - hl) * w) * (alpha * alpha) + (hl * w) * alpha + c;
    const float fp = dens;                                 // do/dt = dy/dx
    const float dldx = ((hr - hl) / (dens * w)) / a.span;
    float A = 0.0f;
    if (inv) {
        A = (gy - gl * dldx) / fp;                         // the inverse's logabsdet is -log f'(x_out)
        gy = -A;
        gl = -gl;
    } else if (!(o >= 0.0f && o <= 1.0f)) {
        gy = 0.0f;
    }
    const float go = gy * a.span;
    const float ga = go * w * dens + gl * (hr - hl) / dens;          // wrt alpha
    const float gloc = -ga / w;
    const float gwd = go * (0.5f * (hr - hl) * (alpha * alpha) + hl * alpha) - ga * alpha / w;
    const float ghl = go * w * (alpha - 0.5f * (alpha * alpha)) + gl * (1.0f - alpha) / dens;
    const float ghr = go * 0.5f * w * (alpha * alpha) + gl * alpha / dens;
    const float gc = go;
    // dL/d h_k (normalised knot heights), k = 0..K: through the cdf prefix and the bin's two knots
    auto gh = [&](int k) {
        float v = 0.0f;
        if (k >= 1 && k - 1 < b) v += gc * 0.5f * width(k - 1);
        if (k < b) v += gc * 0.5f * width(k);
        if (k == b) v += ghl;
        if (k == b + 1) v += ghr;
        return v;
    };
    float sgh = 0.0f;
    for (int k = 0; k <= K; ++k) sgh += gh(k) * H(k);
    const float garea = -(omh / (area * area)) * sgh;
    // dL/d H_k -> uh slots
    float gcst = 0.0f;
    for (int k = 0; k <= K; ++k) {
        const float v = omh * gh(k) / area + garea * 0.5f * (width(k - 1) + width(k));
        if (!a.tails) gE[k] = v;
        else if (k == 0 || k == K) gcst += v;
        else gE[k - 1] = v;
    }
    // dL/d width_k
    for (int k = 0; k < K; ++k) {
        float v = garea * 0.5f * (H(k) + H(k + 1));
        if (k < b) v += gloc + gc * 0.5f * (hgt(k) + hgt(k + 1));
        if (k == b) v += gwd;
        gw[k] = v;
    }
    if (a.tails) {
        const float gnum = gcst / den;
        const float gden = -gcst * num / (den * den);
        const float E0 = eh(0), EL = eh(K - 2);
        gw[0] += -0.25f * gden + 0.25f * E0 * gnum;
        gw[K - 1] += -0.25f * gden + 0.25f * EL * gnum;
        gE[0] += 0.25f * width(0) * gnum;
        gE[K - 2] += 0.25f * width(K - 1) * gnum;
        for (int j = 0; j + 1 < K - 1; ++j) {
            const float wj = width(j + 1);
            gE[j] += gnum * 0.5f * wj;
            gE[j + 1] += gnum * 0.5f * wj;
            gw[j + 1] += gnum * 0.5f * (eh(j) + eh(j + 1));
        }
        for (int j = 0; j < K - 1; ++j) gE[j] *= eh(j);
    } else {
        for (int k = 0; k <= K; ++k) gE[k] *= eh(k);
    }
    // softmax of the widths
    float S = 0.0f;
    for (int k = 0; k < K; ++k) S += a.scale_w * gw[k] * sm(k);
    for (int k = 0; k < K; ++k) gw[k] = sm(k) * (a.scale_w * gw[k] - S);
    return inv ? A : gy * fp + gl * dldx;
}

__device__ __forceinline__ float cubic_bwd_eval(float* p, float* g, float x, float gy, float gl, const RqsConst& a, bool inv) {
    const int K = a.K, P = a.P;
    if (a.tails && !(x >= a.left && x <= a.right)) {
        for (int k = 0; k < P; ++k) g[k] = 0.0f;
        return gy;
    }
    float* uw = p;
    float* uh = p + K;
    const bool scaled = a.pre_div != 1.0f;
    auto raw = [&](const float* u, int k) { return scaled ? u[k] / a.pre_div : u[k]; };
    const float t = (x - a.left) / a.span;
    float mw = -INFINITY, mh = -INFINITY;
    for (int k = 0; k < K; ++k) {
        mw = fmaxf(mw, raw(uw, k));
        mh = fmaxf(mh, raw(uh, k));
    }
    float sw = 0.0f, sh_ = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float ew = nf_exp(raw(uw, k) - mw), eh = nf_exp(raw(uh, k) - mh);
        uw[k] = ew;
        uh[k] = eh;
        sw += ew;
        sh_ += eh;
    }
    for (int k = 0; k < K; ++k) {                         // cached over the logits: the exact forward widths / heights
        uw[k] = a.min_w + a.scale_w * (uw[k] / sw);
        uh[k] = a.min_h + a.scale_h * (uh[k] / sh_);
    }
    const float isw = 1.0f / a.scale_w, ish = 1.0f / a.scale_h;
    auto smw = [&](int k) { return (uw[k] - a.min_w) * isw; };
    auto smh = [&](int k) { return (uh[k] - a.min_h) * ish; };
    auto width = [&](int k) { return uw[k]; };
    auto height = [&](int k) { return uh[k]; };
    double accw = 0.0;
    float cw0 = 0.0f, lw = 0.0f;
    int b = 0;
    for (int k = 0; k < K; ++k) {
        accw += (double)width(k);
        if (t >= cw0) { b = k; lw = cw0; }
        cw0 = (float)accw;
    }
    const float wb = width(b), hb = height(b), sb = hb / wb;
    float w0 = 1.0f, h0 = 1.0f, s0 = 1.0f, w1 = 1.0f, h1 = 1.0f, s1 = 1.0f, sigl = 0.0f, sigr = 0.0f;
    float dl, dr;
    auto interior = [&](float wa, float sa, float wc, float sc) {
        return 2.0f * fminf(fminf(sa, sc), 0.5f * (wc * sa + wa * sc) / (wa + wc));       // slopes are positive
    };
    if (b == 0) {
        sigl = 1.0f / (1.0f + nf_exp(-p[2 * K]));
        dl = sigl * 3.0f * sb;
    } else {
        w0 = width(b - 1); h0 = height(b - 1); s0 = h0 / w0;
        dl = interior(w0, s0, wb, sb);
    }
    if (b == K - 1) {
        sigr = 1.0f / (1.0f + nf_exp(-p[2 * K + 1]));
        dr = sigr * 3.0f * sb;
    } else {
        w1 = width(b + 1); h1 = height(b + 1); s1 = h1 / w1;
        dr = interior(wb, sb, w1, s1);
    }
    const float ca = (dl + dr - 2.0f * sb) / (wb * wb);
    const float cb = (3.0f * sb - 2.0f * dl - dr) / wb;
    const float cc = dl;
    const float s = t - lw;
    const float Dv = 3.0f * ca * (s * s) + 2.0f * cb * s + cc;        // do/dt = dy/dx
    const float D2 = 6.0f * ca * s + 2.0f * cb;
    float g_ca, g_cb, g_cc, g_d, g_lw, ret;
    if (!inv) {
        const float go = gy * a.span;
        const float gs = go * Dv + gl * D2 / Dv;
        ret = gs / a.span;
        g_lw = -gs;
        g_ca = go * (s * s * s) + gl * 3.0f * (s * s) / Dv;
        g_cb = go * (s * s) + gl * 2.0f * s / Dv;
        g_cc = go * s + gl / Dv;
        g_d = go;
    } else {
        // the reference solves the cubic, or - for |a| < 1e-3 - the quadratic b s^2 + c s + d = t (cubic.py:191-198)
        const bool quad = fabsf(ca) < 1e-3f;
        const float Gp = quad ? 2.0f * cb * s + cc : Dv;
        const float s3 = quad ? 0.0f : s * s * s;
        const float gon = gy * a.span;
        const float gs = gon - gl * D2 / Dv;
        const float q = gs / Gp;
        ret = q / a.span;
        g_lw = gon;
        g_ca = -gl * 3.0f * (s * s) / Dv - q * s3;
        g_cb = -gl * 2.0f * s / Dv - q * (s * s);
        g_cc = -gl / Dv - q * s;
        g_d = -q;
    }
    // coefficients -> knot derivatives, slope and width of bin b
    const float iw = 1.0f / wb, iw2 = iw * iw;
    const float g_dl = g_ca * iw2 - 2.0f * g_cb * iw + g_cc;
    const float g_dr = g_ca * iw2 - g_cb * iw;
    float g_sb = -2.0f * g_ca * iw2 + 3.0f * g_cb * iw;
    float g_wb = -2.0f * ca * g_ca * iw - cb * g_cb * iw;
    float g_w0 = 0.0f, g_w1 = 0.0f, g_s0 = 0.0f, g_s1 = 0.0f, g_ul = 0.0f, g_ur = 0.0f;
    if (b == 0) {
        g_ul = g_dl * 3.0f * sb * sigl * (1.0f - sigl);
        g_sb += g_dl * 3.0f * sigl;
    } else {
        const float m2 = 0.5f * (wb * s0 + w0 * sb) / (w0 + wb), gm = 2.0f * g_dl;
        if (fminf(s0, sb) <= m2) {
            if (s0 <= sb) g_s0 += gm; else g_sb += gm;
        } else {
            const float id = 1.0f / (w0 + wb);
            g_s0 += gm * 0.5f * wb * id;
            g_sb += gm * 0.5f * w0 * id;
            g_wb += gm * (0.5f * s0 - m2) * id;
            g_w0 += gm * (0.5f * sb - m2) * id;
        }
    }
    if (b == K - 1) {
        g_ur = g_dr * 3.0f * sb * sigr * (1.0f - sigr);
        g_sb += g_dr * 3.0f * sigr;
    } else {
        const float m2 = 0.5f * (w1 * sb + wb * s1) / (wb + w1), gm = 2.0f * g_dr;
        if (fminf(sb, s1) <= m2) {
            if (sb <= s1) g_sb += gm; else g_s1 += gm;
        } else {
            const float id = 1.0f / (wb + w1);
            g_sb += gm * 0.5f * w1 * id;
            g_s1 += gm * 0.5f * wb * id;
            g_wb += gm * (0.5f * s1 - m2) * id;
            g_w1 += gm * (0.5f * sb - m2) * id;
        }
    }
    const float g_hb = g_sb * iw;
    g_wb += -g_sb * sb * iw;
    float g_h0 = 0.0f, g_h1 = 0.0f;
    if (b > 0) { g_h0 = g_s0 / w0; g_w0 += -g_s0 * s0 / w0; }
    if (b < K - 1) { g_h1 = g_s1 / w1; g_w1 += -g_s1 * s1 / w1; }
    // back through min + scale * softmax(u / pre_div) of the widths and of the heights
    auto gwk = [&](int k) { return (k < b ? g_lw : 0.0f) + (k == b - 1 ? g_w0 : 0.0f) + (k == b ? g_wb : 0.0f) + (k == b + 1 ? g_w1 : 0.0f); };
    auto ghk = [&](int k) { return (k < b ? g_d : 0.0f) + (k == b - 1 ? g_h0 : 0.0f) + (k == b ? g_hb : 0.0f) + (k == b + 1 ? g_h1 : 0.0f); };
    float Sw = 0.0f, Sh = 0.0f;
    for (int k = 0; k < K; ++k) {
        Sw += a.scale_w * gwk(k) * smw(k);
        Sh += a.scale_h * ghk(k) * smh(k);
    }
    for (int k = 0; k < K; ++k) {
        g[k] = smw(k) * (a.scale_w * gwk(k) - Sw) / a.pre_div;
        g[K + k] = smh(k) * (a.scale_h * ghk(k) - Sh) / a.pre_div;
    }
    g[2 * K] = g_ul;
    g[2 * K + 1] = g_ur;
    return ret;
}

// The same reverse sweep with the bin count as a template constant: the element's 2 KT + 2 parameters move to registers
// once (every index below is static after unrolling; the runtime-K version above re-reads its cached widths / heights
// from LDS in every loop, 2-way bank conflicts at the even stride P = 2K + 2), quotients by one refined reciprocal per
// denominator (nf_div_ieee: the bits of `/` in the normal range), the bin's neighbours picked by the search loop, and
// the gradients written over the parameters IN PLACE - no second LDS region.  Same formulas in the same order.
template <int KT>
__device__ __forceinline__ float cubic_bwd_eval_t(float* p, float x, float gy, float gl, const RqsConst& a, bool inv) {
    constexpr int K = KT, P = 2 * KT + 2;
    typedef __attribute__((address_space(3))) float* lds_w;
    lds_w pl = (lds_w)p;
    if (a.tails && !(x >= a.left && x <= a.right)) {
#pragma unroll
        for (int k = 0; k < P; ++k) pl[k] = 0.0f;
        return gy;
    }
    float wv[K], hv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { wv[k] = pl[k]; hv[k] = pl[K + k]; }
    const float ul = pl[2 * K], ur = pl[2 * K + 1];
    const NfDivisor dpre = nf_divisor(a.pre_div);
    if (a.pre_div != 1.0f) {
#pragma unroll
        for (int k = 0; k < K; ++k) { wv[k] = nf_div_ieee(wv[k], dpre); hv[k] = nf_div_ieee(hv[k], dpre); }
    }
    const float t = nf_div_ieee(x - a.left, nf_divisor(a.span));
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) { mw = fmaxf(mw, wv[k]); mh = fmaxf(mh, hv[k]); }
    float sw = 0.0f, sh_ = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wv[k] = nf_exp(wv[k] - mw);
        hv[k] = nf_exp(hv[k] - mh);
        sw += wv[k];
        sh_ += hv[k];
    }
    const NfDivisor dsw = nf_divisor(sw), dsh = nf_divisor(sh_);
    float smw[K], smh[K];                                  // the two softmaxes; widths / heights = min + scale * softmax
#pragma unroll
    for (int k = 0; k < K; ++k) {
        smw[k] = nf_div_ieee(wv[k], dsw);
        smh[k] = nf_div_ieee(hv[k], dsh);
        wv[k] = a.min_w + a.scale_w * smw[k];
        hv[k] = a.min_h + a.scale_h * smh[k];
    }
    double accw = 0.0;
    float cw0 = 0.0f, lw = 0.0f;
    int b = 0;
    float wb = 1.0f, hb = 1.0f, w0 = 1.0f, h0 = 1.0f, w1 = 1.0f, h1 = 1.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        accw += (double)wv[k];
        if (t >= cw0) {
            b = k; lw = cw0; wb = wv[k]; hb = hv[k];
            if (k > 0) { w0 = wv[k - 1]; h0 = hv[k - 1]; }
            if (k < K - 1) { w1 = wv[k + 1]; h1 = hv[k + 1]; }
        }
        cw0 = (float)accw;
    }
    const NfDivisor dwb = nf_divisor(wb), dw0 = nf_divisor(w0), dw1 = nf_divisor(w1);
    const float sb = nf_div_ieee(hb, dwb), s0 = nf_div_ieee(h0, dw0), s1 = nf_div_ieee(h1, dw1);
    const bool first = b == 0, last = b == K - 1;
    // one sigmoid serves whichever end bin the element sits in (K >= 2: never both)
    const float ue = first ? ul : ur;
    const float sig = 1.0f / (1.0f + nf_exp(-ue));          // (`/`: the denominator may be +inf)
    const NfDivisor dl01 = nf_divisor(w0 + wb), dr01 = nf_divisor(wb + w1);
    const float m2l = nf_div_ieee(0.5f * (wb * s0 + w0 * sb), dl01);
    const float m2r = nf_div_ieee(0.5f * (w1 * sb + wb * s1), dr01);
    const float dl = first ? sig * 3.0f * sb : 2.0f * fminf(fminf(s0, sb), m2l);
    const float dr = last ? sig * 3.0f * sb : 2.0f * fminf(fminf(sb, s1), m2r);
    const float iw = dwb.r, iw2 = iw * iw;
    const float ca = nf_div_ieee(dl + dr - 2.0f * sb, nf_divisor(wb * wb));
    const float cb = nf_div_ieee(3.0f * sb - 2.0f * dl - dr, dwb);
    const float cc = dl;
    const float s = t - lw;
    const float Dv = 3.0f * ca * (s * s) + 2.0f * cb * s + cc;
    const float D2 = 6.0f * ca * s + 2.0f * cb;
    const NfDivisor dDv = nf_divisor(Dv);
    const float glv = nf_div_ieee(gl, dDv);                // gl / f'
    float g_ca, g_cb, g_cc, g_d, g_lw, ret;
    if (!inv) {
        const float go = gy * a.span;
        const float gs = go * Dv + glv * D2;
        ret = nf_div_ieee(gs, nf_divisor(a.span));
        g_lw = -gs;
        g_ca = go * (s * s * s) + glv * 3.0f * (s * s);
        g_cb = go * (s * s) + glv * 2.0f * s;
        g_cc = go * s + glv;
        g_d = go;
    } else {
        const bool quad = fabsf(ca) < 1e-3f;
        const float Gp = quad ? 2.0f * cb * s + cc : Dv;
        const float s3 = quad ? 0.0f : s * s * s;
        const float gon = gy * a.span;
        const float gs = gon - glv * D2;
        const float q = nf_div_ieee(gs, nf_divisor(Gp));
        ret = nf_div_ieee(q, nf_divisor(a.span));
        g_lw = gon;
        g_ca = -glv * 3.0f * (s * s) - q * s3;
        g_cb = -glv * 2.0f * s - q * (s * s);
        g_cc = -glv - q * s;
        g_d = -q;
    }
    const float g_dl = g_ca * iw2 - 2.0f * g_cb * iw + g_cc;
    const float g_dr = g_ca * iw2 - g_cb * iw;
    float g_sb = -2.0f * g_ca * iw2 + 3.0f * g_cb * iw;
    float g_wb = -2.0f * ca * g_ca * iw - cb * g_cb * iw;
    float g_w0 = 0.0f, g_w1 = 0.0f, g_s0 = 0.0f, g_s1 = 0.0f, g_ul = 0.0f, g_ur = 0.0f;
    {
        const float gm = 2.0f * g_dl;
        const bool edge = fminf(s0, sb) <= m2l;            // the min picked a slope, not the weighted mean
        const float id = dl01.r;
        g_ul = first ? g_dl * 3.0f * sb * sig * (1.0f - sig) : 0.0f;
        g_sb += first ? g_dl * 3.0f * sig : (edge ? (s0 <= sb ? 0.0f : gm) : gm * 0.5f * w0 * id);
        g_s0 += first ? 0.0f : (edge ? (s0 <= sb ? gm : 0.0f) : gm * 0.5f * wb * id);
        g_wb += first || edge ? 0.0f : gm * (0.5f * s0 - m2l) * id;
        g_w0 += first || edge ? 0.0f : gm * (0.5f * sb - m2l) * id;
    }
    {
        const float gm = 2.0f * g_dr;
        const bool edge = fminf(sb, s1) <= m2r;
        const float id = dr01.r;
        g_ur = last ? g_dr * 3.0f * sb * sig * (1.0f - sig) : 0.0f;
        g_sb += last ? g_dr * 3.0f * sig : (edge ? (sb <= s1 ? gm : 0.0f) : gm * 0.5f * w1 * id);
        g_s1 += last ? 0.0f : (edge ? (sb <= s1 ? 0.0f : gm) : gm * 0.5f * wb * id);
        g_wb += last || edge ? 0.0f : gm * (0.5f * s1 - m2r) * id;
        g_w1 += last || edge ? 0.0f : gm * (0.5f * sb - m2r) * id;
    }
    const float g_hb = g_sb * iw;
    g_wb += -g_sb * sb * iw;
    const float g_h0 = first ? 0.0f : g_s0 * dw0.r, g_h1 = last ? 0.0f : g_s1 * dw1.r;
    g_w0 += first ? 0.0f : -g_s0 * s0 * dw0.r;
    g_w1 += last ? 0.0f : -g_s1 * s1 * dw1.r;
    // back through min + scale * softmax(u / pre_div) of the widths and of the heights
    float gwv[K], ghv[K];
    float Sw = 0.0f, Sh = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        gwv[k] = a.scale_w * ((k < b ? g_lw : 0.0f) + (k == b - 1 ? g_w0 : 0.0f) + (k == b ? g_wb : 0.0f) + (k == b + 1 ? g_w1 : 0.0f));
        ghv[k] = a.scale_h * ((k < b ? g_d : 0.0f) + (k == b - 1 ? g_h0 : 0.0f) + (k == b ? g_hb : 0.0f) + (k == b + 1 ? g_h1 : 0.0f));
        Sw += gwv[k] * smw[k];
        Sh += ghv[k] * smh[k];
    }
    const float ipd = dpre.r;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        pl[k] = smw[k] * (gwv[k] - Sw) * ipd;
        pl[K + k] = smh[k] * (ghv[k] - Sh) * ipd;
    }
    pl[2 * K] = g_ul;
    pl[2 * K + 1] = g_ur;
    return ret;
}

// Linear and quadratic reverse sweeps with the bin count as a template constant, on the same scheme as cubic_bwd_eval_t:
// the element's parameters move to registers once, every index is static after unrolling, quotients share one refined
// reciprocal per denominator (nf_div_ieee), the gradients are written over the parameters in place.  Same formulas in
// the same order as lin_bwd_eval / quad_bwd_eval; the softmax values are kept (those re-derive them from the widths).
template <int KT>
__device__ __forceinline__ float lin_bwd_eval_t(float* p, float x, float gy, float gl, const RqsConst& a, bool inv) {
    constexpr int K = KT;
    typedef __attribute__((address_space(3))) float* lds_w;
    lds_w pl = (lds_w)p;
    if (a.tails && !(x >= a.left && x <= a.right)) {
#pragma unroll
        for (int k = 0; k < K; ++k) pl[k] = 0.0f;
        return gy;
    }
    float e[K];
#pragma unroll
    for (int k = 0; k < K; ++k) e[k] = pl[k];
    const float t = nf_div_ieee(x - a.left, nf_divisor(a.span));
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) m = fmaxf(m, e[k]);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        e[k] = nf_exp(e[k] - m);
        s += e[k];
    }
    const NfDivisor ds = nf_divisor(s);
    const float pos = t * (float)K;
    int b = (int)floorf(pos);
    b = b >= K ? K - 1 : (b < 0 ? 0 : b);
    const float alpha = pos - (float)b;
    float cb = 0.0f, pb = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        e[k] = nf_div_ieee(e[k], ds);                     // the normalised pdf
        cb += k < b ? e[k] : 0.0f;
        pb = k == b ? e[k] : pb;
    }
    const float o = cb + alpha * pb;
    const float fp = (float)K * pb;
    float A = 0.0f;
    if (inv) {
        A = gy / fp;
        gy = -A;
        gl = -gl;
    } else if (!(o >= 0.0f && o <= 1.0f)) {
        gy = 0.0f;
    }
    const float go = gy * a.span;
    const float S = go * o + gl;
    const float gpb = go * alpha + gl / pb;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float gp = k < b ? go : (k == b ? gpb : 0.0f);
        pl[k] = e[k] * (gp - S);
    }
    return inv ? A : gy * fp;
}

template <int KT>
__device__ __forceinline__ float quad_bwd_eval_t(float* p, float x, float gy, float gl, const RqsConst& a, bool inv) {
    constexpr int K = KT;
    typedef __attribute__((address_space(3))) float* lds_w;
    lds_w pl = (lds_w)p;
    const int P = a.P;
    if (a.tails && !(x >= a.left && x <= a.right)) {
        for (int k = 0; k < P; ++k) pl[k] = 0.0f;
        return gy;
    }
    const bool tails = a.tails != 0;
    float wv[K], ev[K + 1];
#pragma unroll
    for (int k = 0; k < K; ++k) wv[k] = pl[k];
#pragma unroll
    for (int j = 0; j < K + 1; ++j) ev[j] = (j < K - 1 || !tails) ? nf_exp(pl[K + j]) : 0.0f;     // K - 1 or K + 1 heights
    const float t = nf_div_ieee(x - a.left, nf_divisor(a.span));
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) m = fmaxf(m, wv[k]);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wv[k] = nf_exp(wv[k] - m);
        s += wv[k];
    }
    const NfDivisor ds = nf_divisor(s);
#pragma unroll
    for (int k = 0; k < K; ++k) wv[k] = a.min_w + a.scale_w * nf_div_ieee(wv[k], ds);
    float cst = 0.0f, num = 0.0f, den = 1.0f;
    if (tails) {
        const float first = 0.5f * wv[0], last = 0.5f * wv[K - 1];
        float mid = 0.0f;
#pragma unroll
        for (int j = 0; j + 1 < K - 1; ++j) mid += ((ev[j] + ev[j + 1]) / 2.0f) * wv[j + 1];
        num = 0.5f * first * ev[0] + 0.5f * last * ev[K - 2] + mid;
        den = 1.0f - 0.5f * first - 0.5f * last;
        cst = num / den;
    }
    float Hk[K + 1];                                     // unnormalised knot heights
#pragma unroll
    for (int k = 0; k <= K; ++k) Hk[k] = !tails ? ev[k] : ((k == 0 || k == K) ? cst : ev[k > 0 ? k - 1 : 0]);
    float area = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) area += ((Hk[k] + Hk[k + 1]) / 2.0f) * wv[k];
    const float omh = 1.0f - a.min_h;
    const NfDivisor da = nf_divisor(area);
    float hg[K + 1];                                     // normalised knot heights
#pragma unroll
    for (int k = 0; k <= K; ++k) hg[k] = a.min_h + omh * nf_div_ieee(Hk[k], da);
    double accw = 0.0, accc = 0.0;
    float loc0 = 0.0f, cdf0 = 0.0f, loc = 0.0f, w = 1.0f, c = 0.0f, hl = 1.0f, hr = 1.0f;
    int b = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        accw += (double)wv[k];
        accc += (double)(((hg[k] + hg[k + 1]) / 2.0f) * wv[k]);
        if (t >= loc0) { b = k; loc = loc0; w = wv[k]; c = cdf0; hl = hg[k]; hr = hg[k + 1]; }
        loc0 = (float)accw;
        cdf0 = (float)accc;
    }
    const NfDivisor dw = nf_divisor(w);
    const float alpha = nf_div_ieee(t - loc, dw);
    const float dens = alpha * (hr - hl) + hl;
    const float o = (0.5f * (hr - hl) * w) * (alpha * alpha) + (hl * w) * alpha + c;
    const float fp = dens;
    const NfDivisor dd = nf_divisor(dens);
    const float dldx = nf_div_ieee(nf_div_ieee(hr - hl, nf_divisor(dens * w)), nf_divisor(a.span));
    float A = 0.0f;
    if (inv) {
        A = nf_div_ieee(gy - gl * dldx, dd);
        gy = -A;
        gl = -gl;
    } else if (!(o >= 0.0f && o <= 1.0f)) {
        gy = 0.0f;
    }
    const float go = gy * a.span;
    const float ga = go * w * dens + nf_div_ieee(gl * (hr - hl), dd);
    const float gloc = -nf_div_ieee(ga, dw);
    const float gwd = go * (0.5f * (hr - hl) * (alpha * alpha) + hl * alpha) - nf_div_ieee(ga * alpha, dw);
    const float ghl = go * w * (alpha - 0.5f * (alpha * alpha)) + nf_div_ieee(gl * (1.0f - alpha), dd);
    const float ghr = go * 0.5f * w * (alpha * alpha) + nf_div_ieee(gl * alpha, dd);
    const float gc = go;
    float gh[K + 1];                                     // dL/d (normalised knot height k)
    float sgh = 0.0f;
#pragma unroll
    for (int k = 0; k <= K; ++k) {
        float v = 0.0f;
        if (k >= 1) v += (k - 1 < b) ? gc * 0.5f * wv[k - 1] : 0.0f;
        if (k < K) v += (k < b) ? gc * 0.5f * wv[k] : 0.0f;
        v += (k == b) ? ghl : 0.0f;
        v += (k == b + 1) ? ghr : 0.0f;
        gh[k] = v;
        sgh += v * Hk[k];
    }
    const float garea = -nf_div_ieee(omh, nf_divisor(area * area)) * sgh;
    float gw[K];
    float (&gE)[K + 1] = gh;                             // the slot of knot k is free once its own value is taken
    float gcst = 0.0f;
#pragma unroll
    for (int k = 0; k <= K; ++k) {
        const float wl = k >= 1 ? wv[k - 1] : 0.0f, wr = k < K ? wv[k] : 0.0f;
        const float v = nf_div_ieee(omh * gh[k], da) + garea * 0.5f * (wl + wr);
        if (!tails) gE[k] = v;
        else if (k == 0 || k == K) gcst += v;
        else gE[k - 1] = v;
    }
    if (tails) gE[K - 1] = gE[K] = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float v = garea * 0.5f * (Hk[k] + Hk[k + 1]);
        v += (k < b) ? gloc + gc * 0.5f * (hg[k] + hg[k + 1]) : 0.0f;
        v += (k == b) ? gwd : 0.0f;
        gw[k] = v;
    }
    if (tails) {
        const float gnum = gcst / den;
        const float gden = -gcst * num / (den * den);
        gw[0] += -0.25f * gden + 0.25f * ev[0] * gnum;
        gw[K - 1] += -0.25f * gden + 0.25f * ev[K - 2] * gnum;
        gE[0] += 0.25f * wv[0] * gnum;
        gE[K - 2] += 0.25f * wv[K - 1] * gnum;
#pragma unroll
        for (int j = 0; j + 1 < K - 1; ++j) {
            const float wj = wv[j + 1];
            gE[j] += gnum * 0.5f * wj;
            gE[j + 1] += gnum * 0.5f * wj;
            gw[j + 1] += gnum * 0.5f * (ev[j] + ev[j + 1]);
        }
    }
#pragma unroll
    for (int j = 0; j <= K; ++j) gE[j] *= ev[j];
    // (the softmax values back from the widths, as quad_bwd_eval has them: eight registers less across the sweep)
    const float isw = 1.0f / a.scale_w;
    float S = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        wv[k] = (wv[k] - a.min_w) * isw;
        S += a.scale_w * gw[k] * wv[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) pl[k] = wv[k] * (a.scale_w * gw[k] - S);
#pragma unroll
    for (int j = 0; j <= K; ++j)
        if (j < K - 1 || !tails) pl[K + j] = gE[j];
    return inv ? A : gy * fp + gl * dldx;
}

