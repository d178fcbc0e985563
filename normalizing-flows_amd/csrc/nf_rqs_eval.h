// Per-element rational-quadratic spline evaluation shared by the standalone kernel (nf_spline.hip)
// and the fused MADE-final-layer + spline kernel (nf_fused.hip).  Both translation units are built
// with -ffp-contract=off: every product and sum rounds once, like the reference's separate ATen ops
// (models/splines/rational_quadratic.py:54-149; search: helper_functions.py:507-512).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// Transcendentals and divisions of the spline: the hardware v_exp_f32 / v_log_f32 / v_rcp_f32 forms
// (about 1-2 ulp) instead of the correctly rounded library routines.  The reference's own CPU kernels
// (Sleef exp at 1-2 ulp, fexp_u20 inside softmax) are no closer to correctly rounded than this, so the
// parity bars (bit-exact bins away from knots, 1e-5 on outputs / logdet) are unaffected, while the
// per-element VALU cost - what bounds the standalone kernel and the fused epilogue - drops by ~45 %.
// Build with -DNF_SPLINE_FAST=0 for the library forms.
#ifndef NF_SPLINE_FAST
#define NF_SPLINE_FAST 1
#endif
__device__ __forceinline__ float nf_exp(float x) {
#if NF_SPLINE_FAST
    // exp2(x*log2e) with the rounding error of the product carried to first order: for |x| ~ 10 the plain
    // product is off by ~1e-6 in the exponent, which the 2T knot scale (T = 50) would turn into 1e-5 knots
    const float th = x * 1.44269504088896341f;
    const float tl = fmaf(x, 1.44269504088896341f, -th) + x * 1.925963033500e-8f;
    const float e = __builtin_amdgcn_exp2f(th);
    return fmaf(e, tl * 0.693147180559945309f, e);
#else
    return expf(x);
#endif
}
__device__ __forceinline__ float nf_log(float x) {
#if NF_SPLINE_FAST
    return __builtin_amdgcn_logf(x) * 0.693147180559945309f;
#else
    return logf(x);
#endif
}
__device__ __forceinline__ float nf_rcp(float x) {
#if NF_SPLINE_FAST
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
__device__ __forceinline__ float nf_div(float a, float b) {
#if NF_SPLINE_FAST
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}

// a / b correctly rounded (what the reference's `/` produces) for MANY numerators over one denominator: the
// Newton-refined reciprocal is computed once, each quotient then takes the same multiply + two fused residual
// corrections as the compiler's own IEEE expansion (v_div_scale / v_div_fmas / v_div_fixup left out: they only matter
// when b, 1/b or a/b leave the normal range - here b is a softmax denominator or a normalising area, and a quotient
// that underflows is a bin of zero mass either way).  5 instructions per quotient instead of 11.
struct NfDivisor {
    float b, r;
};
__device__ __forceinline__ NfDivisor nf_divisor(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, r, 1.0f);
    r = fmaf(e, r, r);
    return NfDivisor{b, r};
}
__device__ __forceinline__ float nf_div_ieee(float a, const NfDivisor& d) {
    float q = a * d.r;
    float rem = fmaf(-d.b, q, a);
    q = fmaf(rem, d.r, q);
    rem = fmaf(-d.b, q, a);
    return fmaf(rem, d.r, q);
}

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
#if NF_SPLINE_FAST
    return u > 20.0f ? u : nf_log(1.0f + nf_exp(u));
#else
    return u > 20.0f ? u : log1pf(nf_exp(u));
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Evaluates one element.  p points at its P parameters in LDS, STRIDE floats apart (1 = contiguous;
// the fused kernel keeps its parameter tile feature-major, one row of samples per parameter).
// BASE2: the width / height logits arrive already multiplied by log2(e) (folded into the weights that
// produced them), so the softmax exponentials are bare exp2 instructions.
// STRAIGHT: no early exit for tail elements (they are evaluated at the interval centre and replaced by
// the identity at the end), so that several evaluations inlined back to back form one basic block the
// scheduler can interleave.
// PT: pointer type of the parameters (an explicit LDS address-space pointer keeps a runtime-K caller off flat loads).
// PREDIV false: the caller guarantees pre_div == 1 (the fused kernels fold it into the weights), which removes a
// multiply and a select per logit.
template <int KT, bool INV, int STRIDE = 1, bool BASE2 = false, bool STRAIGHT = false, class PT = const float*,
          bool PREDIV = true>
__device__ __forceinline__ void rqs_eval(PT p, float x, const RqsConst& a, float& y, float& ld, int& bin) {
    const int K = KT > 0 ? KT : a.K;
    const bool outside = a.tails && !(x >= a.left && x <= a.right);   // linear tails: identity, logdet 0 (NaN too)
    const float x_in = x;
    if (STRAIGHT) {
        x = outside ? 0.5f * (a.left + a.right) : x;
    } else if (outside) {
        y = x;
        ld = 0.0f;
        bin = -1;
        return;
    }
    const PT uw = p;
    const PT uh = p + K * STRIDE;
    const PT ud = p + 2 * K * STRIDE;
    const bool scaled = PREDIV && a.pre_div != 1.0f;

    // softmax denominators: max, then sum of exp(u - max) left to right
    auto raw = [&](PT u, int k) {
        float v = u[k * STRIDE];
        return scaled ? nf_div(v, a.pre_div) : v;
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
        w = BASE2 ? __builtin_amdgcn_exp2f(w - mw) : nf_exp(w - mw);
        h = BASE2 ? __builtin_amdgcn_exp2f(h - mh) : nf_exp(h - mh);
        if constexpr (KT > 0) { ew[k] = w; eh[k] = h; }
        sw += w;
        sh += h;
    }
    const float rw = nf_rcp(sw), rh = nf_rcp(sh);

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
        else if (BASE2) { w = __builtin_amdgcn_exp2f(raw(uw, k) - mw); h = __builtin_amdgcn_exp2f(raw(uh, k) - mh); }
        else { w = nf_exp(raw(uw, k) - mw); h = nf_exp(raw(uh, k) - mh); }
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
        // (value selects, five v_cndmask: as an `if` block hipcc made this a divergent branch per knot - saveexec,
        // branch, moves, restore - in the fused epilogues)
        bin = ge ? k : bin;
        x0 = ge ? kw0 : x0;
        x1 = ge ? kw1 : x1;
        y0 = ge ? kh0 : y0;
        y1 = ge ? kh1 : y1;
        kw0 = kw1;
        kh0 = kh1;
    }

    // slopes at the two knots of the bin
    float u0, u1;
    if (a.tails) {
        // both reads unconditional at clamped indices, then a VALUE select: a select between the address of the
        // constant and an LDS address would turn these into flat loads, which wait on every outstanding global load
        const float v0 = ud[(bin == 0 ? 0 : bin - 1) * STRIDE];
        const float v1 = ud[(bin == K - 1 ? K - 2 : bin) * STRIDE];
        u0 = bin == 0 ? a.tail_const : v0;
        u1 = bin == K - 1 ? a.tail_const : v1;
    } else {
        u0 = ud[bin * STRIDE];
        u1 = ud[(bin + 1) * STRIDE];
    }
    const float d0 = a.min_d + softplus_f(u0);
    const float d1 = a.min_d + softplus_f(u1);

    const float bw = x1 - x0;
    const float bh = y1 - y0;
    // IEEE divisions (s and t feed the cancelling terms below); the forward direction has two over the same
    // denominator: one refined reciprocal, then nf_div_ieee's multiply + two residual corrections each - the bits of
    // `/` for a bin width in the normal range (>= min_bin_width * span)
    const NfDivisor dbw = nf_divisor(bw);
    const float s = INV ? bh / bw : nf_div_ieee(bh, dbw);
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
        t = nf_div_ieee(x - x0, dbw);
    }
    const float t1mt = t * (1.0f - t);
    den = s + curv * t1mt;
    if (!INV) {
        const float num = bh * (s * (t * t) + d0 * t1mt);
        y = y0 + nf_div(num, den);
    }
    const float omt = 1.0f - t;
    const float dnum = (s * s) * (d1 * (t * t) + 2.0f * s * t1mt + d0 * (omt * omt));
    const float l = nf_log(dnum) - 2.0f * nf_log(den);
    ld = INV ? -l : l;
    if (STRAIGHT && outside) {
        y = x_in;
        ld = 0.0f;
        bin = -1;
    }
}


// Inverse of one element per lane PAIR: lanes l and l + 32 of a wave (all 64 lanes active, same p and x in both).
// The softmax and the knot prefix sums are the long, identical part of the work for widths and heights, so the lanes
// with `widths` set build the width knots while their partners build the height knots on the same instruction
// stream; the partners search the bin, fetch the two width knots of that bin by lane exchange and finish.  Same
// operations in the same order as rqs_eval<KT, true> (pre_div = 1): y, ld, bin are valid in the lanes with
// widths == false.
// Value of lane (l ^ 32): gfx950's v_permlane32_swap_b32 exchanges the two 32-lane halves of a register pair on the
// vector ALU - an LDS-path permute (ds_bpermute_b32, what __shfl_xor compiles to) costs an LDS round trip, and
// rqs_inverse_pair sits on the row-resident inverse's critical path with three of them.  Every lane must be active.
__device__ __forceinline__ int nf_xor32(int v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);     // {v with its upper half <- lower half,
    return (__lane_id() & 32) ? (int)r[0] : (int)r[1];                       //  v with its lower half <- upper half}
}
__device__ __forceinline__ float nf_xor32(float v) { return __int_as_float(nf_xor32(__float_as_int(v))); }

template <int KT, int STRIDE, class PT>
__device__ __forceinline__ void rqs_inverse_pair(PT p, float x, const RqsConst& a, bool widths, float& y, float& ld,
                                                 int& bin) {
    static_assert(KT > 0, "bin count must be a template constant");
    constexpr int K = KT;
    const bool outside = a.tails && !(x >= a.left && x <= a.right);
    const float x_in = x;
    x = outside ? 0.5f * (a.left + a.right) : x;
    const PT u = p + (widths ? 0 : K * STRIDE);
    const PT ud = p + 2 * K * STRIDE;
    const float mn = widths ? a.min_w : a.min_h, sc = widths ? a.scale_w : a.scale_h;
    float e[K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        e[k] = u[k * STRIDE];
        m = fmaxf(m, e[k]);
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        e[k] = nf_exp(e[k] - m);
        s += e[k];
    }
    const float r = nf_rcp(s);
    double accd = 0.0;
    float kn[K + 1];
    kn[0] = a.left;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float w = mn + sc * (e[k] * r);
        accd += (double)w;
        kn[k + 1] = a.span * (float)accd + a.left;
    }
    kn[K] = a.right;
    // bin of x among this lane's knots (meaningful in the height lanes)
    int b = 0;
    float y0 = kn[0], y1 = kn[1];
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (x >= kn[k]) { b = k; y0 = kn[k]; y1 = kn[k + 1]; }
    // the width lanes pick the knots of their partner's bin and hand them over
    const int bpartner = nf_xor32(b);                // (unconditional: a lane exchange inside a divergent branch reads inactive lanes)
    const int bsel = widths ? bpartner : b;
    float x0 = kn[0], x1 = kn[1];
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (bsel == k) { x0 = kn[k]; x1 = kn[k + 1]; }
    x0 = nf_xor32(x0);
    x1 = nf_xor32(x1);

    float u0, u1;
    if (a.tails) {
        const float v0 = ud[(b == 0 ? 0 : b - 1) * STRIDE];
        const float v1 = ud[(b == K - 1 ? K - 2 : b) * STRIDE];
        u0 = b == 0 ? a.tail_const : v0;
        u1 = b == K - 1 ? a.tail_const : v1;
    } else {
        u0 = ud[b * STRIDE];
        u1 = ud[(b + 1) * STRIDE];
    }
    const float d0 = a.min_d + softplus_f(u0);
    const float d1 = a.min_d + softplus_f(u1);
    const float bw = x1 - x0;
    const float bh = y1 - y0;
    const float sl = bh / bw;
    const float curv = d0 + d1 - 2.0f * sl;
    const float dy = x - y0;
    const float qa = dy * curv + bh * (sl - d0);
    const float qb = bh * d0 - dy * curv;
    const float qc = -sl * dy;
    const float disc = qb * qb - 4.0f * qa * qc;
    const float t = (2.0f * qc) / (-qb - sqrtf(disc));
    y = t * bw + x0;
    const float t1mt = t * (1.0f - t);
    const float den = sl + curv * t1mt;
    const float omt = 1.0f - t;
    const float dnum = (sl * sl) * (d1 * (t * t) + 2.0f * sl * t1mt + d0 * (omt * omt));
    ld = -(nf_log(dnum) - 2.0f * nf_log(den));
    bin = b;
    if (outside) {
        y = x_in;
        ld = 0.0f;
        bin = -1;
    }
}


// ---- backward (forward direction of the spline): d(y, ld)/d(x, params) --------------------------------
// Recomputes the forward quantities from (x, params) instead of saving ~10 intermediates per element
// (what autograd keeps for the reference's op sequence), then overwrites the element's P parameters in
// `p` (LDS) with their gradients.  gy = dL/dy, gl = dL/d(logabsdet of this sample).  Returns dL/dx.
// inv: backward of the INVERSE direction, x_out = f^-1(y_in), lad = -log f'(x_out).  Called at x = x_out (the inverse
// pass's output) with gy = dL/dx_out and gl = dL/dlad; by implicit differentiation of y_in = f(x_out; params)
//   dL/dy_in  = a = (gy - gl * d(log f')/dx) / f'(x)
//   dL/dparams = -[ a * df/dparams + gl * d(log f')/dparams ]   = minus the forward-direction backward with (gy, gl) := (a, gl)
// so the same formulas run on the substituted upstream gradient and the parameter gradients change sign.  Returns a.
template <int KT>
__device__ __forceinline__ float rqs_bwd_eval(float* __restrict__ p, float x, float gy, float gl, const RqsConst& a,
                                              bool inv = false) {
    const int K = KT > 0 ? KT : a.K;
    const int P = a.P;
    if (a.tails && !(x >= a.left && x <= a.right)) {   // identity tail: y = x, ld = 0
        for (int j = 0; j < P; ++j) p[j] = 0.0f;
        return gy;
    }
    float* uw = p;
    float* uh = p + K;
    float* ud = p + 2 * K;
    const bool scaled = a.pre_div != 1.0f;
    auto raw = [&](const float* u, int k) {
        float v = u[k];
        return scaled ? nf_div(v, a.pre_div) : v;
    };
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        mw = fmaxf(mw, raw(uw, k));
        mh = fmaxf(mh, raw(uh, k));
    }
    constexpr int KA = KT > 0 ? KT : 1;
    float ew[KA], eh[KA];
    float sw = 0.0f, sh = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float w = nf_exp(raw(uw, k) - mw), h = nf_exp(raw(uh, k) - mh);
        if constexpr (KT > 0) { ew[k] = w; eh[k] = h; }
        sw += w;
        sh += h;
    }
    const float rw = nf_rcp(sw), rh = nf_rcp(sh);
    double accw = 0.0, acch = 0.0;
    float kw0 = a.left, kh0 = a.left;
    float x0 = a.left, x1 = a.left, y0 = a.left, y1 = a.left;
    int bin = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float w, h;
        if constexpr (KT > 0) { w = ew[k]; h = eh[k]; }
        else { w = nf_exp(raw(uw, k) - mw); h = nf_exp(raw(uh, k) - mh); }
        w = a.min_w + a.scale_w * (w * rw);
        h = a.min_h + a.scale_h * (h * rh);
        accw += (double)w;
        acch += (double)h;
        float kw1 = a.span * (float)accw + a.left;
        float kh1 = a.span * (float)acch + a.left;
        if (k == K - 1) { kw1 = a.right; kh1 = a.right; }
        if (x >= kw0) { bin = k; x0 = kw0; x1 = kw1; y0 = kh0; y1 = kh1; }
        kw0 = kw1;
        kh0 = kh1;
    }
    // slopes and which raw parameter (if any) feeds each
    int i0, i1;                      // index into ud, -1 = pinned boundary constant
    if (a.tails) { i0 = bin == 0 ? -1 : bin - 1; i1 = bin == K - 1 ? -1 : bin; }
    else { i0 = bin; i1 = bin + 1; }
    const float v0 = ud[i0 < 0 ? 0 : i0], v1 = ud[i1 < 0 ? 0 : i1];          // value selects, not address selects
    const float u0 = i0 < 0 ? a.tail_const : v0;
    const float u1 = i1 < 0 ? a.tail_const : v1;
    const float d0 = a.min_d + softplus_f(u0);
    const float d1 = a.min_d + softplus_f(u1);

    const float bw = x1 - x0, bh = y1 - y0;
    const float s = nf_div(bh, bw);
    const float t = nf_div(x - x0, bw);
    const float q = t * (1.0f - t), omt = 1.0f - t;
    const float c = d0 + d1 - 2.0f * s;
    const float N = bh * (s * t * t + d0 * q);
    const float Dn = s + c * q;
    const float M = d1 * t * t + 2.0f * s * q + d0 * omt * omt;

    const float rDn = nf_rcp(Dn), rbw = nf_rcp(bw);
    if (inv) {
        const float fp = (s * s) * M * (rDn * rDn);                                        // f'(x)
        const float Mt = 2.0f * d1 * t + 2.0f * s * (1.0f - 2.0f * t) - 2.0f * d0 * omt;   // dM/dt
        const float ldx = rbw * (nf_div(Mt, M) - 2.0f * c * (1.0f - 2.0f * t) * rDn);      // d(log f')/dx
        gy = nf_div(gy - gl * ldx, fp);
    }
    const float A = gy * rDn;                                  // dL/dN
    const float Bc = -gy * N * (rDn * rDn) - 2.0f * gl * rDn;  // dL/dDn
    const float Cm = nf_div(gl, M);                            // dL/dM
    const float g_t = A * bh * (2.0f * s * t + d0 * (1.0f - 2.0f * t)) + Bc * c * (1.0f - 2.0f * t) +
                      Cm * (2.0f * d1 * t + 2.0f * s * (1.0f - 2.0f * t) - 2.0f * d0 * omt);
    const float g_s = A * bh * t * t + Bc * (1.0f - 2.0f * q) + Cm * 2.0f * q + nf_div(2.0f * gl, s);
    float g_bh = A * (s * t * t + d0 * q) + g_s * rbw;
    const float g_d0 = A * bh * q + Bc * q + Cm * omt * omt;
    const float g_d1 = Bc * q + Cm * t * t;
    const float g_bw = -g_t * t * rbw - g_s * s * rbw;
    const float gx = g_t * rbw;
    const float g_x0 = -g_t * rbw - g_bw, g_x1 = g_bw;
    const float g_y0 = gy - g_bh, g_y1 = g_bh;

    // knots b and b+1 move with the widths / heights only when they are interior (ends are pinned)
    const float Gw0 = (bin >= 1) ? a.span * g_x0 : 0.0f, Gw1 = (bin + 1 <= K - 1) ? a.span * g_x1 : 0.0f;
    const float Gh0 = (bin >= 1) ? a.span * g_y0 : 0.0f, Gh1 = (bin + 1 <= K - 1) ? a.span * g_y1 : 0.0f;
    // d knot_k / d w_j = [j < k]:  g_w_j = Gw0 [j < bin] + Gw1 [j <= bin], through w = min + scale * softmax
    float dotw = 0.0f, doth = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float w, h;
        if constexpr (KT > 0) { w = ew[k]; h = eh[k]; }
        else { w = nf_exp(raw(uw, k) - mw); h = nf_exp(raw(uh, k) - mh); }
        const float gw = a.scale_w * ((k < bin ? Gw0 : 0.0f) + (k <= bin ? Gw1 : 0.0f));
        const float gh = a.scale_h * ((k < bin ? Gh0 : 0.0f) + (k <= bin ? Gh1 : 0.0f));
        dotw += (w * rw) * gw;
        doth += (h * rh) * gh;
    }
    const float sig0 = nf_rcp(1.0f + nf_exp(-u0)), sig1 = nf_rcp(1.0f + nf_exp(-u1));
    const int nd = P - 2 * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float w, h;
        if constexpr (KT > 0) { w = ew[k]; h = eh[k]; }
        else { w = nf_exp(raw(uw, k) - mw); h = nf_exp(raw(uh, k) - mh); }
        const float gw = a.scale_w * ((k < bin ? Gw0 : 0.0f) + (k <= bin ? Gw1 : 0.0f));
        const float gh = a.scale_h * ((k < bin ? Gh0 : 0.0f) + (k <= bin ? Gh1 : 0.0f));
        float ow = (w * rw) * (gw - dotw), oh = (h * rh) * (gh - doth);
        if (scaled) { ow = nf_div(ow, a.pre_div); oh = nf_div(oh, a.pre_div); }
        uw[k] = inv ? -ow : ow;
        uh[k] = inv ? -oh : oh;
    }
    for (int j = 0; j < nd; ++j) ud[j] = 0.0f;
    if (i0 >= 0) ud[i0] = inv ? -(g_d0 * sig0) : g_d0 * sig0;
    if (i1 >= 0) ud[i1] = inv ? -(g_d1 * sig1) : g_d1 * sig1;
    return inv ? gy : gx;
}
