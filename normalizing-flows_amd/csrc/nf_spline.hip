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
#include "nf_rqs_eval.h"
#include "nf_spline_families.h"

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
    int D;
    RqsConst c;
    int rows_per_block;
    int vec_ok;                // params base 16-byte aligned
};

// Staging of parameter spans (shared by the forward and backward kernels).  The block's parameters are one contiguous
// run of floats starting at gblk = row0 * D * P; every span of 256 elements starts 256 * P floats after the previous
// one, so all spans of a block sit at the same offset `shift` (0..3 floats) from a 16-byte boundary.  With a 16-byte
// aligned base the spans are fetched as 16-byte pieces through a buffer descriptor whose range check returns zeros
// past the end of the array (the last span may round up past it by < 4 floats) - no per-piece branches; an unaligned
// base (never what torch hands over, but legal at the C ABI) takes scalar loads straight into LDS.
// e / D for 0 <= e < 2^24 without the integer-division sequence: float estimate, corrected by one either way
__device__ __forceinline__ int div_small(int e, int D, float inv_d) {
    int r = (int)((float)e * inv_d);
    r -= (r * D > e) ? 1 : 0;
    r += ((r + 1) * D <= e) ? 1 : 0;
    return r;
}

struct SpanStage {
    __amdgpu_buffer_rsrc_t rsrc;
    const float* base;          // params + gblk (scalar path)
    int shift;                  // floats between the 16-byte boundary below gblk and gblk
    int vec;
};

__device__ __forceinline__ SpanStage span_stage_init(const float* params, int64_t n_param_floats, int64_t gblk, int vec_ok) {
    SpanStage st;
    const int64_t ablk = gblk & ~(int64_t)3;
    st.vec = vec_ok;
    st.shift = vec_ok ? (int)(gblk - ablk) : 0;
    st.base = params + gblk;
    const uint64_t remain = (uint64_t)(n_param_floats - ablk) * 4u;
#ifdef NF_KO_PARAM_LOADS        // timing-only build: an empty range drops every parameter fetch, the instruction stream stays
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(params + ablk), 0, remain > 0 ? 0u : 0u, 0x00020000);
#else
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(params + ablk), 0,
                                                remain > 0xffffffffull ? 0xffffffffu : (uint32_t)remain, 0x00020000);
#endif
    return st;
}

// pieces of the span that starts `rel` floats into the block (rel % 4 == 0) and holds nfl floats
template <int MAXV>
__device__ __forceinline__ void span_issue(const SpanStage& st, int rel, int nfl, int tid, float4 (&pre)[MAXV]) {
    const int nvec = (st.shift + nfl + 3) >> 2;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = tid + i * kThreads;
        const uint32_t off = v < nvec ? (uint32_t)(rel + 4 * v) * 4u : 0xfffffff0u;      // past the range: zeros, no fetch
        pre[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(st.rsrc, off, 0, 0));
    }
}

template <int MAXV>
__device__ __forceinline__ void span_commit(const SpanStage& st, int nfl, int tid, float* stage, const float4 (&pre)[MAXV]) {
    const int nvec = (st.shift + nfl + 3) >> 2;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = tid + i * kThreads;
        if (v < nvec) *reinterpret_cast<float4*>(stage + 4 * v) = pre[i];
    }
}

__device__ __forceinline__ void span_scalar(const SpanStage& st, int rel, int nfl, int tid, float* stage) {
    for (int f = tid; f < nfl; f += kThreads) stage[f] = st.base[rel + f];
}

// The same staging for parameters kept as bf16 (the bf16 training mode's fused forward, nf_made_rqs_fwd_params_bf16): a
// 16-byte piece holds 8 parameters and is widened on its way into the fp32 LDS stage.  shift = parameters between
// the 16-byte boundary below the block and the block (0..7); rel / nfl in parameters, rel % 8 == 0.
__device__ __forceinline__ SpanStage span_stage_init16(const __bf16* params, int64_t n_params, int64_t gblk) {
    SpanStage st;
    const int64_t ablk = gblk & ~(int64_t)7;
    st.vec = 1;
    st.shift = (int)(gblk - ablk);
    st.base = nullptr;
    const uint64_t remain = (uint64_t)(n_params - ablk) * 2u;
    st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(params + ablk), 0,
                                                remain > 0xffffffffull ? 0xffffffffu : (uint32_t)remain, 0x00020000);
    return st;
}

template <int MAXV>
__device__ __forceinline__ void span_issue16(const SpanStage& st, int rel, int nfl, int tid, float4 (&pre)[MAXV]) {
    const int nvec = (st.shift + nfl + 7) >> 3;
#pragma unroll
    for (int i = 0; i < (MAXV + 1) / 2; ++i) {
        const int v = tid + i * kThreads;
        const uint32_t off = v < nvec ? (uint32_t)(rel + 8 * v) * 2u : 0xfffffff0u;
        pre[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(st.rsrc, off, 0, 0));
    }
}

template <int MAXV>
__device__ __forceinline__ void span_commit16(const SpanStage& st, int nfl, int tid, float* stage, const float4 (&pre)[MAXV]) {
    const int nvec = (st.shift + nfl + 7) >> 3;
#pragma unroll
    for (int i = 0; i < (MAXV + 1) / 2; ++i) {
        const int v = tid + i * kThreads;
        if (v < nvec) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 w = __builtin_bit_cast(u4, pre[i]);                       // bf16 -> fp32: the 16 bits moved up
            *reinterpret_cast<float4*>(stage + 8 * v) =
                make_float4(__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u),
                            __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u));
            *reinterpret_cast<float4*>(stage + 8 * v + 4) =
                make_float4(__uint_as_float(w[2] << 16), __uint_as_float(w[2] & 0xffff0000u),
                            __uint_as_float(w[3] << 16), __uint_as_float(w[3] & 0xffff0000u));
        }
    }
}

// FAM: 0 rational-quadratic, 1 linear, 2 quadratic, 3 cubic spline (nf_spline_families.h); the staging is the same
template <int KT, bool INV, int MAXV, int FAM = 0>
__global__ __launch_bounds__(kThreads) void rqs_kernel(const RqsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int P = a.c.P;
    const int D = a.D;
    const int64_t row0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int nrows = (int)min((int64_t)a.rows_per_block, a.B - row0);
    const int E = nrows * D;
    const int stage_floats = (kThreads * P + 4 + 3) & ~3;
    float* stage = smem;
    float* ldbuf = smem + stage_floats;
    const int64_t e0 = row0 * D;                          // first element of the block
    const SpanStage st = span_stage_init(a.params, a.n_param_floats, e0 * P, a.vec_ok);
    // x / y rows back to back and no column index: element e of the block is float e0 + e of both (no division)
    const bool flat = !a.feat && a.x_stride == D && a.y_stride == D;
    const float inv_d = 1.0f / (float)D;

    // Two spans of parameters are in flight (two register sets) while a third is evaluated: one block keeps up to
    // 2 * 256 * P * 4 bytes outstanding, which is what it takes to keep HBM busy at 4 blocks per CU.
    float4 preA[MAXV], preB[MAXV];
    auto issue = [&](int it0, float4 (&pre)[MAXV]) {
        if (st.vec && it0 < E) span_issue<MAXV>(st, it0 * P, min(kThreads, E - it0) * P, tid, pre);
    };
    // element e of the block -> its offsets in x and y
    auto locate = [&](int e, int64_t& xi, int64_t& yi) {
        if (flat) {
            xi = yi = e0 + e;
        } else {
            const int r = div_small(e, D, inv_d);
            const int d = e - r * D;
            const int col = a.feat ? a.feat[d] : d;
            xi = (row0 + r) * a.x_stride + col;
            yi = (row0 + r) * a.y_stride + col;
        }
    };
    // x of the NEXT span is fetched one step ahead: a load issued after the parameter prefetch would have to wait for
    // it (loads return in order), which is a full memory latency per step with nothing to hide it
    float x_ahead = 0.0f;
    {
        int64_t xi, yi;
        if (tid < E) { locate(tid, xi, yi); x_ahead = a.x[xi]; }
    }
    auto step = [&](int it0, float4 (&pre)[MAXV]) {
        const int n = min(kThreads, E - it0);
        __syncthreads();               // previous span fully consumed
        if (st.vec) span_commit<MAXV>(st, n * P, tid, stage, pre);
        else span_scalar(st, it0 * P, n * P, tid, stage);
        __syncthreads();
        const float xv = x_ahead;
        if (it0 + kThreads + tid < E) {
            int64_t xi, yi;
            locate(it0 + kThreads + tid, xi, yi);
            x_ahead = a.x[xi];
        }
        issue(it0 + 2 * kThreads, pre);                   // in flight while this span and the next are evaluated
        if (tid < n) {
            const int e = it0 + tid;
            int64_t xi, yi;
            locate(e, xi, yi);
            typedef __attribute__((address_space(3))) float* lds_w;          // explicit LDS pointers: ds_ ops, never flat
            typedef const __attribute__((address_space(3))) float* lds_r;
            lds_w ps = (lds_w)(stage + st.shift + tid * P);
            float yv, ld;
            int bin;
            if constexpr (FAM == 1 && KT > 0) {
                // compile-time K: the element's K logits move to registers once (every index in lin_eval is static
                // after unrolling), so its cached values never go back to LDS (P = 8: a 16-way bank conflict per access)
                float pv[KT];
                if (KT % 4 == 0 && st.shift == 0) {                   // P = K floats per lane, 16-byte aligned
#pragma unroll
                    for (int q = 0; q < KT / 4; ++q) {
                        typedef float f4 __attribute__((ext_vector_type(4)));
                        const f4 v4 = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(ps + 4 * q);
                        pv[4 * q] = v4[0]; pv[4 * q + 1] = v4[1]; pv[4 * q + 2] = v4[2]; pv[4 * q + 3] = v4[3];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < KT; ++q) pv[q] = ps[q];
                }
                lin_eval<INV, float*, true, KT>(pv, xv, a.c, yv, ld, bin);
            } else if (FAM == 1) lin_eval<INV, lds_w, true, KT>(ps, xv, a.c, yv, ld, bin);
            else if (FAM == 2) quad_eval<INV, lds_w, true, KT>(ps, xv, a.c, yv, ld, bin);
            else if (FAM == 3 && a.c.pre_div != 1.0f) cubic_eval<INV, lds_w, true, KT, 1>(ps, xv, a.c, yv, ld, bin);
            else if (FAM == 3) cubic_eval<INV, lds_w, true, KT, 0>(ps, xv, a.c, yv, ld, bin);
            else if (a.c.pre_div != 1.0f) rqs_eval<KT, INV, 1, false, false, lds_r>((lds_r)ps, xv, a.c, yv, ld, bin);
            else rqs_eval<KT, INV, 1, false, false, lds_r, false>((lds_r)ps, xv, a.c, yv, ld, bin);
            a.y[yi] = yv;
            if (a.bins) a.bins[e0 + e] = bin;
            ldbuf[e] = ld;
        }
    };
    issue(0, preA);
    issue(kThreads, preB);
    for (int it0 = 0; it0 < E; it0 += 2 * kThreads) {
        step(it0, preA);
        if (it0 + kThreads < E) step(it0 + kThreads, preB);
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

// ---- backward of the forward-direction spline (training by maximum likelihood) ------------------------
struct RqsBwdArgs {
    const float* x;
    int64_t x_stride;
    const float* params;       // [B, D*P] contiguous (the forward's conditioner output)
    int64_t n_param_floats;
    const float* gy;           // [B, D] dL/dy
    int64_t gy_stride;
    const float* glad;         // [B]    dL/dlogabsdet
    float* gx;                 // [B, D] dL/dx through the spline only (the conditioner adds its own part)
    int64_t gx_stride;
    float* gparams;            // [B, D*P] contiguous
    int64_t B;
    int D;
    RqsConst c;
    int rows_per_block;
    int vec_ok;                // params AND gparams 16-byte aligned
    int inverse;               // backward of the inverse direction (x = the inverse pass's OUTPUT, see rqs_bwd_eval)
    unsigned* gmax;            // nullable: atomicMax of the bit pattern of the largest finite |grad_params| written
    __bf16* gparams16;         // nullable: grad_params go HERE as bf16 [B, D*P] instead of to gparams (the bf16 training
                               // mode's backward GEMMs round them to 8 bits anyway: half the bytes written and re-read)
    const __bf16* params16;    // PB16 kernels: the parameters as bf16 [B, D*P] (16-byte aligned; gparams16 required)
};

// FAM 0: rational-quadratic (gradients overwrite the staged parameters in place); 1 / 2 / 3: linear / quadratic /
// cubic family (nf_spline_families.h), whose reverse sweeps keep the parameters and write a second LDS region.
// The next span's parameters are in flight (registers) while this one is evaluated and written back.
// (bf16-parameter flavour: 136 registers uncapped = three waves per SIMD; it fits 128 without a spill = four)
template <int KT, int MAXV, int FAM = 0, bool PB16 = false>
__global__ __launch_bounds__(kThreads, (FAM == 0 && PB16) ? 4 : 1) void rqs_bwd_kernel(const RqsBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int P = a.c.P;
    const int D = a.D;
    const int64_t row0 = (int64_t)blockIdx.x * a.rows_per_block;
    const int nrows = (int)min((int64_t)a.rows_per_block, a.B - row0);
    const int E = nrows * D;
    float* stage = smem;
    constexpr bool kInPlace = FAM == 0 || KT > 0;                    // the evaluator overwrites the staged parameters
    float* gout = kInPlace ? smem : smem + ((kThreads * P + 4 + 3) & ~3);
    const int64_t e0 = row0 * D;
    const int64_t gblk = e0 * P;
    const SpanStage st = PB16 ? span_stage_init16(a.params16, a.n_param_floats, gblk)
                              : span_stage_init(a.params, a.n_param_floats, gblk, a.vec_ok);
    const bool flat = a.x_stride == D && a.gy_stride == D && a.gx_stride == D;
    const float inv_d = 1.0f / (float)D;
    float* gbase = a.gparams + (gblk - st.shift);          // 16-byte aligned when st.vec
    __bf16* gbase16 = a.gparams16 ? a.gparams16 + (gblk - st.shift) : nullptr;     // (then 8-byte aligned)

    auto locate = [&](int e, int64_t& xi, int64_t& gyi, int64_t& gxi, int& r) {
        r = div_small(e, D, inv_d);
        if (flat) {
            xi = gyi = gxi = e0 + e;
        } else {
            const int d = e - r * D;
            xi = (row0 + r) * a.x_stride + d;
            gyi = (row0 + r) * a.gy_stride + d;
            gxi = (row0 + r) * a.gx_stride + d;
        }
    };
    // per-element inputs of the NEXT span are fetched one step ahead (see rqs_kernel)
    float x_ahead = 0.0f, gy_ahead = 0.0f, gl_ahead = 0.0f;
    auto fetch_ahead = [&](int e) {
        if (e < E) {
            int64_t xi, gyi, gxi;
            int r;
            locate(e, xi, gyi, gxi, r);
            x_ahead = a.x[xi];
            gy_ahead = a.gy[gyi];
            gl_ahead = a.glad[row0 + r];
        }
    };
    fetch_ahead(tid);
    float gm = 0.0f;                                       // largest finite |gradient| this lane wrote
    auto seen = [&](float v) {
        const float av = fabsf(v);
        gm = fmaxf(gm, av < INFINITY ? av : 0.0f);
    };
    float4 pre[MAXV];
    if (PB16) { if (E > 0) span_issue16<MAXV>(st, 0, min(kThreads, E) * P, tid, pre); }
    else if (st.vec && E > 0) span_issue<MAXV>(st, 0, min(kThreads, E) * P, tid, pre);
    for (int it0 = 0; it0 < E; it0 += kThreads) {
        const int n = min(kThreads, E - it0);
        __syncthreads();
        if (PB16) span_commit16<MAXV>(st, n * P, tid, stage, pre);
        else if (st.vec) span_commit<MAXV>(st, n * P, tid, stage, pre);
        else span_scalar(st, it0 * P, n * P, tid, stage);
        __syncthreads();
        const float xv = x_ahead, gyv = gy_ahead, gl = gl_ahead;
        fetch_ahead(it0 + kThreads + tid);
        if (PB16) {
            if (it0 + kThreads < E) span_issue16<MAXV>(st, (it0 + kThreads) * P, min(kThreads, E - it0 - kThreads) * P, tid, pre);
        } else if (st.vec && it0 + kThreads < E)
            span_issue<MAXV>(st, (it0 + kThreads) * P, min(kThreads, E - it0 - kThreads) * P, tid, pre);
        if (tid < n) {
            const int e = it0 + tid;
            int64_t xi, gyi, gxi;
            int r;
            locate(e, xi, gyi, gxi, r);
            float* ps = stage + st.shift + tid * P;
            float* gs = gout + st.shift + tid * P;
            float gxv;
            if (FAM == 1) {
                if constexpr (KT > 0) gxv = lin_bwd_eval_t<KT>(ps, xv, gyv, gl, a.c, a.inverse != 0);
                else gxv = lin_bwd_eval(ps, gs, xv, gyv, gl, a.c, a.inverse != 0);
            } else if (FAM == 2) {
                if constexpr (KT > 0) gxv = quad_bwd_eval_t<KT>(ps, xv, gyv, gl, a.c, a.inverse != 0);
                else gxv = quad_bwd_eval(ps, gs, xv, gyv, gl, a.c, a.inverse != 0);
            } else if (FAM == 3) {
                if constexpr (KT > 0) gxv = cubic_bwd_eval_t<KT>(ps, xv, gyv, gl, a.c, a.inverse != 0);
                else gxv = cubic_bwd_eval(ps, gs, xv, gyv, gl, a.c, a.inverse != 0);
            }
            else gxv = rqs_bwd_eval<KT>(ps, xv, gyv, gl, a.c, a.inverse != 0);
            a.gx[gxi] = gxv;
        }
        __syncthreads();
        // write the gradients of this span back: floats [lo, hi) of the staged pieces are this span's; whole 16-byte
        // pieces inside go out as such, the two pieces that straddle a neighbouring span float by float
        const int lo = st.shift, hi = st.shift + n * P;
        const int nvec = (hi + 3) >> 2;
        float* gspan = gbase + (int64_t)it0 * P;
        __bf16* gspan16 = gbase16 + (int64_t)it0 * P;
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = tid + i * kThreads;
            if (v >= nvec) continue;
            const int f = 4 * v;
            if (st.vec && f >= lo && f + 3 < hi) {
                const float4 gv = *reinterpret_cast<const float4*>(gout + f);
                if (gbase16) {
                    const bf16x4_t hv = {(__bf16)gv.x, (__bf16)gv.y, (__bf16)gv.z, (__bf16)gv.w};
                    *reinterpret_cast<bf16x4_t*>(gspan16 + f) = hv;
                } else {
                    *reinterpret_cast<float4*>(gspan + f) = gv;
                }
                seen(gv.x); seen(gv.y); seen(gv.z); seen(gv.w);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (f + q >= lo && f + q < hi) {
                        if (gbase16) gspan16[f + q] = (__bf16)gout[f + q];
                        else gspan[f + q] = gout[f + q];
                        seen(gout[f + q]);
                    }
            }
        }
    }
    if (a.gmax) {                                          // one atomic per workgroup (uniform branch)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gm = fmaxf(gm, __shfl_xor(gm, off, 64));
        __syncthreads();
        if ((tid & 63) == 0) smem[tid >> 6] = gm;
        __syncthreads();
        if (tid == 0) {
            gm = fmaxf(fmaxf(smem[0], smem[1]), fmaxf(smem[2], smem[3]));
            if (gm > 0.0f) atomicMax(a.gmax, __float_as_uint(gm));
        }
    }
}

// Launch of one instantiation; more than 64 KB of dynamic LDS (many bins) needs the per-device cap raised first.
template <auto Kern, class Args>
int launch_spline(const Args& a, int grid, size_t lds, hipStream_t s) {
    static unsigned long long raised = 0;
    if (lds > 64 * 1024) {
        if (const int e = nf_raise_lds_cap((const void*)Kern, &raised)) return e;
    }
    hipLaunchKernelGGL(Kern, dim3(grid), dim3(kThreads), lds, s, a);
    return NF_OK;
}

template <int KT>
int launch_rqs_bwd_k(const RqsBwdArgs& a, int grid, size_t lds, hipStream_t s) {
    const int nv = (kThreads * a.c.P + 4 + 3) / 4;
    const int per = (nv + kThreads - 1) / kThreads;
    if (a.params16) {
        if (per <= 6) { if (const int e = launch_spline<rqs_bwd_kernel<KT, 6, 0, true>>(a, grid, lds, s)) return e; }
        else if (per <= 12) { if (const int e = launch_spline<rqs_bwd_kernel<KT, 12, 0, true>>(a, grid, lds, s)) return e; }
        else return NF_ERR_SHAPE;
        return nf_launch_status();
    }
    if (per <= 6) { if (const int e = launch_spline<rqs_bwd_kernel<KT, 6>>(a, grid, lds, s)) return e; }
    else if (per <= 12) { if (const int e = launch_spline<rqs_bwd_kernel<KT, 12>>(a, grid, lds, s)) return e; }
    else if (per <= 25) { if (const int e = launch_spline<rqs_bwd_kernel<KT, 25>>(a, grid, lds, s)) return e; }
    else return NF_ERR_SHAPE;
    return nf_launch_status();
}

template <int FAM>
int launch_family_bwd(const RqsBwdArgs& a, int grid, size_t lds, hipStream_t s) {
    const int nv = (kThreads * a.c.P + 4 + 3) / 4;
    const int per = (nv + kThreads - 1) / kThreads;
    if ((a.c.K == 8 || a.c.K == 10) && per <= 6) {                // bin count as a template constant, gradients in place
        const size_t lds1 = lds / 2;
        if (a.c.K == 8) hipLaunchKernelGGL((rqs_bwd_kernel<8, 6, FAM>), dim3(grid), dim3(kThreads), lds1, s, a);
        else hipLaunchKernelGGL((rqs_bwd_kernel<10, 6, FAM>), dim3(grid), dim3(kThreads), lds1, s, a);
        return nf_launch_status();
    }
    static unsigned long long raised[3] = {0, 0, 0};
#define NF_FAM_BWD(MAXV)                                                                              \
    do {                                                                                               \
        if (int e = nf_raise_lds_cap((const void*)rqs_bwd_kernel<0, MAXV, FAM>, &raised[MAXV == 6 ? 0 : (MAXV == 12 ? 1 : 2)])) return e; \
        hipLaunchKernelGGL((rqs_bwd_kernel<0, MAXV, FAM>), dim3(grid), dim3(kThreads), lds, s, a);     \
    } while (0)
    if (per <= 6) NF_FAM_BWD(6);
    else if (per <= 12) NF_FAM_BWD(12);
    else if (per <= 25) NF_FAM_BWD(25);
    else return NF_ERR_SHAPE;
#undef NF_FAM_BWD
    return nf_launch_status();
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
    const float inv_d = 1.0f / (float)D;
    // rows back to back, interleaved parameters, no column index: element e of the block is float e0 + e of x and y
    // and float2 e0 + e of the parameters - no division, every access a straight stream
    const bool flat = !a.feat && a.x_stride == D && a.y_stride == D && a.layout == NF_AFFINE_INTERLEAVED &&
                      a.p_stride == 2 * (int64_t)D;
    const int64_t e0 = row0 * D;
    // one element: scale = sigmoid(u + 2) + 1e-3, y = scale * x + shift (or its inverse), log scale for the row sum
    auto one = [&](float us, float shift, float xv, float& yv, float& ls) {
        const float z = us + 2.0f;
        const float scale = 1.0f / (1.0f + expf(-z)) + 1e-3f;
        const float l = logf(scale);
        yv = INV ? (xv - shift) / scale : scale * xv + shift;
        ls = INV ? -l : l;
    };
    // Four consecutive elements of a row per lane and step, every access 16 bytes: no column index, D a multiple of 4,
    // every row 16-byte addressable.  Same operations per element as the loop below (same bits).
    // (Blocked parameters only: with interleaved ones a lane's 8 parameters are 32 bytes, two half-used 16-byte
    // accesses per lane measured 25 % SLOWER than the flat loop below; they take two elements per lane instead - one
    // 16-byte parameter access, 8 bytes of x and y.)
    const bool al16 = !a.feat && !((((uintptr_t)a.x) | ((uintptr_t)a.y) | ((uintptr_t)a.params)) & 15);
    const bool vec4 = al16 && a.layout == NF_AFFINE_BLOCKED && !(D & 3) && !((a.x_stride | a.y_stride | a.p_stride) & 3);
    const bool vec2 = al16 && a.layout == NF_AFFINE_INTERLEAVED && !(D & 1) && !((a.x_stride | a.y_stride) & 1) &&
                      !(a.p_stride & 3);
    if (vec2) {
        const int D2 = D >> 1, Q = nrows * D2;
        const float inv_d2 = 1.0f / (float)D2;
        for (int q = tid; q < Q; q += kThreads) {
            const int r = div_small(q, D2, inv_d2);
            const int d = (q - r * D2) << 1;
            const float4 pv = *reinterpret_cast<const float4*>(a.params + (row0 + r) * a.p_stride + 2 * d);
            const float2 xv = *reinterpret_cast<const float2*>(a.x + (row0 + r) * a.x_stride + d);
            float2 yv, ls;
            one(pv.x, pv.y, xv.x, yv.x, ls.x);
            one(pv.z, pv.w, xv.y, yv.y, ls.y);
            *reinterpret_cast<float2*>(a.y + (row0 + r) * a.y_stride + d) = yv;
            *reinterpret_cast<float2*>(smem + r * D + d) = ls;
        }
    } else if (vec4) {
        const int D4 = D >> 2, Q = nrows * D4;
        const float inv_d4 = 1.0f / (float)D4;
        for (int q = tid; q < Q; q += kThreads) {
            const int r = div_small(q, D4, inv_d4);
            const int d = (q - r * D4) << 2;
            const float* prow = a.params + (row0 + r) * a.p_stride;
            const float4 xv = *reinterpret_cast<const float4*>(a.x + (row0 + r) * a.x_stride + d);
            const float4 sh = *reinterpret_cast<const float4*>(prow + d);
            const float4 us = *reinterpret_cast<const float4*>(prow + D + d);
            float4 yv, ls;
            one(us.x, sh.x, xv.x, yv.x, ls.x);
            one(us.y, sh.y, xv.y, yv.y, ls.y);
            one(us.z, sh.z, xv.z, yv.z, ls.z);
            one(us.w, sh.w, xv.w, yv.w, ls.w);
            *reinterpret_cast<float4*>(a.y + (row0 + r) * a.y_stride + d) = yv;
            *reinterpret_cast<float4*>(smem + r * D + d) = ls;
        }
    } else
    for (int e = tid; e < E; e += kThreads) {
        float us, shift;
        int64_t xi, yi;
        if (flat) {
            const float2 v = reinterpret_cast<const float2*>(a.params)[e0 + e];
            us = v.x;
            shift = v.y;
            xi = yi = e0 + e;
        } else {
            const int r = div_small(e, D, inv_d);
            const int d = e - r * D;
            const int col = a.feat ? a.feat[d] : d;
            const float* prow = a.params + (row0 + r) * a.p_stride;
            if (a.layout == NF_AFFINE_INTERLEAVED) {
                const float2 v = *reinterpret_cast<const float2*>(prow + 2 * d);
                us = v.x;
                shift = v.y;
            } else {
                shift = prow[d];
                us = prow[D + d];
            }
            xi = (row0 + r) * a.x_stride + col;
            yi = (row0 + r) * a.y_stride + col;
        }
        float yv, ls;
        one(us, shift, a.x[xi], yv, ls);
        a.y[yi] = yv;
        smem[e] = ls;
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

// Backward of affine_kernel: from x, the parameters (scale recomputed), grad_y and grad_logabsdet to grad_x and
// grad_params (same layout as params).  forward: y = s x + t, lad = sum log s;  inverse: y = (x - t) / s,
// lad = -sum log s;  s = sigmoid(u + 2) + 1e-3, ds/du = sig (1 - sig).  Pure elementwise: one thread per element.
struct AffineBwdArgs {
    const float* x;
    int64_t x_stride;
    const int32_t* feat;
    const float* params;
    int64_t p_stride;
    const float* gy;
    int64_t gy_stride;
    const float* glad;
    float* gx;
    int64_t gx_stride;
    float* gp;
    int64_t gp_stride;
    int64_t B;
    int D;
    int layout;
};

template <bool INV>
__global__ __launch_bounds__(kThreads) void affine_bwd_kernel(const AffineBwdArgs a) {
    const int D = a.D;
    const int64_t total = a.B * D;
    for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
        const int64_t r = e / D;
        const int d = (int)(e - r * D);
        const int col = a.feat ? a.feat[d] : d;
        const float* prow = a.params + r * a.p_stride;
        float us, shift;
        if (a.layout == NF_AFFINE_INTERLEAVED) {
            const float2 v = *reinterpret_cast<const float2*>(prow + 2 * d);
            us = v.x;
            shift = v.y;
        } else {
            shift = prow[d];
            us = prow[D + d];
        }
        const float sig = 1.0f / (1.0f + expf(-(us + 2.0f)));
        const float scale = sig + 1e-3f;
        const float xv = a.x[r * a.x_stride + col];
        const float gy = a.gy[r * a.gy_stride + col];
        const float gl = a.glad ? a.glad[r] : 0.0f;
        float gx, gshift, gscale;
        if (INV) {
            const float inv = 1.0f / scale;
            const float y = (xv - shift) * inv;
            gx = gy * inv;
            gshift = -gx;
            gscale = -(gy * y + gl) * inv;
        } else {
            gx = gy * scale;
            gshift = gy;
            gscale = gy * xv + gl / scale;
        }
        const float gu = gscale * sig * (1.0f - sig);
        a.gx[r * a.gx_stride + col] = gx;
        float* grow = a.gp + r * a.gp_stride;
        if (a.layout == NF_AFFINE_INTERLEAVED) {
            *reinterpret_cast<float2*>(grow + 2 * d) = make_float2(gu, gshift);
        } else {
            grow[d] = gshift;
            grow[D + d] = gu;
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
    const int nv = (kThreads * a.c.P + 4 + 3) / 4;
    const int per = (nv + kThreads - 1) / kThreads;
    if (per <= 6) { if (const int e = launch_spline<rqs_kernel<KT, INV, 6>>(a, grid, lds, s)) return e; }
    else if (per <= 12) { if (const int e = launch_spline<rqs_kernel<KT, INV, 12>>(a, grid, lds, s)) return e; }
    else if (per <= 25) { if (const int e = launch_spline<rqs_kernel<KT, INV, 25>>(a, grid, lds, s)) return e; }
    else return NF_ERR_SHAPE;
    return nf_launch_status();
}

template <bool INV>
int launch_rqs(const RqsArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (a.c.K) {
        case 4: return launch_rqs_k<4, INV>(a, grid, lds, s);
        case 8: return launch_rqs_k<8, INV>(a, grid, lds, s);
        case 10: return launch_rqs_k<10, INV>(a, grid, lds, s);
        case 16: return launch_rqs_k<16, INV>(a, grid, lds, s);
        default: return launch_rqs_k<0, INV>(a, grid, lds, s);
    }
}

template <bool INV, int FAM>
int launch_family(const RqsArgs& a, int grid, size_t lds, hipStream_t s) {
    const int nv = (kThreads * a.c.P + 4 + 3) / 4;
    const int per = (nv + kThreads - 1) / kThreads;
    if (per <= 6 && a.c.K == 8) { if (const int e = launch_spline<rqs_kernel<8, INV, 6, FAM>>(a, grid, lds, s)) return e; }
    else if (per <= 6 && a.c.K == 10) { if (const int e = launch_spline<rqs_kernel<10, INV, 6, FAM>>(a, grid, lds, s)) return e; }
    else if (per <= 6) { if (const int e = launch_spline<rqs_kernel<0, INV, 6, FAM>>(a, grid, lds, s)) return e; }
    else if (per <= 12) { if (const int e = launch_spline<rqs_kernel<0, INV, 12, FAM>>(a, grid, lds, s)) return e; }
    else if (per <= 25) { if (const int e = launch_spline<rqs_kernel<0, INV, 25, FAM>>(a, grid, lds, s)) return e; }
    else return NF_ERR_SHAPE;
    return nf_launch_status();
}

// shared body of the linear / quadratic spline entry points: P parameters per element
template <int FAM>
int family_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index, const float* params,
               int64_t param_row_stride, float* y, int64_t y_row_stride, float* logabsdet, int accumulate,
               int32_t* bin_idx, int64_t B, int D, int K, int P, int tails, double tail_bound, double min_w,
               double min_h, int inverse, void* stream, double pre_div = 1.0, bool symmetric = false) {
    if (B == 0) return NF_OK;
    if (!x || !params || !y || !logabsdet) return NF_ERR_NULL;
    if (param_row_stride != (int64_t)D * P) return NF_ERR_LAYOUT;
    if (x_row_stride < D || y_row_stride < D) return NF_ERR_LAYOUT;
    RqsArgs a;
    a.x = x; a.x_stride = x_row_stride; a.feat = feat_index;
    a.params = params; a.n_param_floats = B * (int64_t)D * P;
    a.y = y; a.y_stride = y_row_stride; a.lad = logabsdet; a.accumulate = accumulate; a.bins = bin_idx;
    a.B = B; a.D = D;
    a.c = nf_make_rqs_const(K, tails, tail_bound, min_w, min_h, 1e-3, pre_div);
    a.c.P = P;
    if (symmetric) {                       // the cubic spline lives on [-bound, bound] with or without tails
        a.c.left = (float)-tail_bound;
        a.c.right = (float)tail_bound;
        a.c.span = (float)(2.0 * tail_bound);
    }
    a.rows_per_block = pick_rows_per_block(D, B);
    a.vec_ok = ((uintptr_t)params & 15) == 0;
    const int stage_floats = (kThreads * P + 4 + 3) & ~3;
    size_t lds = sizeof(float) * ((size_t)stage_floats + (size_t)a.rows_per_block * D);
    if (lds > 160 * 1024) return NF_ERR_SHAPE;
    const int64_t grid = (B + a.rows_per_block - 1) / a.rows_per_block;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    return inverse ? launch_family<true, FAM>(a, (int)grid, lds, s) : launch_family<false, FAM>(a, (int)grid, lds, s);
}

}  // namespace

extern "C" int nf_linear_spline_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index,
                                    const float* params, int64_t param_row_stride, float* y, int64_t y_row_stride,
                                    float* logabsdet, int accumulate, int32_t* bin_idx, int64_t B, int D, int K,
                                    int tails, double tail_bound, int inverse, void* stream) {
    if (B < 0 || D <= 0 || K <= 0 || K > 64) return NF_ERR_SHAPE;
    return family_fwd<1>(x, x_row_stride, feat_index, params, param_row_stride, y, y_row_stride, logabsdet, accumulate,
                         bin_idx, B, D, K, K, tails, tail_bound, 0.0, 0.0, inverse, stream);
}

extern "C" int nf_quadratic_spline_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index,
                                       const float* params, int64_t param_row_stride, float* y, int64_t y_row_stride,
                                       float* logabsdet, int accumulate, int32_t* bin_idx, int64_t B, int D, int K,
                                       int tails, double tail_bound, double min_bin_width, double min_bin_height,
                                       int inverse, void* stream) {
    if (B < 0 || D <= 0 || K < 2 || K > 32) return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    return family_fwd<2>(x, x_row_stride, feat_index, params, param_row_stride, y, y_row_stride, logabsdet, accumulate,
                         bin_idx, B, D, K, tails ? 2 * K - 1 : 2 * K + 1, tails, tail_bound, min_bin_width,
                         min_bin_height, inverse, stream);
}

extern "C" int nf_cubic_spline_fwd(const float* x, int64_t x_row_stride, const int32_t* feat_index, const float* params,
                                   int64_t param_row_stride, double pre_div, float* y, int64_t y_row_stride,
                                   float* logabsdet, int accumulate, int32_t* bin_idx, int64_t B, int D, int K,
                                   int tails, double bound, double min_bin_width, double min_bin_height, int inverse,
                                   void* stream) {
    if (B < 0 || D <= 0 || K <= 0 || K > 32 || !(bound > 0.0) || !(pre_div > 0.0)) return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    return family_fwd<3>(x, x_row_stride, feat_index, params, param_row_stride, y, y_row_stride, logabsdet, accumulate,
                         bin_idx, B, D, K, 2 * K + 2, tails, bound, min_bin_width, min_bin_height, inverse, stream,
                         pre_div, true);
}

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
    a.B = B; a.D = D;
    a.c = nf_make_rqs_const(K, tails, tail_bound, min_bin_width, min_bin_height, min_derivative, pre_div);
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

// out[b] = lad_scale * logabsdet[b] + sum_d (-z[b,d]^2 / 2 - log(2 pi) / 2): the N(0, I) base density joined with the
// flow's log|det J| in one pass (reference: distributions.py:105-107 summed per sample, + the `+ logabsdet` of
// distributions.py:154-157 / the `- logabsdet` of :173-186).  One wave per row, lanes stride the row in 16-byte
// pieces when they can, fixed-order shuffle reduction: deterministic.
__global__ __launch_bounds__(256) void std_normal_logp_kernel(const float* __restrict__ z, int64_t ldz, int64_t B, int D,
                                                              const float* __restrict__ lad, float lad_scale,
                                                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* p = z + row * ldz;
    float s = 0.f;
    if (D % 4 == 0 && ldz % 4 == 0 && ((uintptr_t)z & 15) == 0) {
        for (int c = lane * 4; c < D; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + c);
            s += (-0.5f * (v.x * v.x) - 0.918938533204672742f) + (-0.5f * (v.y * v.y) - 0.918938533204672742f);
            s += (-0.5f * (v.z * v.z) - 0.918938533204672742f) + (-0.5f * (v.w * v.w) - 0.918938533204672742f);
        }
    } else {
        for (int c = lane; c < D; c += 64) s += -0.5f * (p[c] * p[c]) - 0.918938533204672742f;
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = lad ? lad_scale * lad[row] + s : s;
}

extern "C" int nf_std_normal_log_prob(const float* z, int64_t z_row_stride, const float* logabsdet, double lad_scale,
                                      float* out, int64_t B, int D, void* stream) {
    if (B < 0 || D <= 0) return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!z || !out) return NF_ERR_NULL;
    if (z_row_stride < D) return NF_ERR_LAYOUT;
    const int64_t grid = (B + 3) / 4;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    hipLaunchKernelGGL(std_normal_logp_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, z, z_row_stride, B,
                       D, logabsdet, (float)lad_scale, out);
    return nf_launch_status();
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

extern "C" int nf_affine_bwd(const float* x, int64_t x_row_stride, const int32_t* feat_index, const float* params,
                             int64_t param_row_stride, int layout, const float* grad_y, int64_t grad_y_row_stride,
                             const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride,
                             float* grad_params, int64_t grad_param_row_stride, int64_t B, int D, int inverse,
                             void* stream) {
    if (B < 0 || D <= 0) return NF_ERR_SHAPE;
    if (layout != NF_AFFINE_INTERLEAVED && layout != NF_AFFINE_BLOCKED) return NF_ERR_DTYPE;
    if (B == 0) return NF_OK;
    if (!x || !params || !grad_y || !grad_x || !grad_params) return NF_ERR_NULL;
    if (param_row_stride < 2 * (int64_t)D || grad_param_row_stride < 2 * (int64_t)D || x_row_stride < D ||
        grad_y_row_stride < D || grad_x_row_stride < D)
        return NF_ERR_LAYOUT;
    if (layout == NF_AFFINE_INTERLEAVED &&
        ((((uintptr_t)params) & 7) || (param_row_stride & 1) || (((uintptr_t)grad_params) & 7) || (grad_param_row_stride & 1)))
        return NF_ERR_LAYOUT;
    AffineBwdArgs a;
    a.x = x; a.x_stride = x_row_stride; a.feat = feat_index; a.params = params; a.p_stride = param_row_stride;
    a.gy = grad_y; a.gy_stride = grad_y_row_stride; a.glad = grad_logabsdet; a.gx = grad_x; a.gx_stride = grad_x_row_stride;
    a.gp = grad_params; a.gp_stride = grad_param_row_stride; a.B = B; a.D = D; a.layout = layout;
    const int64_t want = (B * D + kThreads - 1) / kThreads;
    const int grid = (int)(want < 16384 ? want : 16384);
    hipStream_t s = (hipStream_t)stream;
    if (inverse) hipLaunchKernelGGL(affine_bwd_kernel<true>, dim3(grid), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL(affine_bwd_kernel<false>, dim3(grid), dim3(kThreads), 0, s, a);
    return nf_launch_status();
}

static int rqs_bwd_impl(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                        double pre_div, const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet,
                        float* grad_x, int64_t grad_x_row_stride, float* grad_params, int64_t B, int D, int K,
                        int tails, double tail_bound, double min_bin_width, double min_bin_height,
                        double min_derivative, int inverse, void* stream, uint32_t* grad_params_max = nullptr,
                        void* grad_params_bf16 = nullptr, const void* params_bf16 = nullptr) {
    if (B < 0 || D <= 0 || K <= 0 || K > 32) return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    if (grad_params_max) {                                 // the slot starts at 0 even for an empty batch
        const hipError_t me = hipMemsetAsync(grad_params_max, 0, sizeof(uint32_t), (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
    }
    if (B == 0) return NF_OK;
    if (!x || (!params && !params_bf16) || !grad_y || !grad_logabsdet || !grad_x || (!grad_params && !grad_params_bf16)) return NF_ERR_NULL;
    if (params_bf16 && (!grad_params_bf16 || (((uintptr_t)params_bf16 | (uintptr_t)grad_params_bf16) & 15) != 0)) return NF_ERR_LAYOUT;
    const int P = tails ? 3 * K - 1 : 3 * K + 1;
    if (param_row_stride != (int64_t)D * P) return NF_ERR_LAYOUT;
    if (x_row_stride < D || grad_y_row_stride < D || grad_x_row_stride < D) return NF_ERR_LAYOUT;
    RqsBwdArgs a;
    a.x = x; a.x_stride = x_row_stride; a.params = params; a.n_param_floats = B * (int64_t)D * P;
    a.gy = grad_y; a.gy_stride = grad_y_row_stride; a.glad = grad_logabsdet; a.gx = grad_x;
    a.gx_stride = grad_x_row_stride; a.gparams = grad_params; a.B = B; a.D = D;
    a.c = nf_make_rqs_const(K, tails, tail_bound, min_bin_width, min_bin_height, min_derivative, pre_div);
    a.rows_per_block = pick_rows_per_block(D, B);
    a.vec_ok = params_bf16 ? 1 : (((uintptr_t)params | (uintptr_t)grad_params) & 15) == 0 && ((uintptr_t)grad_params_bf16 & 7) == 0;
    a.inverse = inverse;
    a.gmax = grad_params_max;
    a.gparams16 = (__bf16*)grad_params_bf16;
    a.params16 = (const __bf16*)params_bf16;
    const size_t lds = sizeof(float) * (size_t)((kThreads * P + 8 + 7) & ~7);
    const int64_t grid = (B + a.rows_per_block - 1) / a.rows_per_block;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 4: return launch_rqs_bwd_k<4>(a, (int)grid, lds, s);
        case 8: return launch_rqs_bwd_k<8>(a, (int)grid, lds, s);
        case 10: return launch_rqs_bwd_k<10>(a, (int)grid, lds, s);
        case 16: return launch_rqs_bwd_k<16>(a, (int)grid, lds, s);
        default: return launch_rqs_bwd_k<0>(a, (int)grid, lds, s);
    }
}

extern "C" int nf_rqs_bwd(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                          double pre_div, const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet,
                          float* grad_x, int64_t grad_x_row_stride, float* grad_params, int64_t B, int D, int K,
                          int tails, double tail_bound, double min_bin_width, double min_bin_height,
                          double min_derivative, void* stream) {
    return rqs_bwd_impl(x, x_row_stride, params, param_row_stride, pre_div, grad_y, grad_y_row_stride, grad_logabsdet,
                        grad_x, grad_x_row_stride, grad_params, B, D, K, tails, tail_bound, min_bin_width,
                        min_bin_height, min_derivative, 0, stream);
}

// Backward of the linear / quadratic / cubic spline entry points, either direction (inverse: x = that pass's OUTPUT).
template <int FAM>
static int family_bwd(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                      const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet, float* grad_x,
                      int64_t grad_x_row_stride, float* grad_params, int64_t B, int D, int K, int P, int tails,
                      double tail_bound, double min_w, double min_h, int inverse, void* stream, double pre_div = 1.0,
                      bool symmetric = false) {
    if (B == 0) return NF_OK;
    if (!x || !params || !grad_y || !grad_logabsdet || !grad_x || !grad_params) return NF_ERR_NULL;
    if (param_row_stride != (int64_t)D * P) return NF_ERR_LAYOUT;
    if (x_row_stride < D || grad_y_row_stride < D || grad_x_row_stride < D) return NF_ERR_LAYOUT;
    RqsBwdArgs a;
    a.x = x; a.x_stride = x_row_stride; a.params = params; a.n_param_floats = B * (int64_t)D * P;
    a.gy = grad_y; a.gy_stride = grad_y_row_stride; a.glad = grad_logabsdet; a.gx = grad_x;
    a.gx_stride = grad_x_row_stride; a.gparams = grad_params; a.B = B; a.D = D;
    a.c = nf_make_rqs_const(K, tails, tail_bound, min_w, min_h, 1e-3, pre_div);
    a.c.P = P;
    if (symmetric) {
        a.c.left = (float)-tail_bound;
        a.c.right = (float)tail_bound;
        a.c.span = (float)(2.0 * tail_bound);
    }
    a.rows_per_block = pick_rows_per_block(D, B);
    a.vec_ok = (((uintptr_t)params | (uintptr_t)grad_params) & 15) == 0;
    a.inverse = inverse;
    a.gmax = nullptr;
    a.gparams16 = nullptr;
    a.params16 = nullptr;
    const size_t lds = 2 * sizeof(float) * (size_t)((kThreads * P + 4 + 3) & ~3);     // parameters + gradients
    if (lds > 160 * 1024) return NF_ERR_SHAPE;
    const int64_t grid = (B + a.rows_per_block - 1) / a.rows_per_block;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    return launch_family_bwd<FAM>(a, (int)grid, lds, (hipStream_t)stream);
}

extern "C" int nf_linear_spline_bwd(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                                    const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet,
                                    float* grad_x, int64_t grad_x_row_stride, float* grad_params, int64_t B, int D,
                                    int K, int tails, double tail_bound, int inverse, void* stream) {
    if (B < 0 || D <= 0 || K <= 0 || K > 64) return NF_ERR_SHAPE;
    return family_bwd<1>(x, x_row_stride, params, param_row_stride, grad_y, grad_y_row_stride, grad_logabsdet, grad_x,
                         grad_x_row_stride, grad_params, B, D, K, K, tails, tail_bound, 0.0, 0.0, inverse, stream);
}

extern "C" int nf_quadratic_spline_bwd(const float* x, int64_t x_row_stride, const float* params,
                                       int64_t param_row_stride, const float* grad_y, int64_t grad_y_row_stride,
                                       const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride,
                                       float* grad_params, int64_t B, int D, int K, int tails, double tail_bound,
                                       double min_bin_width, double min_bin_height, int inverse, void* stream) {
    if (B < 0 || D <= 0 || K < 2 || K > 32) return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    return family_bwd<2>(x, x_row_stride, params, param_row_stride, grad_y, grad_y_row_stride, grad_logabsdet, grad_x,
                         grad_x_row_stride, grad_params, B, D, K, tails ? 2 * K - 1 : 2 * K + 1, tails, tail_bound,
                         min_bin_width, min_bin_height, inverse, stream);
}

extern "C" int nf_cubic_spline_bwd(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                                   double pre_div, const float* grad_y, int64_t grad_y_row_stride,
                                   const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride,
                                   float* grad_params, int64_t B, int D, int K, int tails, double bound,
                                   double min_bin_width, double min_bin_height, int inverse, void* stream) {
    if (B < 0 || D <= 0 || K <= 0 || K > 32 || !(bound > 0.0) || !(pre_div > 0.0)) return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    return family_bwd<3>(x, x_row_stride, params, param_row_stride, grad_y, grad_y_row_stride, grad_logabsdet, grad_x,
                         grad_x_row_stride, grad_params, B, D, K, 2 * K + 2, tails, bound, min_bin_width,
                         min_bin_height, inverse, stream, pre_div, true);
}

// nf_rqs_bwd that also leaves the bit pattern of the largest finite |grad_params| in *grad_params_max (zeroed here,
// then atomicMax per workgroup): the operand scale the f16x2 GEMMs of the conditioner's backward need, without the
// extra pass over the [B, D*P] gradient that finding it afterwards costs.
extern "C" int nf_rqs_bwd_max(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                              double pre_div, const float* grad_y, int64_t grad_y_row_stride,
                              const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride, float* grad_params,
                              uint32_t* grad_params_max, int64_t B, int D, int K, int tails, double tail_bound,
                              double min_bin_width, double min_bin_height, double min_derivative, int inverse,
                              void* stream) {
    if (!grad_params_max) return NF_ERR_NULL;
    return rqs_bwd_impl(x, x_row_stride, params, param_row_stride, pre_div, grad_y, grad_y_row_stride, grad_logabsdet,
                        grad_x, grad_x_row_stride, grad_params, B, D, K, tails, tail_bound, min_bin_width,
                        min_bin_height, min_derivative, inverse ? 1 : 0, stream, grad_params_max);
}

// nf_rqs_bwd_max with grad_params written as bf16 [B, D*P] (8-byte aligned): for a caller whose next step rounds them
// to bf16 anyway - the bf16 training mode, where the two backward GEMMs of the final layer are their only readers.
// The slot still receives the largest |gradient| before rounding.
extern "C" int nf_rqs_bwd_max_bf16(const float* x, int64_t x_row_stride, const float* params, int64_t param_row_stride,
                                   double pre_div, const float* grad_y, int64_t grad_y_row_stride,
                                   const float* grad_logabsdet, float* grad_x, int64_t grad_x_row_stride,
                                   void* grad_params_bf16, uint32_t* grad_params_max, int64_t B, int D, int K, int tails,
                                   double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                                   int inverse, void* stream) {
    if (!grad_params_max || !grad_params_bf16) return NF_ERR_NULL;
    if (((uintptr_t)grad_params_bf16 & 7) != 0) return NF_ERR_LAYOUT;
    return rqs_bwd_impl(x, x_row_stride, params, param_row_stride, pre_div, grad_y, grad_y_row_stride, grad_logabsdet,
                        grad_x, grad_x_row_stride, nullptr, B, D, K, tails, tail_bound, min_bin_width, min_bin_height,
                        min_derivative, inverse ? 1 : 0, stream, grad_params_max, grad_params_bf16);
}

// nf_rqs_bwd_max_bf16 on parameters that were KEPT as bf16 (nf_made_rqs_fwd_params_bf16): both [B, D*P] tensors of the
// backward are 2 bytes per element (forward direction only).
extern "C" int nf_rqs_bwd_max_bf16p(const float* x, int64_t x_row_stride, const void* params_bf16, int64_t param_row_stride,
                                    const float* grad_y, int64_t grad_y_row_stride, const float* grad_logabsdet,
                                    float* grad_x, int64_t grad_x_row_stride, void* grad_params_bf16,
                                    uint32_t* grad_params_max, int64_t B, int D, int K, int tails, double tail_bound,
                                    double min_bin_width, double min_bin_height, double min_derivative, void* stream) {
    if (!grad_params_max || !grad_params_bf16 || !params_bf16) return NF_ERR_NULL;
    return rqs_bwd_impl(x, x_row_stride, nullptr, param_row_stride, 1.0, grad_y, grad_y_row_stride, grad_logabsdet,
                        grad_x, grad_x_row_stride, nullptr, B, D, K, tails, tail_bound, min_bin_width, min_bin_height,
                        min_derivative, 0, stream, grad_params_max, grad_params_bf16, params_bf16);
}

// Backward of nf_rqs_fwd(inverse = 1): x_out = the inverse pass's OUTPUT, grad_x_out / grad_logabsdet the gradients
// wrt that output and its (negated) logdet; writes the gradient wrt the pass's INPUT and wrt the parameters.
extern "C" int nf_rqs_inverse_bwd(const float* x_out, int64_t x_row_stride, const float* params,
                                  int64_t param_row_stride, double pre_div, const float* grad_x_out,
                                  int64_t grad_x_out_row_stride, const float* grad_logabsdet, float* grad_y_in,
                                  int64_t grad_y_in_row_stride, float* grad_params, int64_t B, int D, int K, int tails,
                                  double tail_bound, double min_bin_width, double min_bin_height,
                                  double min_derivative, void* stream) {
    return rqs_bwd_impl(x_out, x_row_stride, params, param_row_stride, pre_div, grad_x_out, grad_x_out_row_stride,
                        grad_logabsdet, grad_y_in, grad_y_in_row_stride, grad_params, B, D, K, tails, tail_bound,
                        min_bin_width, min_bin_height, min_derivative, 1, stream);
}
