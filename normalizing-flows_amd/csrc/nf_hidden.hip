// The MADE hidden stack in ONE launch (gfx950, f16x2 arithmetic).
//
//   temps = initial_layer(x); for block in blocks: temps = block(temps)        (models/made.py:352-357)
//   residual block:  temps + L1(relu(L0(relu(temps))))                        (made.py:268-284)
//   feed-forward block: relu(L(temps))                                         (made.py:196-207)
//
// As separate launches each of the five 512-wide layers of cfg-3 reads and writes a [B, 512] fp32 tensor (268 MB per
// 131072 rows, 0.21 ms each: HBM time, not matrix time) and a sixth pass re-reads the last one to split it into the fp16
// term planes the fused final-layer kernel takes.  Here a 512-thread workgroup carries 64 samples through the whole
// stack: the layer input lives in LDS as two fp16 term planes (power-of-two scaled by the largest magnitude of THIS
// 64-sample tile: the scale is per row group, not per launch), every layer is D^T[unit][sample] = W . X^T on
// v_mfma_f32_32x32x16_f16 (three MFMAs per product: W1 X0 + W0 X1 + W0 X0, fp32 accumulate), the hidden state itself
// stays in fp32 accumulator registers from layer to layer (the residual sum is taken there), and what leaves the CU is
// only the final hidden state - already as the term planes nf_made_rqs_fwd_planes reads, with the tile's scale beside it.
//
// Work split: wave p of 8 owns the 32-unit groups p and NG-1-p.  With MADE's degree-ordered masks group g reads hidden
// units 0 .. 32 g + 31 only, so the two groups of a wave always add up to the same K range (34 steps of 16 at H = 512)
// and no wave waits for another one's longer loop.  Weights are private to a wave and go from L2 straight into MFMA
// operand registers (packed in operand order, three register sets deep); the activations are shared by all waves and
// are read from LDS, each fragment reloaded for the next K step right behind its last MFMA.
#include <string.h>
#include "nf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int HS_ROWS = 64;            // samples per workgroup
constexpr int HS_T = 512;              // threads (8 waves)
constexpr int HS_MAXH = 512;           // hidden width limit: 64 x 512 x 2 terms x 2 B = 128 KB of LDS
constexpr int HS_MAXSTEPS = HS_MAXH / 16;
constexpr int HS_MAXL = 8;             // layers of the stack
constexpr int HS_PLANES = HS_MAXSTEPS * 4 * 1024;   // [K step][term][k half][64 samples][16 B]

enum { HS_INIT = 0, HS_RES_A = 1, HS_RES_B = 2, HS_FF = 3 };

struct HiddenLayerDesc {
    int type;                // HS_INIT: H = L(x); HS_RES_A: T = L(relu(H)); HS_RES_B: H += L(relu(T)); HS_FF: H = relu(L(H))
    int kin;                 // input features
    int ksl;                 // K steps stored per group (ceil(kin / 16))
    unsigned wmax;           // bit pattern of max |W * mask| (filled on the device)
    int kend[16];            // live K steps of each 32-unit group (filled on the device)
    long long w_off;         // byte offsets into the blob
    long long b_off;
};

struct HiddenHdr {
    int n_layers, hidden, ng, pad;
    HiddenLayerDesc layer[HS_MAXL];
};

constexpr size_t HS_HDR_BYTES = 4096;
static_assert(sizeof(HiddenHdr) <= HS_HDR_BYTES, "header region");

__device__ __forceinline__ void pow2_pair_h(unsigned bits, float& s, float& inv) {
    // s = 2^e with m * 2^e in [2^13, 2^14) for the float with this bit pattern (0 / denormal: s = 1); inv = 2^-e
    const int ex = (int)((bits >> 23) & 0xff) - 126;
    int e = ((bits >> 23) & 0xff) == 0 ? 0 : 14 - ex;
    e = max(-100, min(100, e));
    s = __uint_as_float((unsigned)(e + 127) << 23);
    inv = __uint_as_float((unsigned)(127 - e) << 23);
}

// ---- pack ------------------------------------------------------------------------------------------------
__global__ void hs_header_kernel(HiddenHdr h, HiddenHdr* dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = h;
}

// max |W * mask| and the live K range of every 32-unit group of one layer
__global__ __launch_bounds__(256) void hs_scan_kernel(const float* __restrict__ W, const float* __restrict__ mask,
                                                      int out, int in, HiddenHdr* hdr, int layer) {
    float mx = 0.f;
    const int64_t total = (int64_t)out * in;
    const int lane = threadIdx.x & 63;
    // a wave takes 64 consecutive elements - as a rule one weight row, so one 32-unit group: its largest live K step
    // goes out as ONE atomic (one per unmasked element was a quarter of a million atomics on 16 addresses, all in
    // flight at once: 1.2 ms per layer).  The loop is wave-uniform so that the shuffles see every lane.
    const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(int64_t)63;
    for (int64_t base = w0; base < total; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + lane;
        const bool valid = i < total;
        const int64_t ic = valid ? i : total - 1;
        const int u = (int)(ic / in), k = (int)(ic - (int64_t)u * in);
        const float m = !valid ? 0.f : (mask ? mask[ic] : 1.f);
        int ks = 0;
        if (m != 0.f) {
            const float a = fabsf(W[ic] * m);
            if (a < INFINITY) mx = fmaxf(mx, a);
            ks = k / 16 + 1;
        }
        const int grp = u >> 5;
        const int g0 = __builtin_amdgcn_readfirstlane(grp);
        if (__all(grp == g0)) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ks = max(ks, __shfl_xor(ks, off, 64));
            if (lane == 0 && ks > 0) atomicMax(&hdr->layer[layer].kend[g0], ks);
        } else if (ks > 0) {
            atomicMax(&hdr->layer[layer].kend[grp], ks);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if (lane == 0 && mx > 0.f) atomicMax(&hdr->layer[layer].wmax, __float_as_uint(mx));
}

// weights of one layer in MFMA A-operand order: [group][K step][term][lane][8 halfs], lane l = unit 32 g + l % 32,
// k = 16 step + 8 (l / 32) .. + 7; masked, scaled by the layer's power of two, zero beyond [out, in]
__global__ __launch_bounds__(256) void hs_pack_kernel(const float* __restrict__ W, const float* __restrict__ mask,
                                                      const float* __restrict__ bias, int out, int in, int ng,
                                                      const HiddenHdr* hdr, int layer, char* blob) {
    const HiddenLayerDesc& L = hdr->layer[layer];
    float ws, wi;
    pow2_pair_h(L.wmax, ws, wi);
    const int ksl = L.ksl;
    f16x8* wp = (f16x8*)(blob + L.w_off);
    const int64_t total = (int64_t)ng * ksl * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t gs = i >> 6;
        const int kt = (int)(gs % ksl), g = (int)(gs / ksl);
        const int u = 32 * g + (lane & 31), k0 = 16 * kt + 8 * (lane >> 5);
        f16x8 t0, t1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (u < out && k0 + j < in) {
                const int64_t idx = (int64_t)u * in + k0 + j;
                v = W[idx] * (mask ? mask[idx] : 1.f) * ws;
            }
            const _Float16 a = (_Float16)v;
            t0[j] = a;
            t1[j] = (_Float16)(v - (float)a);
        }
        wp[(gs * 2 + 0) * 64 + lane] = t0;
        wp[(gs * 2 + 1) * 64 + lane] = t1;
    }
    float* bp = (float*)(blob + L.b_off);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 32 * ng; i += gridDim.x * blockDim.x)
        bp[i] = (bias && i < out) ? bias[i] : 0.f;
}

// ---- the stack -------------------------------------------------------------------------------------------
struct HiddenArgs {
    const float* x;          // [M, >= kin of layer 0]
    int64_t ldx;
    const char* blob;
    void* planes;            // out: [row tile of 128][K step][term][slot = 2 row + (half ^ bit 3 of the row)][8 halfs]
    unsigned* row_scale;     // out: bit pattern of max |final hidden state| per 64-row group
    float* h_out;            // nullable [M, hidden]: the final hidden state in fp32 as well
    int64_t ldh;
    int64_t M;
    int in0;                 // features of x
};

// LDS address of the 16-byte chunk (K chunk cc = 2 step + half, term, sample): samples rotated by 8 per K chunk so
// that the 32 B-strided writes of the input staging spread over all banks; a fragment read (32 consecutive samples of
// one chunk) stays one contiguous run modulo the wrap
__device__ __forceinline__ int hs_chunk(int cc, int term, int sample) {
    return ((cc >> 1) * 4 + term * 2 + (cc & 1)) * 1024 + ((sample + 8 * (cc & 7)) & 63) * 16;
}

__device__ __forceinline__ void hs_split4(float a, float b, float c, float d, float s, f16x4& t0, f16x4& t1) {
    const float v[4] = {a * s, b * s, c * s, d * s};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 h = (_Float16)v[j];
        t0[j] = h;
        t1[j] = (_Float16)(v[j] - (float)h);
    }
}

#define HS_SB() __builtin_amdgcn_sched_barrier(0);

__global__ __launch_bounds__(HS_T, 1) void made_hidden_f16x2_kernel(const HiddenArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* red = (float*)(lds + HS_PLANES);                       // [8] wave maxima
    const HiddenHdr* hdr = (const HiddenHdr*)g.blob;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * HS_ROWS;
    const int NL = hdr->n_layers, H = hdr->hidden, NG = hdr->ng;
    // this wave's unit groups: B = NG-1-w (the longer K range under degree-ordered masks), A = w
    const int gA = w, gB = NG - 1 - w;
    const bool hasB = gB >= gA, hasA = gB > gA;
    const int lh = lane >> 5, ls = lane & 31;

    // ---- stage x: planes <- split(x * 2^e), e from the tile's largest |x| over the columns any unit reads --------
    int ksx = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) ksx = max(ksx, q < NG ? hdr->layer[0].kend[q] : 0);
    ksx = min(ksx, HS_MAXSTEPS);
    float sc_act, inv_act;
    {
        const int row = tid >> 3, c = tid & 7;                    // 8 lanes read 256 contiguous bytes of a row
        const int64_t grow = min(m0 + row, g.M - 1);
        const float* xr = g.x + grow * g.ldx;
        const int nch = ksx * 2;                                  // 8-float chunks per row
        float4 v[8][2];
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cc = i * 8 + c;
            const int k0 = cc * 8;
            v[i][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            v[i][1] = v[i][0];
            if (cc < nch) {
                if (k0 + 8 <= g.in0 && ((g.ldx & 3) == 0) && (((uintptr_t)g.x & 15) == 0)) {
                    v[i][0] = *reinterpret_cast<const float4*>(xr + k0);
                    v[i][1] = *reinterpret_cast<const float4*>(xr + k0 + 4);
                } else {
                    float t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = k0 + j < g.in0 ? xr[k0 + j] : 0.f;
                    v[i][0] = make_float4(t[0], t[1], t[2], t[3]);
                    v[i][1] = make_float4(t[4], t[5], t[6], t[7]);
                }
            }
            const float a = fmaxf(fmaxf(fmaxf(fabsf(v[i][0].x), fabsf(v[i][0].y)), fmaxf(fabsf(v[i][0].z), fabsf(v[i][0].w))),
                                  fmaxf(fmaxf(fabsf(v[i][1].x), fabsf(v[i][1].y)), fmaxf(fabsf(v[i][1].z), fabsf(v[i][1].w))));
            mx = fmaxf(mx, a < INFINITY ? a : 0.f);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        if (lane == 0) red[w] = mx;
        __syncthreads();
        float m8 = red[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) m8 = fmaxf(m8, red[q]);
        pow2_pair_h(__float_as_uint(m8), sc_act, inv_act);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int cc = i * 8 + c;
            if (cc < nch) {
                f16x4 a0, a1, b0, b1;
                hs_split4(v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, sc_act, a0, a1);
                hs_split4(v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w, sc_act, b0, b1);
                f16x8 t0, t1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { t0[j] = a0[j]; t0[4 + j] = b0[j]; t1[j] = a1[j]; t1[4 + j] = b1[j]; }
                *reinterpret_cast<f16x8*>(lds + hs_chunk(cc, 0, row)) = t0;
                *reinterpret_cast<f16x8*>(lds + hs_chunk(cc, 1, row)) = t1;
            }
        }
    }

    f32x16 Hh[2][2], Tt[2][2];          // [group A / B][sample tile]: hidden state and accumulator / temporary
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { Hh[a][n][r] = 0.f; Tt[a][n][r] = 0.f; }

#pragma unroll 1
    for (int l = 0; l < NL; ++l) {
        const HiddenLayerDesc& L = hdr->layer[l];
        const int type = L.type, ksl = L.ksl;
        float wsc, winv;
        pow2_pair_h(L.wmax, wsc, winv);
        const int kcap = min(ksl, HS_MAXSTEPS);
        const int nA = hasA ? min(L.kend[gA & 15], kcap) : 0;
        const int nBl = hasB ? min(L.kend[gB & 15], kcap) : 0;
        const int n2 = max(nA, nBl);                  // steps of group B (zero weights past its own range)
        const int n1 = nA;                            // steps both groups take
        const f16x8* wA = (const f16x8*)(g.blob + L.w_off) + (int64_t)(hasA ? gA : 0) * ksl * 128 + lane;
        const f16x8* wB = (const f16x8*)(g.blob + L.w_off) + (int64_t)(hasB ? gB : 0) * ksl * 128 + lane;
        const int lastA = max(n1 - 1, 0), lastB = max(n2 - 1, 0);

        // weights of the first two steps: issued before the barrier, they land while the planes are being finished
        f16x8 wa0[2], wa1[2], wa2[2], wb0[2], wb1[2], wb2[2];
#define HS_LDW(S_, P_, K_) { S_[0] = (P_)[(int64_t)(K_) * 128]; S_[1] = (P_)[(int64_t)(K_) * 128 + 64]; }
        HS_LDW(wa0, wA, 0) HS_LDW(wb0, wB, 0)
        HS_LDW(wa1, wA, min(1, lastA)) HS_LDW(wb1, wB, min(1, lastB))
        wa2[0] = wa0[0]; wa2[1] = wa0[1]; wb2[0] = wb0[0]; wb2[1] = wb0[1];

        // accumulators start at bias * 2^(e_w + e_act) (exact scaling): out = acc * 2^-(e_w + e_act)
        {
            const float S = wsc * sc_act;
            const float* bp = (const float*)(g.blob + L.b_off);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 ba = *reinterpret_cast<const float4*>(bp + 32 * (hasA ? gA : 0) + 8 * q + 4 * lh);
                const float4 bb = *reinterpret_cast<const float4*>(bp + 32 * (hasB ? gB : 0) + 8 * q + 4 * lh);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    Tt[0][n][4 * q + 0] = ba.x * S; Tt[0][n][4 * q + 1] = ba.y * S;
                    Tt[0][n][4 * q + 2] = ba.z * S; Tt[0][n][4 * q + 3] = ba.w * S;
                    Tt[1][n][4 * q + 0] = bb.x * S; Tt[1][n][4 * q + 1] = bb.y * S;
                    Tt[1][n][4 * q + 2] = bb.z * S; Tt[1][n][4 * q + 3] = bb.w * S;
                }
            }
        }
        __syncthreads();                              // the planes of this layer's input are complete

        // activation fragments of step k: [sample tile][term], lane = sample 32 n + l % 32, k half l / 32
        f16x8 x00, x01, x10, x11;
#define HS_LDX(N_, K_, T0_, T1_) {                                                                   \
            const int cc_ = 2 * (K_) + lh;                                                           \
            T0_ = *reinterpret_cast<const f16x8*>(lds + hs_chunk(cc_, 0, 32 * (N_) + ls));           \
            T1_ = *reinterpret_cast<const f16x8*>(lds + hs_chunk(cc_, 1, 32 * (N_) + ls)); }
#define HS_MM(ACC_, W_, X0_, X1_)                                                                    \
        ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(W_[0], X1_, ACC_, 0, 0, 0);                    \
        ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(W_[1], X0_, ACC_, 0, 0, 0);                    \
        ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(W_[0], X0_, ACC_, 0, 0, 0);
        // one K step of both groups: weights of step k in (CA_, CB_), those of step k + 2 requested into (LA_, LB_)
#define HS_STEP2(CA_, CB_, LA_, LB_) {                                                               \
            const int kp_ = k + 2, kn_ = min(k + 1, lastB);                                          \
            HS_LDW(LA_, wA, min(kp_, lastA)) HS_LDW(LB_, wB, min(kp_, lastB))                        \
            HS_SB()                                                                                  \
            HS_MM(Tt[0][0], CA_, x00, x01) HS_MM(Tt[1][0], CB_, x00, x01)                            \
            HS_SB()                                                                                  \
            HS_LDX(0, kn_, x00, x01)                                                                 \
            HS_SB()                                                                                  \
            HS_MM(Tt[0][1], CA_, x10, x11) HS_MM(Tt[1][1], CB_, x10, x11)                            \
            HS_SB()                                                                                  \
            HS_LDX(1, kn_, x10, x11)                                                                 \
            HS_SB()                                                                                  \
            ++k; }
#define HS_STEP1(CB_, LB_) {                                                                         \
            const int kp_ = k + 2, kn_ = min(k + 1, lastB);                                          \
            HS_LDW(LB_, wB, min(kp_, lastB))                                                         \
            HS_SB()                                                                                  \
            HS_MM(Tt[1][0], CB_, x00, x01)                                                           \
            HS_SB()                                                                                  \
            HS_LDX(0, kn_, x00, x01)                                                                 \
            HS_SB()                                                                                  \
            HS_MM(Tt[1][1], CB_, x10, x11)                                                           \
            HS_SB()                                                                                  \
            HS_LDX(1, kn_, x10, x11)                                                                 \
            HS_SB()                                                                                  \
            ++k; }
        HS_LDX(0, 0, x00, x01) HS_LDX(1, 0, x10, x11)
        int k = 0, slot = 0;
        for (;;) {
            if (k >= n1) { slot = 0; break; }
            HS_STEP2(wa0, wb0, wa2, wb2)
            if (k >= n1) { slot = 1; break; }
            HS_STEP2(wa1, wb1, wa0, wb0)
            if (k >= n1) { slot = 2; break; }
            HS_STEP2(wa2, wb2, wa1, wb1)
        }
        while (k < n2) {
            if (slot == 0) { HS_STEP1(wb0, wb2) slot = 1; if (k >= n2) break; }
            if (slot == 1) { HS_STEP1(wb1, wb0) slot = 2; if (k >= n2) break; }
            if (slot == 2) { HS_STEP1(wb2, wb1) slot = 0; }
        }
#undef HS_STEP1
#undef HS_STEP2
#undef HS_MM
#undef HS_LDX
#undef HS_LDW

        // ---- combine: out = acc * 2^-(e_w + e_act), into H or T as the layer's role says (one uniform branch per
        //      role around straight-line loops: as per-element selects on `type` this phase was 5000 instructions) -----
        const float inv = winv * inv_act;
        const int next = l + 1 < NL ? hdr->layer[l + 1].type : -1;
        const float lo = (next == HS_RES_A || next == HS_RES_B) ? 0.f : -INFINITY;     // ReLU on the next layer's input
        const bool src_t = next == HS_RES_B;          // the next layer reads T (else H; the final planes are H)
#define HS_ALL(BODY_)                                                                                \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                \
        _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { BODY_ }
        if (type == HS_RES_A) { HS_ALL(Tt[a][n][r] = Tt[a][n][r] * inv;) }
        else if (type == HS_RES_B) { HS_ALL(Hh[a][n][r] = Hh[a][n][r] + Tt[a][n][r] * inv;) }
        else if (type == HS_FF) { HS_ALL(Hh[a][n][r] = fmaxf(Tt[a][n][r] * inv, 0.f);) }
        else { HS_ALL(Hh[a][n][r] = Tt[a][n][r] * inv;) }
#undef HS_ALL
        // largest magnitude of what the next layer reads, over the groups this wave owns
        float mx = 0.f;
#define HS_MAX(R_)                                                                                   \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                              \
            if (a == 0 ? !hasA : !hasB) continue;                                                    \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                            \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
                const float av = fabsf(fmaxf(R_[a][n][r], lo));                                      \
                mx = fmaxf(mx, av < INFINITY ? av : 0.f);                                            \
            }                                                                                        \
        }
        if (src_t) { HS_MAX(Tt) } else { HS_MAX(Hh) }
#undef HS_MAX
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        __syncthreads();                              // every wave is done reading the planes (and `red`)
        if (lane == 0) red[w] = mx;
        __syncthreads();
        float m8 = red[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) m8 = fmaxf(m8, red[q]);
        pow2_pair_h(__float_as_uint(m8), sc_act, inv_act);
        if (next < 0 && tid == 0) g.row_scale[blockIdx.x] = __float_as_uint(m8);

        // ---- planes <- split(src * 2^e): a lane holds 4 consecutive units per accumulator quad = half a K chunk ----
#define HS_WRITE(R_)                                                                                 \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                              \
            if (a == 0 ? !hasA : !hasB) continue;                                                    \
            const int gg = a == 0 ? gA : gB;                                                         \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                            \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                          \
                f16x4 t0, t1;                                                                        \
                hs_split4(fmaxf(R_[a][n][4 * q], lo), fmaxf(R_[a][n][4 * q + 1], lo),                \
                          fmaxf(R_[a][n][4 * q + 2], lo), fmaxf(R_[a][n][4 * q + 3], lo), sc_act, t0, t1);   \
                const int cc = 4 * gg + q;                                                           \
                *reinterpret_cast<f16x4*>(lds + hs_chunk(cc, 0, 32 * n + ls) + 8 * lh) = t0;         \
                *reinterpret_cast<f16x4*>(lds + hs_chunk(cc, 1, 32 * n + ls) + 8 * lh) = t1;         \
            }                                                                                        \
        }
        if (src_t) { HS_WRITE(Tt) } else { HS_WRITE(Hh) }
#undef HS_WRITE
        if (next < 0 && g.h_out) {                    // the final hidden state in fp32 as well (callers that keep it)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (a == 0 ? !hasA : !hasB) continue;
                const int gg = a == 0 ? gA : gB;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int64_t row = m0 + 32 * n + ls;
                    if (row < g.M)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(g.h_out + row * g.ldh + 32 * gg + 8 * q + 4 * lh) =
                                make_float4(Hh[a][n][4 * q], Hh[a][n][4 * q + 1], Hh[a][n][4 * q + 2], Hh[a][n][4 * q + 3]);
                }
            }
        }
    }
    __syncthreads();

    // ---- the final planes to global memory in the fused final-layer kernel's tiled layout --------------------------
    {
        const int steps = H / 16;
        char* dst = (char*)g.planes + ((m0 >> 7) * steps) * (int64_t)8192 + (m0 & 64) * 32;   // + (step * 2 + term) * 4096 + slot * 16
        for (int i = tid; i < steps * 2 * 128; i += HS_T) {
            const int slot = i & 127, kt2 = i >> 7;          // kt2 = step * 2 + term
            const int sample = slot >> 1, half = (slot & 1) ^ ((sample >> 3) & 1);
            const f16x8 v = *reinterpret_cast<const f16x8*>(lds + hs_chunk((kt2 >> 1) * 2 + half, kt2 & 1, sample));
            *reinterpret_cast<f16x8*>(dst + (int64_t)kt2 * 4096 + slot * 16) = v;
        }
    }
}

inline int hs_supported(int n_layers, int hidden, int k_live0) {
    return n_layers >= 1 && n_layers <= HS_MAXL && hidden >= 32 && hidden <= HS_MAXH && hidden % 32 == 0 &&
           k_live0 >= 1 && k_live0 <= HS_MAXH;
}

inline size_t hs_layer_w_bytes(int hidden, int kin) { return (size_t)(hidden / 32) * ((kin + 15) / 16) * 2048; }

}  // namespace

extern "C" int nf_made_hidden_supported(int n_layers, int hidden, int k_live0) {
    return hs_supported(n_layers, hidden, k_live0) ? 1 : 0;
}

extern "C" size_t nf_made_hidden_packed_bytes(int n_layers, int hidden, const int* in_features) {
    if (!in_features || n_layers < 1 || n_layers > HS_MAXL || hidden < 32 || hidden % 32) return 0;
    size_t total = HS_HDR_BYTES;
    for (int l = 0; l < n_layers; ++l) {
        if (in_features[l] <= 0) return 0;
        total += hs_layer_w_bytes(hidden, in_features[l]) + (size_t)hidden * 4;
    }
    return total;
}

extern "C" int nf_made_hidden_pack(const float* const* W, const float* const* mask, const float* const* bias,
                                   const int* in_features, const int* types, int n_layers, int hidden, int k_live0,
                                   void* packed, void* stream) {
    if (!W || !in_features || !types || !packed) return NF_ERR_NULL;
    if (!hs_supported(n_layers, hidden, k_live0)) return NF_ERR_SHAPE;
    if (((uintptr_t)packed & 15) != 0) return NF_ERR_LAYOUT;
    hipStream_t s = (hipStream_t)stream;
    HiddenHdr h;
    memset(&h, 0, sizeof(h));
    h.n_layers = n_layers; h.hidden = hidden; h.ng = hidden / 32;
    size_t off = HS_HDR_BYTES;
    for (int l = 0; l < n_layers; ++l) {
        if (!W[l] || in_features[l] <= 0) return NF_ERR_NULL;
        if (l > 0 && in_features[l] != hidden) return NF_ERR_SHAPE;
        if (types[l] < HS_INIT || types[l] > HS_FF || (l == 0) != (types[l] == HS_INIT)) return NF_ERR_SHAPE;
        if (types[l] == HS_RES_B && types[l - 1] != HS_RES_A) return NF_ERR_SHAPE;
        if (types[l] == HS_RES_A && (l + 1 >= n_layers || types[l + 1] != HS_RES_B)) return NF_ERR_SHAPE;
        HiddenLayerDesc& L = h.layer[l];
        L.type = types[l]; L.kin = in_features[l]; L.ksl = (in_features[l] + 15) / 16;
        L.w_off = (long long)off;
        off += hs_layer_w_bytes(hidden, in_features[l]);
        L.b_off = (long long)off;
        off += (size_t)hidden * 4;
    }
    HiddenHdr* dh = (HiddenHdr*)packed;
    hipLaunchKernelGGL(hs_header_kernel, dim3(1), dim3(64), 0, s, h, dh);
    for (int l = 0; l < n_layers; ++l) {
        const int64_t total = (int64_t)hidden * in_features[l];
        const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
        hipLaunchKernelGGL(hs_scan_kernel, dim3(grid), dim3(256), 0, s, W[l], mask ? mask[l] : nullptr, hidden,
                           in_features[l], dh, l);
    }
    for (int l = 0; l < n_layers; ++l) {
        const int64_t total = (int64_t)h.ng * h.layer[l].ksl * 64;
        const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(hs_pack_kernel, dim3(grid), dim3(256), 0, s, W[l], mask ? mask[l] : nullptr,
                           bias ? bias[l] : nullptr, hidden, in_features[l], h.ng, dh, l, (char*)packed);
    }
    return nf_launch_status();
}

extern "C" size_t nf_made_hidden_planes_bytes(int64_t B, int hidden) {
    if (B <= 0 || hidden <= 0) return 0;
    return (size_t)((B + 127) / 128 * 128) * (size_t)((hidden + 15) / 16 * 16) * 4;
}

extern "C" int nf_made_hidden_planes_fwd(const float* x, int64_t x_row_stride, const void* packed, int n_layers,
                                         int hidden, int in0, void* planes, uint32_t* row_scale, float* h_out,
                                         int64_t h_row_stride, int64_t B, void* stream) {
    if (B < 0 || in0 <= 0 || n_layers < 1 || n_layers > HS_MAXL || hidden < 32 || hidden > HS_MAXH || hidden % 32)
        return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!x || !packed || !planes || !row_scale) return NF_ERR_NULL;
    if (x_row_stride < in0) return NF_ERR_LAYOUT;
    if (((uintptr_t)planes & 15) != 0 || ((uintptr_t)packed & 15) != 0) return NF_ERR_LAYOUT;
    if (h_out && (h_row_stride < hidden || (h_row_stride & 3) != 0 || ((uintptr_t)h_out & 15) != 0)) return NF_ERR_LAYOUT;
    const int64_t grid = (B + HS_ROWS - 1) / HS_ROWS;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    static unsigned long long cap_done = 0;
    const int rc = nf_raise_lds_cap((const void*)made_hidden_f16x2_kernel, &cap_done);
    if (rc != NF_OK) return rc;
    HiddenArgs g;
    g.x = x; g.ldx = x_row_stride; g.blob = (const char*)packed; g.planes = planes; g.row_scale = row_scale;
    g.h_out = h_out; g.ldh = h_row_stride; g.M = B; g.in0 = in0;
    hipLaunchKernelGGL(made_hidden_f16x2_kernel, dim3((unsigned)grid), dim3(HS_T), HS_PLANES + 64, (hipStream_t)stream, g);
    return nf_launch_status();
}
