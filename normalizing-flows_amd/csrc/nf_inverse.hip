// Autoregressive inverse (sampling) of a rational-quadratic spline MAF layer, row-resident (gfx950).
//
// The reference inverts feature by feature with D full conditioner passes (models/transforms.py:493-500).
// The MADE masks (models/made.py:119-148) make that a recurrence with very little new work per feature:
// with hidden units sorted by degree (unit j has degree j+1, every hidden layer alike), output feature f
// reads hidden units 0..f-1 only, and input f reaches unit k only for k >= f.  So once x[f] is known
//   * unit f of every hidden layer becomes final (nothing later can change it), and
//   * the parameters of feature f+1 are one 32-column row block of the final layer over units 0..f.
// The samples never interact, so a workgroup keeps 32 samples from x[0] to x[H] without any grid-wide step,
// as a software pipeline with ONE barrier per feature ("round" j = feature j / unit j):
//
//   wave 0 (the critical path): completes the parameters of feature j (bulk part from LDS + the rank-1 terms of
//     the two newest hidden units, one MFMA), inverts the spline (nf_rqs_eval.h: lanes 0..31 = samples, lanes
//     32..63 build the width knots of the same samples), then finishes unit j of every hidden layer for its
//     sample - a chain of NSTEP scalar FMAs, because everything older than unit j-1 is already summed in a partial
//     that unit j's owner exported a round earlier - and publishes x[j], unit j's activations and its final hidden
//     value through LDS.
//   all waves, one round behind ("right-looking"): wave w owns units 64w..64w+63 of every hidden layer, their
//     running sums for the 32 samples in MFMA accumulator layout (NSTEP x 32 registers); unit j-1's activations
//     enter as one rank-1 v_mfma_f32_32x32x1_2b_f32 per layer (weights: transposed pack, a 2 KB coalesced row per
//     layer and round, reloaded a round ahead); the wave holding unit j+1 exports its sums.
//   waves 1..7, "left-looking": the final-layer row block of feature j+2 over hidden units 0..j-1, a
//     32 x 32 x j product on v_mfma_f32_32x32x2_f32 with K split over the waves (weights packed per feature in
//     operand order, triangular: only the live K range is stored and read); the partial tiles of the
//     previous round are summed in fixed order into the bulk parameters wave 0 reads next round.
//
// Features past H+1 depend on no further hidden unit: the kernel leaves the final hidden state in HBM and
// the caller finishes them with one windowed launch of the fused layer kernel (nf_made_rqs_fwd_tiles).
// Work per layer = about one forward pass (the reference: D forward passes; the windowed incremental inverse
// in transforms.py: one 128/256-column tile per layer per feature, ~10-100 x more MACs than needed).
#include "nf_common.h"
#include "nf_rqs_eval.h"
#include "nf_spline_families.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

constexpr int IR = 32;            // samples per workgroup
constexpr int IT = 512;           // threads = hidden units a workgroup can own
constexpr int IW = IT / 64;
constexpr int MAXSTEP = 5;        // hidden vectors: initial layer + 2 per residual block (1 per feed-forward block)
constexpr int PSTR = 33;          // stride of the parameter tiles [column][sample]
constexpr int SS = 8;             // hidden units per K super-step of the final layer (4 MFMAs of K = 2)
constexpr int NP = 2;             // B-operand slots loaded a round ahead (kept in rs[4..12) across the barrier)
constexpr int NQ = 4;             // B-operand slots loaded at the top of the round and reused in place
constexpr int MW = IW - 1;        // waves on the final-layer product (wave 0 inverts the spline meanwhile)
constexpr int NI = (IT / SS + MW - 1) / MW;   // super-steps per wave at most

constexpr int F_RELU_IN = 1, F_RESIDUAL = 2, F_RELU_OUT = 4;
// elementwise transform instead of the RQ spline (KT >= 0: its bin count, 0 = runtime): affine (2 parameters per
// feature), linear / quadratic / cubic spline (nf_spline_families.h)
constexpr int KT_AFFINE = -1, KT_LINEAR = -2, KT_QUADRATIC = -3, KT_CUBIC = -4;

// First super-step of feature f in the triangular final-layer pack: sum over g < f of ceil(g / 8)
__host__ __device__ inline int first_superstep(int f) {
    if (f <= 1) return 0;
    const int m = f - 1, q = m / SS;
    return (q + 1) * (m - (SS / 2) * q);
}

struct InvArgs {
    const float* z;
    int64_t ldz;
    float* x;
    int64_t ldx;
    float* hid;
    int64_t ldh;
    float* partial;               // [tiles][B] logdet partial sums of the fused layer kernel
    const float* wt[MAXSTEP];     // [H][HS]: wt[s][j*HS + k] = weight (input j -> unit k), mask applied
    const float* bias[MAXSTEP];   // [HS]
    const float* bf;              // [H+1][32] final-layer bias of features 0..H, columns >= P zero
    const float* wl;              // [H+1][2][32] wl[f][0][c] = W[f*P+c][f-1], wl[f][1][c] = W[f*P+c][f-2]
    const float* dg;              // [H][32] unit j: [s] = step s's weight on unit j of the step before, [8+s] = on unit j-1,
                                  //         [16+s], [24+s] = eval-BatchNorm scale / shift on the input of step s (1 / 0 without)
    const float4* wf;             // [super-step][lane]: lane (kh, col) holds W[f*P+col][8S+4kh .. +3], units >= f-2 zeroed
    const float* ctx0;            // nullable [B][ldc]: context term added to the initial layer's output (made.py:354-355)
    const float* gate[2];         // nullable [B][ldc]: sigmoid gates of the residual blocks' second layer (made.py:279-283)
    int64_t ldc;
    int flags[MAXSTEP];
    int64_t B;
    int H, HS, fpt, full_tiles;
    RqsConst c;
};

template <int NSTEP, int KT>
__global__ __launch_bounds__(IT, 1) void made_rqs_inverse_kernel(const InvArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hlT = smem;                              // [HS][IR] final hidden state, unit-major
    float* part = hlT + g.HS * IR;                  // [2][MW][IR][32] per-wave partial parameter tiles, by feature parity
    float* prmb = part + 2 * MW * IR * 32;          // [2][IR][32] bulk parameters, by feature parity
    float* prm = prmb + 2 * IR * 32;                // [32][PSTR] wave 0's completed parameters
    float* bc = prm + 32 * PSTR;                    // [2][MAXSTEP][IR] activations of the unit just finished, by parity
    float* sh = bc + 2 * MAXSTEP * IR;              // [2][MAXSTEP][IR] exported running sums of the next unit, by parity

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar branches on the wave's role)
    const int64_t m0 = (int64_t)blockIdx.x * IR;
    const int H = g.H, HS = g.HS;
    const bool unit = tid < H;
    const int kcol = unit ? tid : 0;

    // Running sums of the hidden units, in the accumulator layout of v_mfma_f32_32x32x1_2b_f32: wave w holds units
    // 64w .. 64w+63 as two 32 x 32 (unit x sample) blocks per step; register r of lane l is
    //   unit 64w + 32 (r >> 4) + (r & 3) + 8 ((r & 15) >> 2) + 4 (l >> 5),  sample l & 31.
    f32x32 acc[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int u = 64 * wave + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * (lane >> 5);
            acc[s][r] = u < H ? g.bias[s][u] : 0.0f;
        }
    }
    for (int e = tid; e < HS * IR; e += IT) hlT[e] = 0.0f;
    for (int e = tid; e < 2 * MW * IR * 32; e += IT) part[e] = 0.0f;
    for (int e = tid; e < IR * 32; e += IT) prmb[e] = g.bf[e & 31];       // feature 0: bias alone
    if (tid < IR) {                                 // unit 0's running sums: nothing precedes it
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) sh[s * IR + tid] = g.bias[s][0];
    }

    // Per-role state in one register array (wave 0 never runs the bulk product, the others never invert splines):
    //   wave 0:     rs[0..NSTEP) = activations of the unit finished last round, rs[5] = next unit's weight record,
    //               rs[6] = final hidden value of unit j-1, rs[7] = next input z, rs[8] = running logdet sum,
    //               rs[9] = next feature's two newest-unit weight rows (lane = [row][column]),
    //               rs[10], rs[11] = next unit's context gates (residual blocks), rs[12] = its context term
    //   waves 1..7: rs[0..4) = next feature's bias, four columns (waves 1..4), rs[4..12) = next round's first two B operands
    float rs[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) rs[i] = 0.0f;
    if (wave == 0) {
        rs[7] = tid < IR ? g.z[((m0 + tid < g.B) ? m0 + tid : g.B - 1) * g.ldz] : 0.0f;
        rs[5] = g.dg[tid & 31];
        if (g.ctx0 && tid < IR) {
            const int64_t crow = ((m0 + tid < g.B) ? m0 + tid : g.B - 1) * g.ldc;
            rs[12] = g.ctx0[crow];
            if (g.gate[0]) rs[10] = g.gate[0][crow];
            if (g.gate[1]) rs[11] = g.gate[1][crow];
        }
        rs[9] = g.wl[lane];
    } else if (wave <= 4) {                 // parameter sums: columns 4 (tid & 7) .. + 3 of one sample
        const float4 nb = *reinterpret_cast<const float4*>(g.bf + 32 + ((tid * 4) & 31));
        rs[0] = nb.x; rs[1] = nb.y; rs[2] = nb.z; rs[3] = nb.w;
    }

    float wcur[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wcur[s] = g.wt[s][kcol];
    __syncthreads();

#ifdef NF_INV_STAMP
    // phase timing: per-wave cycle sums in LDS (registers are all spoken for)
    unsigned* stamps = reinterpret_cast<unsigned*>(sh + 2 * MAXSTEP * IR);
    if (threadIdx.x < IW * 8) stamps[threadIdx.x] = 0;
    __syncthreads();
    unsigned tlast = (unsigned)clock64();
#define STAMP(i_) { const unsigned n_ = (unsigned)clock64(); if ((threadIdx.x & 63) == 0) stamps[wave * 8 + i_] += n_ - tlast; tlast = n_; }
#else
#define STAMP(i_)
#endif
    for (int j = 0; j <= H; ++j) {
        const int par = j & 1;
        // Everything derived from the thread index is recomputed per round from an opaque copy: hoisted out of the
        // loop those values (addresses, row numbers) stay live across the spline inversion, the point of highest
        // register pressure, and were spilled to scratch - with a scratch reload in front of every MFMA group.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, arow = tid & 31, kh = (tid >> 5) & 1;
        const int kcol = tid < H ? tid : 0;
        if (wave == 0) {
            // ---- the critical path: feature j and hidden unit j
            // parameters of feature j = bulk (units <= j-3) + [h(j-1) h(j-2)] x their two weight rows: one
            // 32 x 32 x 2 MFMA on the bulk tile (lanes 0..31 hold h(j-1) of their sample, lanes 32..63 fetch h(j-2))
            {
                const float a = lane < IR ? rs[6] : (j >= 2 ? hlT[(j - 2) * IR + arow] : 0.0f);
                const float* pb = prmb + par * IR * 32 + 4 * kh * 32 + arow;
                f32x16 c;
#pragma unroll
                for (int i = 0; i < 16; ++i) c[i] = pb[((i & 3) + 8 * (i >> 2)) * 32];
                c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, rs[9], c, 0, 0, 0);
                if constexpr (KT <= KT_LINEAR) {        // [sample][parameter]: the family evaluators read unit stride
                    float* po = prm + 4 * kh * PSTR + arow;
#pragma unroll
                    for (int i = 0; i < 16; ++i) po[((i & 3) + 8 * (i >> 2)) * PSTR] = c[i];
                } else {                                // [parameter][sample]
                    float* po = prm + arow * PSTR + 4 * kh;
#pragma unroll
                    for (int i = 0; i < 16; ++i) po[(i & 3) + 8 * (i >> 2)] = c[i];
                }
                rs[9] = g.wl[(int64_t)(j < H ? j + 1 : H) * 64 + lane];      // next round's weight rows
            }
            STAMP(0)
            float y, ld;
            int bin;
            typedef const __attribute__((address_space(3))) float* lds_ptr;
            if constexpr (KT > 0) {
                // lanes 32..63 build the width knots of the sample of lane - 32 (see rqs_inverse_pair)
                const float zother = nf_xor32(rs[7]);
                const float zin = lane < IR ? rs[7] : zother;          // z of sample (lane & 31)
                rqs_inverse_pair<KT, PSTR, lds_ptr>((lds_ptr)(prm + arow), zin, g.c, lane >= IR, y, ld, bin);
            }
            if (lane < IR) {
                const int64_t row = (m0 + tid < g.B) ? m0 + tid : g.B - 1;
                const bool row_ok = m0 + tid < g.B;
#ifdef NF_NOEVAL
                y = prm[tid] + rs[7]; ld = prm[tid + PSTR]; bin = 0;
#else
                if constexpr (KT == KT_AFFINE) {
                    // scale = sigmoid(u + 2) + 1e-3, x = (z - shift) / scale (models/transforms.py:540-566; same
                    // arithmetic as affine_kernel in nf_spline.hip)
                    const float us = prm[tid], shift = prm[PSTR + tid];
                    const float scale = 1.0f / (1.0f + expf(-(us + 2.0f))) + 1e-3f;
                    y = (rs[7] - shift) / scale;
                    ld = -logf(scale);
                    bin = 0;
                } else if constexpr (KT == 0) {
                    rqs_eval<KT, true, PSTR, false, false, lds_ptr, false>((lds_ptr)(prm + tid), rs[7], g.c, y, ld, bin);
                } else if constexpr (KT == KT_LINEAR) {
                    lin_eval<true, lds_ptr>((lds_ptr)(prm + tid * PSTR), rs[7], g.c, y, ld, bin);
                } else if constexpr (KT == KT_QUADRATIC) {
                    quad_eval<true, lds_ptr>((lds_ptr)(prm + tid * PSTR), rs[7], g.c, y, ld, bin);
                } else if constexpr (KT == KT_CUBIC) {
                    cubic_eval<true, lds_ptr>((lds_ptr)(prm + tid * PSTR), rs[7], g.c, y, ld, bin);
                }
#endif
                if (row_ok) g.x[row * g.ldx + j] = y;
                rs[8] += ld;
                if ((j + 1) % g.fpt == 0) {
                    const int t = j / g.fpt;
                    if (row_ok && t < g.full_tiles) g.partial[(int64_t)t * g.B + row] = rs[8];
                    rs[8] = 0.0f;
                }
                STAMP(1)
                if (j < H) {
                    rs[7] = g.z[row * g.ldz + j + 1];
                    // unit j's record: weights on itself (lanes 0..7) and on unit j-1 (8..15), BatchNorm scale / shift (16.., 24..)
                    const int rec = __float_as_int(rs[5]);
                    // running sums of unit j: its owner's export (units <= j-2) + unit j-1's term
                    float cur[NSTEP];
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s)
                        cur[s] = fmaf(__int_as_float(__builtin_amdgcn_readlane(rec, 8 + s)), rs[s],
                                      sh[(par * MAXSTEP + s) * IR + tid]);
                    // unit j of every hidden layer, one layer after the other
                    float val[NSTEP];
                    rs[0] = y;
                    val[0] = fmaf(__int_as_float(__builtin_amdgcn_readlane(rec, 0)), y, cur[0]);
                    if (g.ctx0) val[0] += rs[12];
#pragma unroll
                    for (int s = 1; s < NSTEP; ++s) {
                        // input of step s: (eval-mode BatchNorm, folded to scale / shift, then) ReLU of unit j one step back
                        const float bn = fmaf(val[s - 1], __int_as_float(__builtin_amdgcn_readlane(rec, 16 + s)),
                                              __int_as_float(__builtin_amdgcn_readlane(rec, 24 + s)));
                        rs[s] = (g.flags[s] & F_RELU_IN) ? fmaxf(bn, 0.0f) : bn;
                        val[s] = fmaf(__int_as_float(__builtin_amdgcn_readlane(rec, s)), rs[s], cur[s]);
                        if (s >= 2 && (g.flags[s] & F_RESIDUAL)) {
                            // x + t, or with a context x + t * sigmoid(context layer) (made.py:279-283)
                            const float t = g.gate[s >= 4 ? 1 : 0] ? val[s] * rs[s >= 4 ? 11 : 10] : val[s];
                            val[s] = val[s >= 2 ? s - 2 : 0] + t;
                        }
                        if (g.flags[s] & F_RELU_OUT) val[s] = fmaxf(val[s], 0.0f);
                    }
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) bc[(par * MAXSTEP + s) * IR + tid] = rs[s];
                    hlT[j * IR + tid] = val[NSTEP - 1];
                    rs[6] = val[NSTEP - 1];
                    rs[5] = g.dg[(int64_t)(j + 1 < H ? j + 1 : H - 1) * 32 + (tid & 31)];
                    if (g.ctx0) {
                        const int64_t cnext = row * g.ldc + (j + 1 < H ? j + 1 : H - 1);
                        rs[12] = g.ctx0[cnext];
                        if (g.gate[0]) rs[10] = g.gate[0][cnext];
                        if (g.gate[1]) rs[11] = g.gate[1][cnext];
                    }
                }
                STAMP(2)
            }
        }
        if (j == H) break;

        // ---- waves 1..7: the B operands of this round's product first, the update below covers their latency
        const int NS = (j + SS - 1) / SS;
        const int64_t fo = first_superstep(j + 2 <= H ? j + 2 : H);
        float4 bq[NQ];
        if (wave > 0 && j + 2 <= H) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int S = wave - 1 + MW * (NP + i);
                if (S < NS) bq[i] = g.wf[(fo + S) * 64 + lane];
            }
        }

        STAMP(7)
        // Waves w and w + 4 share a SIMD.  Waves 1..3 own the low units, which are finished early, so they take
        // the matrix product first and their update (if any) second; waves 4..7 the other way round: one wave's
        // MFMAs run under the other's VALU work instead of both queueing on the same pipe.
        const bool product_first = wave >= 1 && wave <= 3;
        auto product = [&]() {
            // ---- feature j+2 over the hidden units 0..j-1 (units j, j+1 join as rank-1 terms in wave 0)
            if (wave > 0 && j + 2 <= H) {
                f32x16 d;
#pragma unroll
                for (int i = 0; i < 16; ++i) d[i] = 0.0f;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int S = wave - 1 + MW * i;
                    if (S < NS) {
                        const float* ap = hlT + (SS * S + 4 * kh) * IR + arow;
                        const float a0 = ap[0], a1 = ap[IR], a2 = ap[2 * IR], a3 = ap[3 * IR];
                        // the first NP super-steps' operands came in a round ahead (the latency of these loads is 1500+
                        // cycles: the weights live in MALL / HBM), the next NQ at the top of this round, the rest follow
                        // in place NQ super-steps behind their slot's previous user
                        float4 b;
                        if (i < NP) {
                            b = make_float4(rs[4 + 4 * (i % NP)], rs[5 + 4 * (i % NP)], rs[6 + 4 * (i % NP)], rs[7 + 4 * (i % NP)]);
                        } else {
                            b = bq[(i - NP) % NQ];
                            if (i + NQ < NI && S + NQ * MW < NS) bq[(i - NP) % NQ] = g.wf[(fo + S + NQ * MW) * 64 + lane];
                        }
                        d = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.x, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.y, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b.z, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b.w, d, 0, 0, 0);
                    }
                }
                float* pt = part + (par * MW + wave - 1) * IR * 32;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r = (i & 3) + 8 * (i >> 2) + 4 * kh;
                    pt[r * 32 + arow] = d[i];
                }
                if (j + 3 <= H) {                           // next round: feature j+3 over units 0..j
                    const int NS1 = (j + SS) / SS;
                    const int64_t o = first_superstep(j + 3);
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        const int S = wave - 1 + MW * i;
                        if (S < NS1) {
                            const float4 n = g.wf[(o + S) * 64 + lane];
                            rs[4 + 4 * i] = n.x; rs[5 + 4 * i] = n.y; rs[6 + 4 * i] = n.z; rs[7 + 4 * i] = n.w;
                        }
                    }
                }
            }
            STAMP(4)
        };
        auto sums = [&]() {
            // ---- bulk parameters of feature j+1: bias + last round's partial tiles, fixed order; waves 1..4,
            //      one sample and four columns per thread
            {
                const int e = (tid - 64) * 4;                       // [sample][column] element
                const float* pp = part + (par ^ 1) * MW * IR * 32 + e;
                float4 s = make_float4(rs[0], rs[1], rs[2], rs[3]);
#pragma unroll
                for (int w = 0; w < MW; ++w) {
                    const float4 t = *reinterpret_cast<const float4*>(pp + w * IR * 32);
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                *reinterpret_cast<float4*>(prmb + (par ^ 1) * IR * 32 + e) = s;
                const float4 nb = *reinterpret_cast<const float4*>(g.bf + (j + 2 <= H ? j + 2 : H) * 32 + (e & 31));
                rs[0] = nb.x; rs[1] = nb.y; rs[2] = nb.z; rs[3] = nb.w;
            }
        };
        if (product_first) {
            sums();                                     // (LDS only: covers the latency of the B operands)
            product();
        }
        // ---- everyone, one round behind: unit j-1's activations into the running sums of the units after it
        if (j > 0 && wave * 64 + 63 > j) {              // units <= j are finished or being finished by wave 0
            // one rank-1 update per hidden step: (this wave's 64 weights on unit j-1) x (its 32 activations), a
            // 2-block 32 x 32 x 1 MFMA each - the matrix pipe has room beside the bulk product, the VALU did not
            // (160 DPP FMAs per wave and round were 2500 cycles with two waves per SIMD)
            const float* b = bc + ((par ^ 1) * MAXSTEP) * IR + (tid & 31);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s)
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x1f32(wcur[s], b[s * IR], acc[s], 0, 0, 0);
        }
        if (wave * 64 + 63 > j + 1) {                   // (a wave whose units are all finished has no update next round:
            // each weight register is reloaded for the next round right after its use      wave 0 from round 62 on)
            const int64_t wrow = (int64_t)(j < H ? j : H - 1) * HS + kcol;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) wcur[s] = g.wt[s][wrow];
        }
        if (wave == ((j + 1) >> 6)) {                   // unit j+1's sums over units <= j-1, for wave 0's next round
            const int v = (j + 1) & 63, row = v & 31;
            const int r = 16 * (v >> 5) + (row & 3) + 4 * (row >> 3);       // its register; lanes of half (row >> 2) & 1
            float ex[NSTEP];
#define NF_EX(R_) case R_: _Pragma("unroll") for (int s = 0; s < NSTEP; ++s) ex[s] = acc[s][R_]; break;
            switch (r) {
                NF_EX(0) NF_EX(1) NF_EX(2) NF_EX(3) NF_EX(4) NF_EX(5) NF_EX(6) NF_EX(7) NF_EX(8) NF_EX(9) NF_EX(10)
                NF_EX(11) NF_EX(12) NF_EX(13) NF_EX(14) NF_EX(15) NF_EX(16) NF_EX(17) NF_EX(18) NF_EX(19) NF_EX(20)
                NF_EX(21) NF_EX(22) NF_EX(23) NF_EX(24) NF_EX(25) NF_EX(26) NF_EX(27) NF_EX(28) NF_EX(29) NF_EX(30)
                default:
#pragma unroll
                    for (int s = 0; s < NSTEP; ++s) ex[s] = acc[s][31];
            }
#undef NF_EX
            if (((tid >> 5) & 1) == ((row >> 2) & 1)) {
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) sh[((par ^ 1) * MAXSTEP + s) * IR + (tid & 31)] = ex[s];
            }
        }
        STAMP(3)
        if (wave >= 4) product();
        if (wave == 4) sums();
        STAMP(5)
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait out every global load issued a round
        // ahead on purpose (next weights, B operands, z) - 1500 cycles per round when measured
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(6)
    }
#ifdef NF_INV_STAMP
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)
        printf("wave %d: params %u | eval %u | unit/export %u | loads %u | update %u | mfma %u | sums %u | barrier %u (cycles per round)\n", wave,
               stamps[wave * 8] / H, stamps[wave * 8 + 1] / H, stamps[wave * 8 + 2] / H, stamps[wave * 8 + 7] / H, stamps[wave * 8 + 3] / H,
               stamps[wave * 8 + 4] / H, stamps[wave * 8 + 5] / H, stamps[wave * 8 + 6] / H);
#endif

    // final hidden state for the features past H (one windowed launch of the fused layer kernel)
    __syncthreads();
    if (unit) {
        for (int r = 0; r < IR; ++r)
            if (m0 + r < g.B) g.hid[(m0 + r) * g.ldh + tid] = hlT[tid * IR + r];
    }
}

inline int hs_of(int H) { return (H + 63) / 64 * 64; }
inline int64_t total_supersteps(int H) { return first_superstep(H + 1); }

template <int NSTEP>
int launch_steps(const InvArgs& g, int K, unsigned grid, size_t lds, hipStream_t s) {
    auto go = [&](auto kern) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(IT), lds, s, g);
        return nf_launch_status();
    };
    if (K == KT_AFFINE) return go(made_rqs_inverse_kernel<NSTEP, KT_AFFINE>);
    if (K == KT_LINEAR) return go(made_rqs_inverse_kernel<NSTEP, KT_LINEAR>);
    if (K == KT_QUADRATIC) return go(made_rqs_inverse_kernel<NSTEP, KT_QUADRATIC>);
    if (K == KT_CUBIC) return go(made_rqs_inverse_kernel<NSTEP, KT_CUBIC>);
    return K == 8 ? go(made_rqs_inverse_kernel<NSTEP, 8>) : go(made_rqs_inverse_kernel<NSTEP, 0>);
}

}  // namespace

extern "C" int nf_made_rqs_inverse_rows_supported(int features, int hidden, int n_steps, int K, int tails) {
    const int P = tails ? 3 * K - 1 : 3 * K + 1;
    return features >= 2 && hidden >= 1 && hidden <= IT && hidden <= features - 1 && hidden % 8 == 0 && n_steps >= 2 &&
                   n_steps <= MAXSTEP && K >= 1 && P <= 32
               ? 1 : 0;
}

extern "C" size_t nf_made_inverse_blob_floats(int hidden, int n_steps) {
    if (hidden <= 0 || hidden > IT || n_steps < 1 || n_steps > MAXSTEP) return 0;
    const size_t H = hidden, HS = hs_of(hidden);
    return n_steps * (H + 1) * HS + 3 * (H + 1) * 32 + H * 32 + (size_t)total_supersteps(hidden) * 256;
}

static int inverse_rows_impl(const float* blob, int hidden, int n_steps, const int* step_flags, const float* z,
                             int64_t z_row_stride, float* x, int64_t x_row_stride, float* hidden_out,
                             int64_t h_row_stride, const float* context_term, const float* gate0, const float* gate1,
                             int64_t context_row_stride, float* workspace, int64_t B, int features, int K,
                             const RqsConst& c, void* stream) {
    if (B == 0) return NF_OK;
    if (!blob || !step_flags || !z || !x || !hidden_out) return NF_ERR_NULL;
    if ((gate0 || gate1) && !context_term) return NF_ERR_NULL;
    if (context_term && context_row_stride < hidden) return NF_ERR_LAYOUT;
    if (z_row_stride < features || x_row_stride < features || h_row_stride < hidden || ((uintptr_t)blob & 15) != 0)
        return NF_ERR_LAYOUT;
    InvArgs g;
    g.c = c;
    g.fpt = 256 / g.c.P;
    const int H = hidden, HS = hs_of(hidden);
    const float* p = blob;
    for (int s = 0; s < MAXSTEP; ++s) {
        g.wt[s] = nullptr; g.bias[s] = nullptr; g.flags[s] = 0;
        if (s < n_steps) {
            g.wt[s] = p;
            g.bias[s] = p + (size_t)H * HS;
            g.flags[s] = step_flags[s];
            p += (size_t)(H + 1) * HS;
        }
    }
    g.bf = p; p += (size_t)(H + 1) * 32;
    g.wl = p; p += (size_t)(H + 1) * 64;
    g.dg = p; p += (size_t)H * 32;
    g.wf = reinterpret_cast<const float4*>(p);
    g.z = z; g.ldz = z_row_stride; g.x = x; g.ldx = x_row_stride; g.hid = hidden_out; g.ldh = h_row_stride;
    g.partial = workspace; g.B = B; g.H = H; g.HS = HS;
    g.ctx0 = context_term; g.gate[0] = gate0; g.gate[1] = gate1; g.ldc = context_row_stride;
    g.full_tiles = workspace ? (H + 1) / g.fpt : 0;
    const int64_t grid = (B + IR - 1) / IR;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    const size_t lds = sizeof(float) * ((size_t)HS * IR + 2 * MW * IR * 32 + 2 * IR * 32 + 32 * PSTR + 4 * MAXSTEP * IR + IW * 8);
    hipStream_t st = (hipStream_t)stream;
    switch (n_steps) {
        case 2: return launch_steps<2>(g, K, (unsigned)grid, lds, st);
        case 3: return launch_steps<3>(g, K, (unsigned)grid, lds, st);
        case 4: return launch_steps<4>(g, K, (unsigned)grid, lds, st);
        case 5: return launch_steps<5>(g, K, (unsigned)grid, lds, st);
    }
    return NF_ERR_SHAPE;
}

extern "C" int nf_made_rqs_inverse_rows(const float* blob, int hidden, int n_steps, const int* step_flags,
                                        const float* z, int64_t z_row_stride, float* x, int64_t x_row_stride,
                                        float* hidden_out, int64_t h_row_stride, const float* context_term,
                                        const float* gate0, const float* gate1, int64_t context_row_stride,
                                        float* workspace, size_t workspace_bytes, int64_t B, int features, int K, int tails,
                                        double tail_bound, double min_bin_width, double min_bin_height,
                                        double min_derivative, void* stream) {
    if (B < 0 || !nf_made_rqs_inverse_rows_supported(features, hidden, n_steps, K, tails)) return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    if (B > 0 && !workspace) return NF_ERR_NULL;
    const RqsConst c = nf_make_rqs_const(K, tails, tail_bound, min_bin_width, min_bin_height, min_derivative, 1.0);
    const int fpt = 256 / c.P, all_tiles = (features + fpt - 1) / fpt;
    if (workspace_bytes < sizeof(float) * (size_t)all_tiles * (size_t)B) return NF_ERR_SCRATCH;
    return inverse_rows_impl(blob, hidden, n_steps, step_flags, z, z_row_stride, x, x_row_stride, hidden_out,
                             h_row_stride, context_term, gate0, gate1, context_row_stride, workspace, B, features, K, c,
                             stream);
}

extern "C" int nf_made_affine_inverse_rows_supported(int features, int hidden, int n_steps) {
    return features >= 2 && hidden >= 1 && hidden <= IT && hidden <= features - 1 && hidden % 8 == 0 && n_steps >= 2 &&
                   n_steps <= MAXSTEP ? 1 : 0;
}

extern "C" int nf_made_affine_inverse_rows(const float* blob, int hidden, int n_steps, const int* step_flags,
                                           const float* z, int64_t z_row_stride, float* x, int64_t x_row_stride,
                                           float* hidden_out, int64_t h_row_stride, const float* context_term,
                                           const float* gate0, const float* gate1, int64_t context_row_stride,
                                           int64_t B, int features, void* stream) {
    if (B < 0 || !nf_made_affine_inverse_rows_supported(features, hidden, n_steps)) return NF_ERR_SHAPE;
    RqsConst c = nf_make_rqs_const(1, 1, 1.0, 0.0, 0.0, 0.0, 1.0);
    c.P = 2;                                    // (scale, shift) interleaved per feature
    return inverse_rows_impl(blob, hidden, n_steps, step_flags, z, z_row_stride, x, x_row_stride, hidden_out,
                             h_row_stride, context_term, gate0, gate1, context_row_stride, nullptr, B, features,
                             KT_AFFINE, c, stream);
}

// family: 1 linear (P = K), 2 quadratic (P = 2K+1, or 2K-1 with linear tails), 3 cubic (P = 2K+2, on [-bound, bound])
static int family_params(int family, int K, int tails) {
    if (K < 1) return 0;
    switch (family) {
        case 1: return K;
        case 2: return K < 2 ? 0 : (tails ? 2 * K - 1 : 2 * K + 1);
        case 3: return 2 * K + 2;
    }
    return 0;
}

extern "C" int nf_made_family_inverse_rows_supported(int family, int features, int hidden, int n_steps, int K, int tails) {
    const int P = family_params(family, K, tails);
    return P >= 1 && P <= 32 && nf_made_affine_inverse_rows_supported(features, hidden, n_steps) ? 1 : 0;
}

extern "C" int nf_made_family_inverse_rows(int family, const float* blob, int hidden, int n_steps, const int* step_flags,
                                           const float* z, int64_t z_row_stride, float* x, int64_t x_row_stride,
                                           float* hidden_out, int64_t h_row_stride, const float* context_term,
                                           const float* gate0, const float* gate1, int64_t context_row_stride,
                                           int64_t B, int features, int K, int tails, double tail_bound,
                                           double min_bin_width, double min_bin_height, double pre_div, void* stream) {
    if (B < 0 || !nf_made_family_inverse_rows_supported(family, features, hidden, n_steps, K, tails) || !(pre_div > 0.0))
        return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    // the constants of the standalone family kernels (nf_spline.hip, family_fwd)
    RqsConst c = nf_make_rqs_const(K, tails, tail_bound, family == 1 ? 0.0 : min_bin_width,
                                   family == 1 ? 0.0 : min_bin_height, 1e-3, family == 3 ? pre_div : 1.0);
    c.P = family_params(family, K, tails);
    if (family == 3) {
        c.left = (float)-tail_bound;
        c.right = (float)tail_bound;
        c.span = (float)(2.0 * tail_bound);
    }
    return inverse_rows_impl(blob, hidden, n_steps, step_flags, z, z_row_stride, x, x_row_stride, hidden_out,
                             h_row_stride, context_term, gate0, gate1, context_row_stride, nullptr, B, features,
                             family == 1 ? KT_LINEAR : family == 2 ? KT_QUADRATIC : KT_CUBIC, c, stream);
}
