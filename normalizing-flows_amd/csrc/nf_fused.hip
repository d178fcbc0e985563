// Final MADE layer fused with the rational-quadratic spline (gfx950).
//
//   params[b, d*P + j] = h[b,:] . Wm[d*P + j, :] + bias      (models/made.py:150-152, :358)
//   y[b,d], ld[b,d]    = RQ spline(x[b,d]; params[b, d*P .. d*P+P-1])   (transforms.py:819-860,
//                                                                        rational_quadratic.py:13-149)
//   logabsdet[b]       = sum_d ld[b,d]                                   (helper_functions.py:393-398)
//
// The reference materialises `params` ([B, D*P] floats: 72 KB per sample at D=784, K=8), the largest HBM
// object of the whole path, only to consume it once.  Here a workgroup owns 64 samples x FPT features
// (FPT = 256 / P whole features, 11 at P=23, so the 256-column weight tile holds only whole parameter
// rows), accumulates the 64 x 256 tile on the matrix cores, drops it into LDS 32 samples at a time
// (aliasing the operand staging buffers) and evaluates the spline from there: `params` never reaches
// HBM.  Per-sample logdet partial sums go to a [n_tiles, B] scratch (4 bytes per sample per tile) that a
// second tiny kernel adds up in a fixed order - deterministic, no float atomics.
//
// Occupancy is the design point: 256 threads (one wave per SIMD), 128 VGPRs and ~40 KB of LDS per
// workgroup put FOUR independent workgroups on a CU.  A workgroup's spline epilogue (VALU, ~20 % of its
// lifetime) and its barrier phases issue no MFMA; with four barrier domains per SIMD the other three keep
// the matrix pipe busy meanwhile (measured with the NF_STAMP phase clocks: two 512-thread workgroups per
// CU left the pipe idle 30 % of the time, benchmarks/diag_stamp.py).
#include <stdlib.h>
#include <string.h>
#include "nf_common.h"
#include "nf_rqs_eval.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int FM = 64, FN = 256;        // block tile: 64 samples x 256 parameter columns
constexpr int FT = 256;                 // threads: 4 waves side by side, 64 x 64 outputs each (2x2 MFMA tiles)
#ifndef NF_FBK
#define NF_FBK 16
#endif
#ifndef NF_FOCC
#define NF_FOCC 4
#endif
constexpr int BK = NF_FBK, LDA = BK + 4; // fp32 staging: K tile and padded LDS row (floats)
constexpr int BK16 = 32, LDA16 = BK16 + 8;   // bf16 staging (elements); both rows are 80 bytes: conflict-free b128
constexpr int KPAD = 64;                // packed weight rows are padded to this many K elements
constexpr int HR = 32;                  // epilogue: samples per pass through the LDS parameter tile
constexpr int LDT = HR + 4;             // parameter tile [columns][HR + 4], column-major: a lane's 4 consecutive
                                        // accumulator rows go out as one ds_write_b128, conflict-free

constexpr int XOT = 72;            // row stride (floats) of the split GEMM's output staging tile

struct FusedArgs {
    const float* h;          // [M, K] conditioner hidden state
    int64_t ldh;
    const void* w;           // packed [n_tiles*256][kpad] (f32 or bf16)
    int kpad;
    const int32_t* ktab;     // [n_tiles][2]: live K range of each column tile in 16-wide K steps
    const float* bias;       // packed [n_tiles*256]
    const float* x;          // [M, D] transform inputs
    int64_t ldx;
    float* y;
    int64_t ldy;
    float* partial;          // [n_tiles][M]
    int32_t* bins;           // nullable [M, D]
    const int32_t* feat_index;   // nullable [D]: column of x / y that holds feature f (coupling layers)
    const unsigned* wmax;    // f16x2: bit pattern of max |packed weight| before scaling (kept after the K-range table)
    const unsigned* hmax;    // f16x2: bit pattern of max |A operand| of this launch (workspace slot or the producer's)
    int hmax_rows;           // t kernels: hmax is an array, one entry per group of 64 rows (planes from nf_made_hidden_planes_fwd)
    const float* residual;   // split hidden-layer GEMM: nullable [M, N] added to the output
    int64_t ldr;
    const float* add2;       // split hidden-layer GEMM, relu_out == 2 only: nullable [M, N] added AFTER the gate (the skip
    int64_t ldadd2;          // connection's gradient joining an input-gradient GEMM: no add pass, no max pass over the sum)
    int N, relu_out;         // split hidden-layer GEMM: output columns, ReLU on the output
    unsigned* out_max;       // split hidden-layer GEMM: nullable slot that receives max |output| (next layer's hmax)
    const void* hp;          // split-operand full launches: h pre-split into three bf16 term planes [3][M][hp_ld]
    int64_t hp_ld;
    int64_t M;
    int K, D;
    int fpt;                 // features per tile
    unsigned fpt_magic;      // ceil(2^20 / fpt): i / fpt == (i * fpt_magic) >> 20 for i < 2^13 (the t kernels' tile indices)
    int n_tiles;             // column tiles computed by this launch, starting at tn0 (window)
    int tn0;
    int feat_limit;          // features >= feat_limit are left untouched (incremental inverse); D = all
    int vec_out;             // split GEMM: y / residual rows take 16-byte accesses (LDS-staged epilogue)
    int64_t m_tiles;
    int group_m;             // row tiles that share a weight tile back to back (L2 reuse)
    int k_reverse;           // split GEMM: walk a tile's live K range from its end down (input-gradient launches)
    float* params_out;       // made_rqs_kernel: nullable [M, D * P], the conditioner output a training forward keeps
    int params_bf16;         // ... as bf16 (params_out is then a __bf16 array; ldp in elements either way)
    int64_t ldp;
    float unfold;            // ... whose logit columns are multiplied by this (the pack folded log2(e) / logit_div in)
    RqsConst c;
};

__device__ __forceinline__ void tile_of_block(const FusedArgs& g, int64_t& tm, int& tn) {
    // (32-bit arithmetic: the host refuses grids of 2^31 workgroups or more; the 64-bit divisions this replaced were
    // some 2000 cycles of every workgroup's prologue)
    const unsigned nwg = gridDim.x;
    const unsigned bid = blockIdx.x;
    const unsigned q = nwg / 8, r = nwg % 8;
    const unsigned xcd = bid % 8;
    const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    const unsigned per_group = (unsigned)g.group_m * (unsigned)g.n_tiles;
    const unsigned grp = wg / per_group;
    const unsigned first_m = grp * g.group_m;
    const unsigned gm = min((unsigned)g.group_m, (unsigned)g.m_tiles - first_m);
    const unsigned in = wg - grp * per_group;
    const unsigned col = in / gm;
    tm = first_m + (in - col * gm);
    // heaviest column tiles first: with triangular masks the K extent grows with the tile index, and
    // starting the long workgroups early leaves the short ones to fill the tail of the launch
    tn = g.n_tiles - 1 - (int)col + g.tn0;
}

// ---- operand staging (fp32 / bf16 flavours share the kernel skeleton through this trait) -------------
template <bool BF16>
struct Stage;

template <>
struct Stage<false> {
    static constexpr int kBK = BK;
    // 16-byte chunks: CPR per row, the 256 threads cover RPP rows per pass: NA passes for A, NB for B
    static constexpr int CPR = BK / 4, RPP = FT / CPR, NA = FM / RPP, NB = FN / RPP;
    const float* a[NA];
    const float* b0;
    int64_t bstep;
    float4 ra[NA], rb[NB];
    float *as, *bs;
    int skc, K;

    __device__ __forceinline__ void init(const FusedArgs& g, char* lds, int64_t m0, int tn, int tid) {
        const int srow = tid / CPR;
        skc = (tid % CPR) * 4;
        K = g.K;
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = g.h + min(m0 + srow + i * RPP, g.M - 1) * g.ldh;
        b0 = (const float*)g.w + (int64_t)(tn * FN + srow) * g.kpad + skc;
        bstep = (int64_t)RPP * g.kpad;
        as = (float*)lds + srow * LDA + skc;
        bs = (float*)lds + FM * LDA + srow * LDA + skc;
    }
    __device__ __forceinline__ void load(int kt) {
        const int k = kt * BK + skc;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(a[i] + min(k, K - 4));   // K % 8 == 0
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4*>(b0 + i * bstep + kt * BK);
    }
    __device__ __forceinline__ void store(int kt) {
        const bool live = kt * BK + skc < K;                            // whole chunk in or out (K % 4 == 0)
#pragma unroll
        for (int i = 0; i < NA; ++i)
            *reinterpret_cast<float4*>(as + i * RPP * LDA) = live ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(bs + i * RPP * LDA) = rb[i];
    }
    // BK/2 x 4 MFMAs (32x32x2 f32) on the staged tile: within an 8-wide K chunk lanes 0-31 own k 0..3 and
    // lanes 32-63 own k 4..7 of both operands (a reordering of the K sum).
    __device__ __forceinline__ void mma(const char* lds, int wc, int lane, f32x16 (&acc)[2][2]) const {
        const float* As = (const float*)lds;
        const float* Bs = As + FM * LDA;
        const int frow = lane & 31, fk = (lane >> 5) * 4;
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4*>(&As[(i * 32 + frow) * LDA + kc * 8 + fk]);
                b[i] = *reinterpret_cast<const float4*>(&Bs[(wc * 64 + i * 32 + frow) * LDA + kc * 8 + fk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
};

template <>
struct Stage<true> {
    static constexpr int kBK = BK16;
    // 16-byte (8 x bf16) chunks: A 64 rows x 4 = 256 -> 1 per thread (two float4 loads, converted on the
    // way into LDS), B 256 rows x 4 = 1024 -> 4 per thread
    const float* a0;
    const __bf16 *b0, *b1, *b2, *b3;
    float4 ra0l, ra0h;
    bf16x8 rb0, rb1, rb2, rb3;
    __bf16 *as, *bs;
    int skc, K;

    __device__ __forceinline__ void init(const FusedArgs& g, char* lds, int64_t m0, int tn, int tid) {
        const int srow = tid >> 2;
        skc = (tid & 3) * 8;
        K = g.K;
        a0 = g.h + min(m0 + srow, g.M - 1) * g.ldh;
        b0 = (const __bf16*)g.w + (int64_t)(tn * FN + srow) * g.kpad + skc;
        b1 = b0 + (int64_t)64 * g.kpad;
        b2 = b0 + (int64_t)128 * g.kpad;
        b3 = b0 + (int64_t)192 * g.kpad;
        as = (__bf16*)lds + srow * LDA16 + skc;
        bs = (__bf16*)lds + FM * LDA16 + srow * LDA16 + skc;
    }
    __device__ __forceinline__ void load(int kt) {
        const int kk = min(kt * BK16 + skc, K - 8);                     // K % 8 == 0
        ra0l = *reinterpret_cast<const float4*>(a0 + kk);
        ra0h = *reinterpret_cast<const float4*>(a0 + kk + 4);
        rb0 = *reinterpret_cast<const bf16x8*>(b0 + kt * BK16);
        rb1 = *reinterpret_cast<const bf16x8*>(b1 + kt * BK16);
        rb2 = *reinterpret_cast<const bf16x8*>(b2 + kt * BK16);
        rb3 = *reinterpret_cast<const bf16x8*>(b3 + kt * BK16);
    }
    __device__ __forceinline__ void store(int kt) {
        const bool live = kt * BK16 + skc < K;                          // whole chunk in or out (K % 8 == 0)
        const float v[8] = {ra0l.x, ra0l.y, ra0l.z, ra0l.w, ra0h.x, ra0h.y, ra0h.z, ra0h.w};
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)(live ? v[j] : 0.f);
        *reinterpret_cast<bf16x8*>(as) = o;
        *reinterpret_cast<bf16x8*>(bs) = rb0;
        *reinterpret_cast<bf16x8*>(bs + 64 * LDA16) = rb1;
        *reinterpret_cast<bf16x8*>(bs + 128 * LDA16) = rb2;
        *reinterpret_cast<bf16x8*>(bs + 192 * LDA16) = rb3;
    }
    __device__ __forceinline__ void mma(const char* lds, int wc, int lane, f32x16 (&acc)[2][2]) const {
        const __bf16* As = (const __bf16*)lds;
        const __bf16* Bs = As + FM * LDA16;
        const int frow = lane & 31, fk = (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(&As[(i * 32 + frow) * LDA16 + ks * 16 + fk]);
                b[i] = *reinterpret_cast<const bf16x8*>(&Bs[(wc * 64 + i * 32 + frow) * LDA16 + ks * 16 + fk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

// One 32-row block of a wave's accumulators (its two column tiles) into the column-major LDS tile:
// lane holds column (lane & 31) and rows 8q + 4*(lane >> 5) + 0..3, q = 0..3, of each MFMA tile.
__device__ __forceinline__ void drop_block(float* tile, int ncol, int wc, int lane, const f32x16& a0, const f32x16& a1) {
    const int col = wc * 64 + (lane & 31);
    float* tp = tile + col * LDT + 4 * (lane >> 5);
    if (col < ncol) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(tp + 8 * q) = make_float4(a0[4 * q], a0[4 * q + 1], a0[4 * q + 2], a0[4 * q + 3]);
    }
    if (col + 32 < ncol) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(tp + 32 * LDT + 8 * q) = make_float4(a1[4 * q], a1[4 * q + 1], a1[4 * q + 2], a1[4 * q + 3]);
    }
}

// Barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access of the wave
// (vmcnt(0)) - behind the parameter stores of a training forward that is a full drain of 128 KB per workgroup in front
// of the spline evaluation they are meant to overlap with.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// KEEP: the training forward - the parameters also go to g.params_out (see below)
template <bool BF16, int KT, bool INV, bool KEEP = false>
__global__ __launch_bounds__(FT, NF_FOCC) void made_rqs_kernel(const FusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int fpt = g.fpt, P = g.c.P;
    const int ncol = fpt * P;                        // live parameter columns of a tile (<= 256)
    // [0, ncol*LDT*4): operand staging during the K loop (20 KB), then the epilogue parameter tile
    float* tile = (float*)lds;
    float* xs = tile + max(ncol * LDT, (FM + FN) * LDA);    // [64][fpt] inputs, overwritten by outputs
    float* ldp = xs + FM * fpt;                      // [4 waves][64] logdet sums of the features a wave evaluated

    int64_t tm;
    int tn;
    tile_of_block(g, tm, tn);
    const int64_t m0 = tm * FM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wc = tid >> 6;
    const int f0 = tn * fpt;
    const int fv = min(fpt, g.feat_limit - f0);      // features of this tile to evaluate (>= 1 by construction)

    // inputs of the tile: consecutive threads take consecutive features of a row
    {
        // (all loads in flight at once: as a rolled loop each waited in front of its LDS store)
        constexpr int NX = (FM * (FN / 11) + FT - 1) / FT;          // fpt <= 256 / 11 (K >= 4)
        float xv[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int i = tid + q * FT;
            const int r = i / fpt, f = i - r * fpt;
            xv[q] = 0.f;
            if (i < FM * fpt && m0 + r < g.M && f < fv)
                xv[q] = g.x[(m0 + r) * g.ldx + (g.feat_index ? g.feat_index[f0 + f] : f0 + f)];
        }
#pragma unroll
        for (int q = 0; q < NX; ++q)
            if (tid + q * FT < FM * fpt) xs[tid + q * FT] = xv[q];
    }

    constexpr int kBK = Stage<BF16>::kBK;
    constexpr int kq = kBK / 16;                     // ktab counts 16-wide K steps
    const int kt_begin = g.ktab[2 * tn] / kq;
    const int kt_end = min((g.ktab[2 * tn + 1] + kq - 1) / kq, (g.K + kBK - 1) / kBK);

    // accumulators start at the bias of their column (a lane's 16 values of one MFMA tile share a column)
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float bv = g.bias[tn * FN + wc * 64 + j * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }

#ifdef NF_STAMP
    long long tA = 0, tS = 0, tB = 0, tL = 0, tM = 0, tE = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0, t0 = clock64(), t1;
    const long long tstart = t0;
#define STAMP(v) __builtin_amdgcn_sched_barrier(0); t1 = clock64(); v += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0);
#else
#define STAMP(v)
#endif
    Stage<BF16> st;
    st.init(g, lds, m0, tn, tid);
    if (kt_begin < kt_end) st.load(kt_begin);
    STAMP(tL)
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        __syncthreads();
        STAMP(tA)
        st.store(kt);
        STAMP(tS)
        __syncthreads();
        STAMP(tB)
#ifndef NF_KO_LOAD
        st.load(min(kt + 1, kt_end - 1));            // prefetch under the MFMAs (last one is a re-read)
#endif
        __builtin_amdgcn_sched_barrier(0);
        STAMP(tL)
        st.mma(lds, wc, lane, acc);
        STAMP(tM)
    }

    // training forward: the parameters (= the accumulators, bias included) also go to memory, straight from the MFMA
    // layout - a lane's 16 values of a tile are 16 rows of ONE column, a store instruction writes two rows x 32
    // consecutive columns.  They are issued here so that they drain under the spline evaluation below (whose barriers
    // do not wait for them).  The pack folded log2(e) / logit_div into the logit rows: undone IN PLACE first (the
    // evaluation below then takes natural-log logits) - a store's data register must not be rewritten while the store
    // is in flight, so storing scaled temporaries made hipcc wait for every store before reusing them.
    if (KEEP) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wc * 64 + j * 32 + (lane & 31);
            const float uf = (col % P) < 2 * g.c.K ? g.unfold : 1.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= uf;
        }
        const bool whole = m0 + FM <= g.M;           // (uniform) every row of the tile exists: no per-store test
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wc * 64 + j * 32 + (lane & 31);
            if (col >= fv * P) continue;
            // this lane's rows are m0 + 4 (lane >> 5) + {0..3} + 8 {0..3} + 32 i: one base pointer, constant row offsets
            const int64_t po = (int64_t)f0 * P + col + (m0 + 4 * (lane >> 5)) * g.ldp;
            if (g.params_bf16) {
                __bf16* pc = (__bf16*)g.params_out + po;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = i * HR + (r & 3) + 8 * (r >> 2);
                        if (whole || m0 + k + 4 * (lane >> 5) < g.M) pc[k * g.ldp] = (__bf16)acc[i][j][r];
                    }
            } else {
                float* pc = g.params_out + po;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = i * HR + (r & 3) + 8 * (r >> 2);
                        if (whole || m0 + k + 4 * (lane >> 5) < g.M) pc[k * g.ldp] = acc[i][j][r];
                    }
            }
        }
    }

    // ---- epilogue: the two 32-sample row blocks of the MFMA tiles, one after the other ---------------
    const int hrow = lane & 31, hsel = lane >> 5;    // eval: lane = sample row of the pass, two features per wave
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        lds_only_barrier();                          // staging / previous pass fully consumed
        STAMP(e0)
        if (half == 0) drop_block(tile, ncol, wc, lane, acc[0][0], acc[0][1]);
        else drop_block(tile, ncol, wc, lane, acc[1][0], acc[1][1]);
        lds_only_barrier();
        STAMP(e1)
        float lsum = 0.f;
        const int r = half * HR + hrow;
        // pass `half` starts its feature pairs at wave 2*half, so the waves that take a second round differ
        for (int e = ((wc + 2 * half) & 3) * 2 + hsel; e < fv; e += 2 * (FT / 64)) {
            float yv, ld;
            int bin;
#ifdef NF_KO_EPI
            yv = tile[e * P * LDT + hrow]; ld = yv; bin = 0;
#else
            rqs_eval<KT, INV, LDT, !KEEP, false, const float*, false>(tile + e * P * LDT + hrow, xs[r * fpt + e], g.c, yv, ld, bin);
#endif
            xs[r * fpt + e] = yv;
            lsum += ld;
#ifndef NF_STAMP
            if (g.bins && m0 + r < g.M) g.bins[(m0 + r) * (int64_t)g.D + f0 + e] = bin;
#endif
        }
        lsum += __shfl_xor(lsum, 32, 64);
        if (hsel == 0) ldp[wc * FM + r] = lsum;
        STAMP(e2)
    }
    lds_only_barrier();
    if (tid < FM) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < FT / 64; ++w) s += ldp[w * FM + tid];          // fixed order: deterministic
        if (m0 + tid < g.M) g.partial[(int64_t)tn * g.M + m0 + tid] = s;
    }
    STAMP(e3)
    float ymax = 0.f;
    for (int i = tid; i < FM * fpt; i += FT) {
        const int r = i / fpt, f = i - r * fpt;
        if (m0 + r < g.M && f < fv) {
            const float v = xs[i];
            g.y[(m0 + r) * g.ldy + (g.feat_index ? g.feat_index[f0 + f] : f0 + f)] = v;
            const float a = fabsf(v);
            ymax = fmaxf(ymax, a < INFINITY ? a : 0.f);
        }
    }
    if (g.out_max) {                                 // max |y|: the operand scale of whoever reads y next (see below)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
        if (lane == 0 && __float_as_uint(ymax) > __builtin_nontemporal_load(g.out_max)) atomicMax(g.out_max, __float_as_uint(ymax));
    }
#ifdef NF_STAMP
    STAMP(tE)
    if (g.bins && lane == 0) {
        long long* dbg = (long long*)g.bins + ((int64_t)blockIdx.x * (FT / 64) + wc) * 16;
        dbg[0] = tA; dbg[1] = tS; dbg[2] = tB; dbg[3] = tL; dbg[4] = tM; dbg[5] = tE; dbg[6] = tstart;
        dbg[7] = kt_end - kt_begin; dbg[8] = e0; dbg[9] = e1; dbg[10] = e2; dbg[11] = e3;
    }
#endif
}

// One row per lane, 64 rows per workgroup (32768 rows: 512 workgroups, not 128); the per-tile partial sums are added in
// tile order (deterministic), eight loads in flight at a time - as a plain loop every load waited for the add before it.
constexpr int kRowsumThreads = 64;
__global__ __launch_bounds__(kRowsumThreads) void rowsum_kernel(const float* __restrict__ partial, int n_tiles, int64_t M,
                                                                float* __restrict__ lad, int accumulate) {
    const int64_t r = (int64_t)blockIdx.x * kRowsumThreads + threadIdx.x;
    if (r >= M) return;
    float s = 0.f;
    int t = 0;
    for (; t + 8 <= n_tiles; t += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = partial[(int64_t)(t + i) * M + r];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; t < n_tiles; ++t) s += partial[(int64_t)t * M + r];
    lad[r] = accumulate ? lad[r] + s : s;
}

// dst row pr of the packed matrix = src row tn*src_rows + j for j < src_rows (and in range), else zero
// Rows of the width / height logits (parameter index < 2K within a feature) are folded with log2(e) / logit_div:
// the kernel's softmax then runs on the hardware exp2 directly, softmax(u / div) = exp2(u' - max u') / sum
// (logit_div is the coupling layers' sqrt(hidden_features), transforms.py:221-229; 1 on the autoregressive path).
template <typename T>
__global__ void pack_fused_kernel(const float* __restrict__ W, const float* __restrict__ mask,
                                  const float* __restrict__ bias, int out, int in, int src_rows, int kpad,
                                  int64_t total, int P, int n_logits, float logit_fold, T* __restrict__ packed,
                                  float* __restrict__ bias_packed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pr = i / kpad;
        const int k = (int)(i - pr * kpad);
        const int tn = (int)(pr / FN), j = (int)(pr % FN);
        const int64_t n = (int64_t)tn * src_rows + j;
        const bool live = j < src_rows && n < out;
        const float fold = (j % P) < n_logits ? logit_fold : 1.0f;
        float v = 0.0f;
        if (live && k < in) {
            v = W[n * in + k];
            if (mask) v *= mask[n * in + k];
            v *= fold;
        }
        packed[i] = (T)v;
        if (k == 0) bias_packed[pr] = (live && bias) ? bias[n] * fold : 0.0f;
    }
}

// K-range table of the packed tiles: [tile][first, one-past-last 16-wide K step with an unmasked weight].  A function of
// the mask only (built at the first pack of a layer, kept across repacks).  Three small launches: raw column bounds
// reset, a scan with KTAB_SLICES workgroups per tile whose consecutive threads read consecutive addresses in either
// orientation (one workgroup per tile took 0.56-0.73 ms on the final layer and its transpose), and the conversion to
// steps.
constexpr int KTAB_SLICES = 128;
__global__ void ktab_reset_kernel(int32_t* __restrict__ ktab, int nt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt) { ktab[2 * i] = 0x7fffffff; ktab[2 * i + 1] = -1; }
}

__global__ __launch_bounds__(256) void ktab_scan_kernel(const float* __restrict__ mask, int out, int in, int src_rows,
                                                        int32_t* __restrict__ ktab, int transposed) {
    const int tn = blockIdx.x, slice = blockIdx.y, t = threadIdx.x;
    const int row0 = tn * src_rows;
    const int rows = min(src_rows, out - row0);
    int lo = 0x7fffffff, hi = -1;
    if (!mask) {
        if (rows > 0 && slice == 0) { lo = 0; hi = in - 1; }
    } else if (!transposed) {                      // mask[out][in]: threads along k
        for (int r = slice; r < rows; r += KTAB_SLICES) {
            const float* __restrict__ m = mask + (int64_t)(row0 + r) * in;
#pragma unroll 4
            for (int k = t; k < in; k += 256)
                if (m[k] != 0.0f) { lo = min(lo, k); hi = max(hi, k); }
        }
    } else {                                       // mask[in][out]: threads along the tile's rows
        for (int k0 = slice; k0 < in; k0 += 8 * KTAB_SLICES) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * KTAB_SLICES;
                for (int r = t; r < rows; r += 256)
                    if (k < in && mask[(int64_t)k * out + row0 + r] != 0.0f) { lo = min(lo, k); hi = max(hi, k); }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, 64));
        hi = max(hi, __shfl_xor(hi, off, 64));
    }
    if ((t & 63) == 0 && hi >= 0) {
        atomicMin(&ktab[2 * tn], lo);
        atomicMax(&ktab[2 * tn + 1], hi);
    }
}

__global__ void ktab_steps_kernel(int32_t* __restrict__ ktab, int nt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const int lo = ktab[2 * i], hi = ktab[2 * i + 1];
    if (hi < 0) { ktab[2 * i] = 0; ktab[2 * i + 1] = 0; }
    else { ktab[2 * i] = lo / 16; ktab[2 * i + 1] = hi / 16 + 1; }
}

static void launch_ktab_fused(const float* mask, int out, int in, int src_rows, int nt, int32_t* ktab, int transposed,
                              hipStream_t s) {
    hipLaunchKernelGGL(ktab_reset_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, ktab, nt);
    hipLaunchKernelGGL(ktab_scan_kernel, dim3(nt, KTAB_SLICES), dim3(256), 0, s, mask, out, in, src_rows, ktab, transposed);
    hipLaunchKernelGGL(ktab_steps_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, ktab, nt);
}

// ======================================================================================================
// Split-bf16 ("bf16x3") flavour: fp32 operands carried as three bf16 terms a = a0 + a1 + a2 (8 + 8 + 8
// mantissa bits, an exact split), products formed from the six term pairs whose weight is >= 2^-16
// (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Dropped pairs
// are below 2^-24 relative: the result is as close to the exact product as the fp32 MFMA path's, at
// 6 x 32 cycles per 32x32x16 block instead of 8 x 64 (2.67x fewer matrix-pipe cycles).
//
// Shape: a 256-thread workgroup owns 128 samples x 256 parameter columns (11 whole features at P = 23);
// each of its 4 waves holds a 128 x 64 accumulator (4 x 2 MFMA tiles, 128 VGPRs), two workgroups per CU.
// K loop (split_k_loop): 16-wide steps, ONE barrier per step.  The A operand (the 128 samples, shared by the
// four waves) goes through double-buffered LDS: while a step's MFMAs run, the registers holding step k+1 go
// to the other stage and the loads of a later step are issued, one in front of each group of MFMAs.  The
// B operand (weights) is private to a wave, so it skips LDS: packed at pack time in MFMA operand order
// ([tile][step][wave][j][term][lane][8]), it is loaded from global memory straight into the operand
// registers, two register sets deep.  The hidden state is split into term planes by a pre-pass
// (whole-layer launches) or on its way into LDS (windowed launches of the inverse).
// Epilogue as in the fp32 kernel, 64 samples per pass.
//
// Split-fp16 ("f16x2") flavour of the same kernel: two fp16 terms a = a0 + a1 (11 + 11 mantissa bits),
// three products a0b0, a0b1, a1b0 on v_mfma_f32_32x32x16_f16; the dropped a1b1 is 2^-22 relative, below
// the fp32 accumulation error of a 512-term sum.  fp16 has 5 exponent bits, so the weights are scaled by
// a power of two chosen at pack time and the hidden state by one found per launch (largest finite magnitude
// just under 2^14 each; both undone exactly in the epilogue), so nothing finite overflows and an element
// keeps 22 bits unless it is 2^-14 times smaller than the largest one (then 2^-39 of that one, absolute).
// Half the matrix work and 2/3 of the operand traffic of bf16x3 - this kernel is power-limited, not
// issue-limited (the clock drops to ~1.7 GHz under it), so energy per product is what buys time.
// knock-out switches for timing experiments (benchmarks/build_variant.sh): results are wrong with any of them set
#ifndef NF_KO_STORE
#define NF_KO_STORE 0
#endif
#ifndef NF_KO_LOADS
#define NF_KO_LOADS 0
#endif
#ifndef NF_KO_MMA
#define NF_KO_MMA 0
#endif
#ifndef NF_KO_EVAL
#define NF_KO_EVAL 0
#endif
#ifndef NF_KO_OUT
#define NF_KO_OUT 0
#endif
#ifndef NF_KO_EPI
#define NF_KO_EPI 0
#endif
#ifndef NF_EPI_PRIO
#define NF_EPI_PRIO 0
#endif
#ifndef NF_F16X2_DEFAULT_FLAVOUR
#define NF_F16X2_DEFAULT_FLAVOUR 4          // (see w8_flavour)
#endif
constexpr int XM = 128, XT = 256, XS = 16;     // samples per workgroup, threads, K elements per step
// LDS stage = NT term planes of A [128 rows][32 B], rows unpadded; the two 16-byte halves of a row are
// swapped when bit 3 of the row is set, which makes the MFMA operand reads (ds_read_b128, lane = row, lane
// groups {0-3,12-15,20-27}...) conflict-free.  Two stages (double buffer).
constexpr int XLD = 16;                        // 16-bit elements per LDS row
constexpr int XA = XM * XLD;                    // elements per A term plane of an LDS stage
constexpr int XHR = 64, XLDT = XHR + 4;        // epilogue pass and its column-major tile stride

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct SplitBf16x3 {
    static constexpr int NT = 3;
    static constexpr bool SCALED = false;
    typedef __bf16 elem;
    typedef bf16x8 vec;
    static __device__ __forceinline__ void split1(float a, elem (&t)[NT]) {
        t[0] = (__bf16)a;
        const float r1 = a - (float)t[0];
        t[1] = (__bf16)r1;
        t[2] = (__bf16)(r1 - (float)t[1]);
    }
    static __device__ __forceinline__ void fma_tile(f32x16& acc, const vec (&a)[NT], const vec (&b)[NT]) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
};

struct SplitF16x2 {
    static constexpr int NT = 2;
    static constexpr bool SCALED = true;
    typedef _Float16 elem;
    typedef f16x8 vec;
    static __device__ __forceinline__ void split1(float a, elem (&t)[NT]) {
        t[0] = (_Float16)a;                       // operands arrive scaled below 2^14: no finite overflow
        t[1] = (_Float16)(a - (float)t[0]);
    }
    static __device__ __forceinline__ void fma_tile(f32x16& acc, const vec (&a)[NT], const vec (&b)[NT]) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
    }
};

// One scaled fp16 term per operand (11 significant bits relative to the tensor's largest element): the weight-gradient
// GEMM of the bf16 training mode - one MFMA per product, one conversion per element.
struct SplitF16x1 {
    static constexpr int NT = 1;
    static constexpr bool SCALED = true;
    typedef _Float16 elem;
    typedef f16x8 vec;
    static __device__ __forceinline__ void split1(float a, elem (&t)[NT]) { t[0] = (_Float16)a; }
    static __device__ __forceinline__ void fma_tile(f32x16& acc, const vec (&a)[NT], const vec (&b)[NT]) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
    }
};

// One bf16 term per operand, unscaled: the plain bf16-MFMA product of the 2e-2 mode (forward and input-gradient GEMMs)
// on the split kernels' tiling and K loop.
struct SplitBf16x1 {
    static constexpr int NT = 1;
    static constexpr bool SCALED = false;
    typedef __bf16 elem;
    typedef bf16x8 vec;
    static __device__ __forceinline__ void split1(float a, elem (&t)[NT]) { t[0] = (__bf16)a; }
    static __device__ __forceinline__ void fma_tile(f32x16& acc, const vec (&a)[NT], const vec (&b)[NT]) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
};

// s = 2^e with m * 2^e in [2^13, 2^14) for the float whose bit pattern is `bits` (0 / denormal: s = 1); inv = 2^-e
__device__ __forceinline__ void pow2_pair(unsigned bits, float& s, float& inv) {
    const int ex = (int)((bits >> 23) & 0xff) - 126;          // m = f * 2^ex, f in [0.5, 1)
    int e = ((bits >> 23) & 0xff) == 0 ? 0 : 14 - ex;
    e = max(-100, min(100, e));
    s = __uint_as_float((unsigned)(e + 127) << 23);
    inv = __uint_as_float((unsigned)(127 - e) << 23);
}

template <class SP, bool RELU = false>
__device__ __forceinline__ void split_chunk(const float4& lo, const float4& hi, bool live, float scale,
                                            typename SP::vec (&t)[SP::NT]) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        typename SP::elem e[SP::NT];
        float a = live ? v[j] : 0.f;
        if (RELU) a = fmaxf(a, 0.f);
        SP::split1(SP::SCALED ? a * scale : a, e);
#pragma unroll
        for (int p = 0; p < SP::NT; ++p) t[p][j] = e[p];
    }
}

// The K loop shared by the split-operand kernels: acc (+)= A[m0.., :] @ Wpacked[tile tn]^T over the tile's live
// 16-wide K steps, accumulators started at bias * bscale.  A comes from g.h (fp32, split on the fly with scale
// hs, ReLU first when RELU) or, PRE, from the term planes g.hp, and is shared by the four waves through LDS
// (double buffered, one barrier per step).  B is not shared - a wave multiplies only its own 64 columns - so
// it never touches LDS: the packed weights are laid out in MFMA operand order ([tile][step][wave][j][term]
// [lane][8]: 1 KB per wave-instruction, fully coalesced) and go from global memory straight into the operand
// registers, two register sets deep (a set is refilled for step k+2 as soon as step k has used it).
template <class SP, bool PRE, bool RELU>
__device__ __forceinline__ void split_k_loop(const FusedArgs& g, char* lds, int64_t m0, int tn, int tid, float bscale,
                                             float hs, f32x16 (&acc)[4][2]) {
    constexpr int NT = SP::NT;
    constexpr int XSTAGE = NT * XA;                    // elements per LDS stage (A term planes only)
    constexpr int BSTEP = 4 * 2 * NT * 64;             // 16-byte units of one packed weight step (all 4 waves)
    typedef typename SP::vec vec;
    typedef typename SP::elem elem;
    elem* stage0 = (elem*)lds;
    const int lane = tid & 63, wc = tid >> 6;
    const int kt_begin = g.ktab[2 * tn];
    const int kt_end = min(g.ktab[2 * tn + 1], (g.K + XS - 1) / XS);
    // The loop below counts LOGICAL steps 0 .. kn-1; KP() is the K step a logical step works on: the live range walked
    // upwards, or - g.k_reverse, the input-gradient GEMM - from its end downwards (see nf_masked_linear_bwd_input)
    const int kn = kt_end - kt_begin;
    const bool krev = g.k_reverse != 0;
#define KP(l_) (krev ? kt_end - 1 - (l_) : kt_begin + (l_))

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float bv = g.bias[tn * FN + wc * 64 + j * 32 + (lane & 31)] * bscale;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }

    // A staging: thread owns 8 consecutive k of one row (PRE: per term plane, zero padded to whole steps)
    const int arow = tid >> 1, ahalf = tid & 1;
    const float* ap = g.h + min(m0 + arow, g.M - 1) * g.ldh;
    const elem* app = (const elem*)g.hp + min(m0 + arow, g.M - 1) * g.hp_ld + ahalf * 8;
    const int64_t aplane = g.M * g.hp_ld;
    const vec* bp = (const vec*)g.w + ((int64_t)tn * (g.kpad / XS) * 4 + wc) * (2 * NT * 64) + lane;
    // LDS element offsets inside a stage: row * 16 + (half ^ bit 3 of the row) * 8
    const int woff = arow * XLD + ((ahalf ^ ((arow >> 3) & 1)) * 8);
    const int frow = lane & 31;
    const int roff = frow * XLD + (((lane >> 5) ^ ((frow >> 3) & 1)) * 8);
    // NSET register sets of staged A: with two (the two-term flavour has the registers for it) a load issued
    // during step k is consumed during step k+2 - a 768-cycle step alone does not cover an L2 round trip
    constexpr int NSET = NT <= 2 ? 2 : 1;
    constexpr int NLA = PRE ? NT : 2;                  // A loads per step: a vector per term plane, or the two fp32 halves
    float4 ra[NSET][2];
    vec rp[NSET][NT];
    vec b[2][2][NT];                                   // [register set = step parity][j][term]
#define X3_LOAD_A(s_, n_, kt_)                                                                        \
    {                                                                                                 \
        if (PRE) rp[s_][n_] = *reinterpret_cast<const vec*>(app + (n_) * aplane + KP(kt_) * XS);        \
        else if ((n_) < 2) ra[s_][(n_) & 1] = *reinterpret_cast<const float4*>(ap + min(KP(kt_) * XS + ahalf * 8, g.K - 8) + 4 * (n_)); \
    }
#define X3_LOAD_B(s_, j_, kt_)                                                                        \
    {                                                                                                 \
        _Pragma("unroll") for (int p_ = 0; p_ < NT; ++p_)                                             \
            b[s_][j_][p_] = bp[(int64_t)KP(kt_) * BSTEP + ((j_) * NT + p_) * 64];                       \
    }
#define X3_STORE_A(s_, st_, kt_)                                                                      \
    {                                                                                                 \
        elem* sa_ = (st_) + woff;                                                                     \
        vec t_[NT];                                                                                   \
        if (PRE) {                                                                                    \
            _Pragma("unroll") for (int p_ = 0; p_ < NT; ++p_) t_[p_] = rp[s_][p_];                    \
        } else {                                                                                      \
            split_chunk<SP, RELU>(ra[s_][0], ra[s_][1], KP(kt_) * XS + ahalf * 8 < g.K, hs, t_);        \
        }                                                                                             \
        _Pragma("unroll") for (int p_ = 0; p_ < NT; ++p_) *reinterpret_cast<vec*>(sa_ + p_ * XA) = t_[p_]; \
    }
    // one K step on LDS stage / B register set (par_): 8 groups of MFMAs; in front of group 1 the A register
    // set (s_) - step kt+1 - goes to the other LDS stage, in front of groups 2.. its refill loads for step
    // kt+1+NSET; B set (par_) is refilled for step kt+2 half by half as the j = 0 / j = 1 groups finish
#define X3_STEP(s_, par_, kt_)                                                                        \
    {                                                                                                 \
        const elem* sA = stage0 + (par_) * XSTAGE;                                                    \
        elem* nxt = stage0 + ((par_) ^ 1) * XSTAGE;                                                   \
        const int kt1 = min((kt_) + 1, kn - 1), kta = min((kt_) + 1 + NSET, kn - 1);          \
        const int ktb = min((kt_) + 2, kn - 1);                                                   \
        vec a[4][NT];                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
            _Pragma("unroll") for (int p = 0; p < NT; ++p)                                            \
                a[i][p] = *reinterpret_cast<const vec*>(sA + p * XA + i * 32 * XLD + roff);           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                const int grp = j * 4 + i;                                                            \
                if (grp == 1 && !NF_KO_STORE) X3_STORE_A(s_, nxt, kt1)                                \
                if (!NF_KO_LOADS) {                                                                   \
                    _Pragma("unroll") for (int n = 0; n < NLA; ++n)                                   \
                        if (grp == 2 + n) X3_LOAD_A(s_, n, kta)                                       \
                    if (grp == 5) X3_LOAD_B(par_, 0, ktb)                                             \
                }                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                    \
                if (!NF_KO_MMA) SP::fma_tile(acc[i][j], a[i], b[par_][j]);                            \
                __builtin_amdgcn_sched_barrier(0);                                                    \
            }                                                                                         \
        }                                                                                             \
        if (!NF_KO_LOADS) X3_LOAD_B(par_, 1, ktb)                                                     \
        __syncthreads();                                                                              \
    }
    // prologue: A step 0 into stage 0 and steps 1 .. NSET into the register sets; B steps 0 and 1
    if (kn > 0) {
#pragma unroll
        for (int n = 0; n < NLA; ++n) X3_LOAD_A(0, n, 0)
        X3_LOAD_B(0, 0, 0)
        X3_LOAD_B(0, 1, 0)
        X3_STORE_A(0, stage0, 0)
        const int kb1 = min(1, kn - 1);
        X3_LOAD_B(1, 0, kb1)
        X3_LOAD_B(1, 1, kb1)
#pragma unroll
        for (int q = 0; q < NSET; ++q) {
            const int kq = min(1 + q, kn - 1);
#pragma unroll
            for (int n = 0; n < NLA; ++n) X3_LOAD_A(q, n, kq)
        }
    }
    __syncthreads();
    // Steps in unconditional pairs, the odd last one after the loop: with the second step under an `if` inside the
    // loop, the loop header had a predecessor in which the loads of the first step were the youngest ones, and hipcc's
    // wait counts for the merged state drained nearly every outstanding load at the top of each iteration (vmcnt(1)
    // in front of the first MFMA: a full memory latency per two steps, which is what the one-term flavour's 256-cycle
    // steps were waiting on).
    int kt = 0;
    for (; kt + 1 < kn; kt += 2) {
        X3_STEP(0, 0, kt)
        X3_STEP(NSET - 1, 1, kt + 1)
    }
    if (kt < kn) X3_STEP(0, 0, kt)
#undef KP
#undef X3_STEP
#undef X3_STORE_A
#undef X3_LOAD_B
#undef X3_LOAD_A
}

template <class SP, int KT, bool INV, bool PRE>
__global__ __launch_bounds__(XT, 2) void made_rqs_split_kernel(const FusedArgs g) {
    constexpr int NT = SP::NT;
    constexpr int XSTAGE_BYTES = 2 * 2 * NT * XA;      // both LDS stages of the A term planes
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int fpt = g.fpt, P = g.c.P;
    const int ncol = fpt * P;
    float* tile = (float*)lds;                       // epilogue: [ncol][XLDT], aliases the staging planes
    float* xs = (float*)(lds + max(XSTAGE_BYTES, ncol * XLDT * 4));   // [128][fpt]
    float* ldp = xs + XM * fpt;                      // [4 waves][128]

    int64_t tm;
    int tn;
    tile_of_block(g, tm, tn);
    const int64_t m0 = tm * XM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wc = tid >> 6;
    const int f0 = tn * fpt;
    const int fv = min(fpt, g.feat_limit - f0);

#ifdef NF_STAMP
    long long sp[12];
    int spn = 0;
#define XSTAMP() { __builtin_amdgcn_sched_barrier(0); sp[spn++] = clock64(); __builtin_amdgcn_sched_barrier(0); }
#else
#define XSTAMP()
#endif
    XSTAMP()                                                     // 0: start
    {
        // all of the tile's inputs in flight at once (as a rolled loop every load waited in front of its LDS store:
        // five to six HBM latencies in a row at the head of each workgroup)
        constexpr int NX = (XM * (FN / 11) + XT - 1) / XT;         // fpt <= 256 / 11 (K >= 4)
        float xv[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int i = tid + q * XT;
            const int r = i / fpt, f = i - r * fpt;
            xv[q] = 0.f;
            if (i < XM * fpt && m0 + r < g.M && f < fv)
                xv[q] = g.x[(m0 + r) * g.ldx + (g.feat_index ? g.feat_index[f0 + f] : f0 + f)];
        }
#pragma unroll
        for (int q = 0; q < NX; ++q)
            if (tid + q * XT < XM * fpt) xs[tid + q * XT] = xv[q];
    }

    float ws = 1.f, wi = 1.f, hs = 1.f, hi = 1.f;      // power-of-two scales of the weights / of this launch's hidden state
    if (SP::SCALED) {
        pow2_pair(*g.wmax, ws, wi);
        pow2_pair(*g.hmax, hs, hi);
    }
    XSTAMP()                                                     // 1: inputs staged
    f32x16 acc[4][2];
#ifdef NF_KPRIO
    __builtin_amdgcn_s_setprio(NF_KPRIO);
#endif
    split_k_loop<SP, PRE, false>(g, lds, m0, tn, tid, ws * hs, hs, acc);
#ifdef NF_KPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    XSTAMP()                                                     // 2: K loop done

    // ---- epilogue: two passes of 64 samples (row blocks 2*half, 2*half + 1 of the accumulators) ------
    const float oscale = wi * hi;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        {
            const int col = wc * 64 + (lane & 31);
            float* tp = tile + col * XLDT + 4 * (lane >> 5);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (col + j * 32 < ncol) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float4 v;
                            if (half == 0) v = make_float4(acc[ii][j][4 * q], acc[ii][j][4 * q + 1], acc[ii][j][4 * q + 2], acc[ii][j][4 * q + 3]);
                            else v = make_float4(acc[2 + ii][j][4 * q], acc[2 + ii][j][4 * q + 1], acc[2 + ii][j][4 * q + 2], acc[2 + ii][j][4 * q + 3]);
                            if (SP::SCALED) { v.x *= oscale; v.y *= oscale; v.z *= oscale; v.w *= oscale; }
                            *reinterpret_cast<float4*>(tp + j * 32 * XLDT + ii * 32 + 8 * q) = v;
                        }
                    }
                }
        }
        __syncthreads();
        XSTAMP()                                                 // 3 / 5: accumulators dropped into the LDS tile
        float lsum = 0.f;                            // lane = sample row of the pass, wave-uniform feature
        const int r = half * XHR + lane;
        // a wave's (up to) three features of the pass in one straight-line block: the evaluations are
        // independent dependency chains that the scheduler interleaves (one wave per SIMD has no other ILP)
        constexpr int NU = 3;
        for (int fb = wc; fb < fv; fb += NU * (XT / 64)) {
            float yv[NU], ld[NU];
            int bin[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int f = min(fb + u * (XT / 64), fv - 1);
                if (NF_KO_EVAL) { yv[u] = tile[f * P * XLDT + lane]; ld[u] = yv[u]; bin[u] = 0; }
                else rqs_eval<KT, INV, XLDT, true, true, const float*, false>(tile + f * P * XLDT + lane, xs[r * fpt + f], g.c, yv[u], ld[u], bin[u]);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int f = fb + u * (XT / 64);
                if (f < fv) {
                    xs[r * fpt + f] = yv[u];
                    lsum += ld[u];
#ifndef NF_STAMP
                    if (g.bins && m0 + r < g.M) g.bins[(m0 + r) * (int64_t)g.D + f0 + f] = bin[u];
#endif
                }
            }
        }
        ldp[wc * XM + r] = lsum;
        XSTAMP()                                                 // 4 / 6: splines of the pass evaluated
    }
    __syncthreads();
    if (tid < XM) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < XT / 64; ++w) s += ldp[w * XM + tid];
        if (m0 + tid < g.M) g.partial[(int64_t)tn * g.M + m0 + tid] = s;
    }
    float ymax = 0.f;
    for (int i = tid; i < XM * fpt; i += XT) {
        const int r = i / fpt, f = i - r * fpt;
        if (m0 + r < g.M && f < fv) {
            const float v = xs[i];
            g.y[(m0 + r) * g.ldy + (g.feat_index ? g.feat_index[f0 + f] : f0 + f)] = v;
            const float a = fabsf(v);
            ymax = fmaxf(ymax, a < INFINITY ? a : 0.f);
        }
    }
    if (g.out_max) {                                 // max |y| for the next layer's operand scale
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
        // (read first: after the first few workgroups almost nobody holds a larger value, and tens of thousands
        //  of same-address atomics per launch would queue up in L2)
        if (lane == 0 && __float_as_uint(ymax) > __builtin_nontemporal_load(g.out_max)) atomicMax(g.out_max, __float_as_uint(ymax));
    }
#ifdef NF_STAMP
    XSTAMP()                                                     // 7: outputs stored
    if (g.bins && lane == 0) {
        long long* dbg = (long long*)g.bins + ((int64_t)blockIdx.x * (XT / 64) + wc) * 16;
        for (int i = 0; i < 8; ++i) dbg[i] = sp[i];
        dbg[8] = g.ktab[2 * tn + 1] - g.ktab[2 * tn];
        dbg[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        dbg[10] = tn;
    }
#endif
#undef XSTAMP
}

// Hidden-layer GEMM on split operands: y = [relu](x) @ (W * mask)^T + b (+ residual) [relu], fp32 in and out,
// 128 x 256 output tiles through the same K loop.  The largest |y| goes to g.out_max (the next layer's operand
// scale comes from it, so chained layers need no pass over their input).
// PRE: the A operand comes from term plane(s) g.hp ([terms][M][hp_ld], already in the operand type) instead of g.h.
template <class SP, bool RELU, bool PRE = false>
__global__ __launch_bounds__(XT, 2) void masked_linear_split_kernel(const FusedArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int64_t tm;
    int tn;
    tile_of_block(g, tm, tn);
    const int64_t m0 = tm * XM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wc = tid >> 6;
    float ws = 1.f, wi = 1.f, hs = 1.f, hi = 1.f;
    if (SP::SCALED) {
        pow2_pair(*g.wmax, ws, wi);
        pow2_pair(*g.hmax, hs, hi);
    }
    f32x16 acc[4][2];
    split_k_loop<SP, PRE, RELU>(g, lds, m0, tn, tid, ws * hs, hs, acc);
    const float oscale = wi * hi;
    float vmax = 0.f;
    if (g.vec_out) {
        // 16-byte epilogue: a wave drops 32 rows x its 64 columns into LDS (row stride 72 floats: conflict-free for
        // the accumulator layout), reads them back four columns per lane and adds the residual / stores with
        // float4 - 32 loads and 32 stores per wave instead of 128 + 128 four-byte ones, whose residual loads each
        // waited in front of their store
        __syncthreads();                                        // (the K loop's stages are dead)
        float* ot = reinterpret_cast<float*>(lds) + wc * (32 * XOT);
        const int64_t c0 = (int64_t)tn * FN + wc * 64 + (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ot[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * XOT + j * 32 + (lane & 31)] = acc[i][j][r];
#pragma unroll
            for (int qh = 0; qh < 8; qh += 4) {
                float4 rv[4], av[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = m0 + i * 32 + (qh + q) * 4 + (lane >> 4);
                    rv[q] = av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (g.residual && row < g.M && c0 < g.N)
                        rv[q] = *reinterpret_cast<const float4*>(g.residual + row * g.ldr + c0);
                    if (g.add2 && row < g.M && c0 < g.N)
                        av[q] = *reinterpret_cast<const float4*>(g.add2 + row * g.ldadd2 + c0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = m0 + i * 32 + (qh + q) * 4 + (lane >> 4);
                    float4 v = *reinterpret_cast<const float4*>(ot + ((qh + q) * 4 + (lane >> 4)) * XOT + (lane & 15) * 4);
                    if (SP::SCALED) { v.x *= oscale; v.y *= oscale; v.z *= oscale; v.w *= oscale; }
                    if (g.relu_out == 2) {          // `residual` is a gate: the ReLU derivative of an input-gradient GEMM
                        v.x = rv[q].x > 0.f ? v.x : 0.f; v.y = rv[q].y > 0.f ? v.y : 0.f;
                        v.z = rv[q].z > 0.f ? v.z : 0.f; v.w = rv[q].w > 0.f ? v.w : 0.f;
                        v.x += av[q].x; v.y += av[q].y; v.z += av[q].z; v.w += av[q].w;
                    } else {
                        v.x += rv[q].x; v.y += rv[q].y; v.z += rv[q].z; v.w += rv[q].w;
                        if (g.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    }
                    if (row < g.M && c0 < g.N && !(NF_KO_OUT && v.x != 12345.f)) {
                        *reinterpret_cast<float4*>(g.y + row * g.ldy + c0) = v;
                        const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
                        vmax = fmaxf(fmaxf(vmax, a0 < INFINITY ? a0 : 0.f), a1 < INFINITY ? a1 : 0.f);
                        vmax = fmaxf(fmaxf(vmax, a2 < INFINITY ? a2 : 0.f), a3 < INFINITY ? a3 : 0.f);
                    }
                }
            }
        }
    } else
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = tn * FN + wc * 64 + j * 32 + (lane & 31);
        if (col < g.N) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < g.M) {
                        float v = SP::SCALED ? acc[i][j][r] * oscale : acc[i][j][r];
                        if (g.relu_out == 2) {
                            v = g.residual[row * g.ldr + col] > 0.f ? v : 0.f;
                            if (g.add2) v += g.add2[row * g.ldadd2 + col];
                        } else {
                            if (g.residual) v += g.residual[row * g.ldr + col];
                            if (g.relu_out) v = fmaxf(v, 0.f);
                        }
                        g.y[row * g.ldy + col] = v;
                        const float a = fabsf(v);
                        vmax = fmaxf(vmax, a < INFINITY ? a : 0.f);
                    }
                }
        }
    }
    if (g.out_max) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0 && __float_as_uint(vmax) > __builtin_nontemporal_load(g.out_max)) atomicMax(g.out_max, __float_as_uint(vmax));
    }
}

// Weight gradient on split operands (two fp16 terms): grad_W[n, k] += mask[n, k] * sum_b grad_y[b, n] * act(x[b, k]),
// a TN GEMM whose reduction runs over the batch.  128 x 128 output tiles, the batch in chunks of 32 rows and split
// over gridDim.z (fp32 atomics), fully masked tiles skipped through the forward K-tile table.  The MFMA wants 8
// consecutive batch rows of one column per lane, so a thread reads its column for 8 rows (one 4-byte load per
// row, coalesced across the 128 columns), splits the 8 values and stores them as one 16-byte row piece of the
// transposed LDS tile [term][column][32 rows + pad].  Both operands are power-of-two scaled from their max |.|.
constexpr int WSB = 32;              // batch rows per step
constexpr int WSLD = WSB + 8;        // fp16 elements per LDS row (80 bytes: conflict-free 16-byte operand reads)

struct WgradSplitArgs {
    const float* x;
    int64_t ldx;
    const float* gy;
    int64_t ldgy;
    const unsigned* x_max;
    const unsigned* gy_max;
    const float* mask;
    const int32_t* ktab;     // nullable: forward K-tile table [out/128][2] in 32-wide units
    float* gw;
    float* gb;               // nullable: column sums of grad_y, accumulated by the tk == 0 blocks
    int64_t M;
    int out, in, relu_in, splits;
    const _Float16* xh;      // wgrad_tr_kernel<true>: act(x) * 2^e as fp16 plane(s) [terms][M][ldxh]
    int64_t ldxh, xh_plane;
    int nt, kt, group;       // tiles along out / in; consecutive work items kept on one XCD
};

// Work item w = (split * nt + tn) * kt + tk.  Workgroups go round-robin over the 8 XCDs by their linear id, so the
// id is folded such that `group` consecutive items (all K tiles of up to 8 output tiles of one batch split: they
// read the same grad_y / x chunks at the same time) land on ONE XCD and share its L2.
// TW = 1: 128 (out) x 128 (in) tiles, 4 waves, two workgroups per CU; TW = 2: 256 x 128 tiles, 8 waves, one workgroup per
// CU (wide layers: a quarter fewer loads and splits per product, grad_y chunks shared by half as many workgroups).
template <int TW, class SP = SplitF16x2>
__global__ __launch_bounds__(256 * TW, 2 / TW) void wgrad_split_kernel(const WgradSplitArgs g) {
    constexpr int NT = SP::NT;                       // fp16 terms per operand: 2 (f16x2, three MFMAs a product) or 1
    constexpr int YC = 128 * TW;                     // grad_y columns (output rows of grad_W) per tile
    constexpr int XO = 2 / TW;                       // x row octets per thread
    constexpr int YSZ = NT * YC * WSLD, XSZ = NT * 128 * WSLD;          // one buffer: all term planes, fp16 elements
    extern __shared__ __attribute__((aligned(16))) char wg_lds[];
    _Float16* const Yt = (_Float16*)wg_lds;                             // [2 buffers][2 terms][YC][WSLD]
    _Float16* const Xt = Yt + 2 * YSZ;                                  // [2 buffers][2 terms][128][WSLD]
    const int slot = blockIdx.x >> 3;
    const int64_t w = ((int64_t)(slot / g.group) * 8 + (blockIdx.x & 7)) * g.group + slot % g.group;
    if (w >= (int64_t)g.nt * g.kt * g.splits) return;
    // output tiles from the last one down: under MADE's masks those have the most unmasked K tiles (long items first)
    const int tk = (int)(w % g.kt), tn = g.nt - 1 - (int)((w / g.kt) % g.nt), split = (int)(w / ((int64_t)g.kt * g.nt));
    const int n0 = tn * YC, k0 = tk * 128;
    const bool do_bias = g.gb != nullptr && tk == 0;
    bool masked_out = false;
    if (g.ktab) {                                    // the table is per 128 output rows: this tile spans TW entries
        masked_out = true;
#pragma unroll
        for (int u = 0; u < TW; ++u) {
            if ((tn * TW + u) * 128 >= g.out) break;
            const int kb = g.ktab[2 * (tn * TW + u)] * 32, ke = g.ktab[2 * (tn * TW + u) + 1] * 32;
            if (k0 < ke && k0 + 128 > kb) masked_out = false;
        }
    }
    if (masked_out && !do_bias) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t per = ((g.M + g.splits - 1) / g.splits + WSB - 1) / WSB * WSB;
    const int64_t b_begin = (int64_t)split * per, b_end = min(g.M, b_begin + per);
    if (b_begin >= b_end) return;
    if (masked_out) {        // bias only: this block's column sums of grad_y
        const int c = tid % YC;
        if (n0 + c < g.out) {
            float s = 0.f;
            for (int64_t b = b_begin + tid / YC; b < b_end; b += 2) s += g.gy[b * g.ldgy + n0 + c];
            atomicAdd(&g.gb[n0 + c], s);
        }
        return;
    }
    float sy, iy, sx, ix;
    pow2_pair(*g.gy_max, sy, iy);
    pow2_pair(*g.x_max, sx, ix);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging: grad_y thread = column (tid % YC) x row octets (tid / YC) and (tid / YC) + 2 of the 32-row chunk; x
    // thread = column (tid & 127) x octets (tid >> 7) + 2 TW o.  Columns
    // past the matrix edge are clamped, not zeroed: they only reach accumulator entries that are never stored.
    // A load is a buffer load: descriptor on the chunk (scalar), this lane's constant byte offset (its column),
    // the row's offset in a scalar register - no vector address arithmetic.  Rows past the batch end are clamped
    // to the last row and grad_y's scale for them is 0 (all scalar work), so the tail costs the lanes nothing.
    const int sc = tid % YC, so = tid / YC, xsc = tid & 127, xso = tid >> 7;
    const int so_u = __builtin_amdgcn_readfirstlane(so);                 // uniform per wave
    const int xso_u = __builtin_amdgcn_readfirstlane(xso);
    const int yc = min(n0 + sc, g.out - 1), xc = min(k0 + xsc, g.in - 1);
    const bool yok = n0 + sc < g.out;
    const int ldy4 = (int)g.ldgy * 4, ldx4 = (int)g.ldx * 4;
    struct Chunk { float y[2][8], x[XO][8]; };
    auto load_chunk = [&](Chunk& c, int64_t b0) {
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.gy + b0 * g.ldgy), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x + b0 * g.ldx), 0, 0x7fffffff, 0x00020000);
        const int last = (int)min((int64_t)WSB, b_end - b0) - 1;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = min((so_u + 2 * o) * 8 + e, last);
                c.y[o][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, yc * 4, r * ldy4, 0));
            }
#pragma unroll
        for (int o = 0; o < XO; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = min((xso_u + 2 * TW * o) * 8 + e, last);
                c.x[o][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xc * 4, r * ldx4, 0));
            }
    };
    const int fr = lane & 31, fb = lane >> 5;
    float bsum = 0.f;                                // in grad_y's scaled units
    // chunk c (rows b0 ..) -> split terms -> LDS buffer `buf`
    auto convert = [&](const Chunk& c, int64_t b0, int buf) {
        const int left = (int)min((int64_t)WSB, b_end - b0);
        _Float16* const yt = Yt + buf * YSZ;
        _Float16* const xt = Xt + buf * XSZ;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            typename SP::vec ty[NT];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                typename SP::elem t[NT];
                const float ys = c.y[o][e] * ((so_u + 2 * o) * 8 + e < left ? sy : 0.f);
                bsum += ys;
                SP::split1(ys, t);
#pragma unroll
                for (int p = 0; p < NT; ++p) ty[p][e] = t[p];
            }
#pragma unroll
            for (int p = 0; p < NT; ++p)
                *reinterpret_cast<typename SP::vec*>(&yt[(p * YC + sc) * WSLD + (so + 2 * o) * 8]) = ty[p];
        }
#pragma unroll
        for (int o = 0; o < XO; ++o) {
            typename SP::vec tx[NT];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                typename SP::elem t[NT];
                const float xa = g.relu_in ? __builtin_amdgcn_fmed3f(c.x[o][e], 0.f, INFINITY) : c.x[o][e];
                SP::split1(xa * sx, t);
#pragma unroll
                for (int p = 0; p < NT; ++p) tx[p][e] = t[p];
            }
#pragma unroll
            for (int p = 0; p < NT; ++p)
                *reinterpret_cast<typename SP::vec*>(&xt[(p * 128 + xsc) * WSLD + (xso + 2 * TW * o) * 8]) = tx[p];
        }
    };
    auto mma = [&](int buf) {
        const _Float16* const yt = Yt + buf * YSZ;
        const _Float16* const xt = Xt + buf * XSZ;
#pragma unroll
        for (int ks = 0; ks < WSB / 16; ++ks) {
            typename SP::vec a[2][NT], b[2][NT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < NT; ++p) {
                    a[i][p] = *reinterpret_cast<const typename SP::vec*>(&yt[(p * YC + wr * 64 + i * 32 + fr) * WSLD + ks * 16 + fb * 8]);
                    b[i][p] = *reinterpret_cast<const typename SP::vec*>(&xt[(p * 128 + wc * 64 + i * 32 + fr) * WSLD + ks * 16 + fb * 8]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) SP::fma_tile(acc[i][j], a[i], b[j]);
        }
    };
    // One barrier per step: while the MFMAs run on buffer s & 1, the same waves split chunk s + 1 into the other
    // buffer (its last readers passed the previous barrier) and refill that chunk's registers with chunk s + 3.
    Chunk c0, c1;
    load_chunk(c0, b_begin);
    if (b_begin + WSB < b_end) load_chunk(c1, b_begin + WSB);
    convert(c0, b_begin, 0);
    if (b_begin + 2 * WSB < b_end) load_chunk(c0, b_begin + 2 * WSB);
    __syncthreads();
    // The two waves that share a SIMD (w and w + 4 with 8 waves) take the two halves of a step in opposite order,
    // so one splits on the vector ALU while the other's MFMAs run.
    const bool mma_first = TW == 2 && __builtin_amdgcn_readfirstlane(wave >> 2) != 0;
    for (int64_t b0 = b_begin; b0 < b_end; b0 += 2 * WSB) {
        if (mma_first) { mma(0); __builtin_amdgcn_sched_barrier(0); }
        if (b0 + WSB < b_end) {
            convert(c1, b0 + WSB, 1);
            if (b0 + 3 * WSB < b_end) load_chunk(c1, b0 + 3 * WSB);
        }
        if (!mma_first) { __builtin_amdgcn_sched_barrier(0); mma(0); }
        __syncthreads();
        if (b0 + WSB >= b_end) break;
        if (mma_first) { mma(1); __builtin_amdgcn_sched_barrier(0); }
        if (b0 + 2 * WSB < b_end) {
            convert(c0, b0 + 2 * WSB, 0);
            if (b0 + 4 * WSB < b_end) load_chunk(c0, b0 + 4 * WSB);
        }
        if (!mma_first) { __builtin_amdgcn_sched_barrier(0); mma(1); }
        __syncthreads();
    }
    bsum *= iy;
    if (do_bias && yok) atomicAdd(&g.gb[n0 + sc], bsum);
    const float oscale = iy * ix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wc * 64 + j * 32 + (lane & 31);
            if (k >= g.in) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n >= g.out) continue;
                const int64_t idx = (int64_t)n * g.in + k;
                float v = acc[i][j][r] * oscale;
                if (g.mask) v *= g.mask[idx];
                if (g.splits > 1) atomicAdd(&g.gw[idx], v);
                else g.gw[idx] += v;
            }
        }
}

// ---- weight gradient, one fp16 term per operand, transposed LDS reads (bf16 training mode) ---------------------
// Same product as wgrad_split_kernel<2, SplitF16x1> (8 waves, 32 batch rows per step, batch splits with fp32
// atomics, masked tiles skipped), different tile and operand path: the MFMA wants, per lane, 8 consecutive BATCH
// rows of one column, while grad_y / x are stored batch-major.  wgrad_split_kernel fetches that shape directly - 24
// four-byte loads per lane and step, which is what bounded it (2.1 ms on the final layer).  Here a chunk is fetched
// the way it lies in memory - 16-byte loads along a row, a wave takes 1 KB contiguous -, scaled, converted and stored
// ROW-major in LDS (8 bytes per load), and the operands come out column-major through gfx950's transposing read
// `ds_read_b64_tr_b16` (per 16 lanes: a 4-row x 16-column block, delivered one column per lane).  LDS images are
// [32 batch rows][128 columns] fp16 with 256-byte rows and the 16-byte chunks of a row XOR-swizzled by
// ((row & 3) << 2) | ((row >> 2) & 3): conflict-free for these reads (cdna_hip_programming.md T10, layout (b)).
// Tile: 128 output rows x WT_K = 256 input columns of grad_W per workgroup (8 waves of 64 x 64).  With 512 columns a
// grad_y chunk would be fetched once for the whole input width of this model's wide layers, but 128 accumulator
// registers leave no room for chunks in flight (measured: spills, 1.7-11 ms); with 256 the two K tiles of an output
// tile are neighbours on one XCD and the second finds grad_y in L2.  The x chunk is the same for every workgroup of a
// batch split and stays in L2.  A buffer = grad_y columns 0..127 + x columns 0..255 as two images (3 x 8 KB), double
// buffered: 48 KB.
constexpr int WT_IMG = 32 * 256;                   // bytes of one image
constexpr int WT_N = 128;                         // (input columns per tile: the kernel's template parameter WT_K)
__device__ __forceinline__ int wt_off(int row, int ch) {     // byte offset of 16-byte chunk ch (0..15) of a row
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// x as a pre-converted plane (XP): act(x) * 2^e in fp16, [M][in] row-major (half_plane_kernel), fetched global -> LDS by
// LDS-DMA with the swizzle applied on the global side (lane l of a 1 KB piece = LDS chunk position l & 15 of row l >> 4
// fetches logical chunk position ^ swizzle(row)): half the bytes of the fp32 rows through L2 -> CU, which is what
// bounds this kernel (every workgroup of a batch split re-reads the same x chunk), no staging registers and no
// conversion work for x in the loop.  Three x buffers: the DMA of chunk s + 2 is issued in step s.
// (NT = 2: the second plane, M * K elements further, holds the fp16 remainders of the first - the f16x2 operand split)
template <int NT>
__global__ __launch_bounds__(256) void half_plane_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int K,
                                                         const unsigned* __restrict__ x_max, int relu,
                                                         _Float16* __restrict__ plane) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    float sx, ix;
    pow2_pair(*x_max, sx, ix);
    const int cpr = K / 4;
    const int64_t total = M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        v.x *= sx; v.y *= sx; v.z *= sx; v.w *= sx;
        const h4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<h4*>(plane + r * K + c) = hv;
        if (NT == 2) {
            const h4 lv = {(_Float16)(v.x - (float)hv[0]), (_Float16)(v.y - (float)hv[1]), (_Float16)(v.z - (float)hv[2]),
                           (_Float16)(v.w - (float)hv[3])};
            *reinterpret_cast<h4*>(plane + M * K + r * K + c) = lv;
        }
    }
}

// one LDS-DMA wave-instruction: lane l's 16 bytes at gsrc land at lds_dst + 16 l (lds_dst wave-uniform); M0 is
// compiler-reserved: saved and restored around the instruction
__device__ __forceinline__ void wt_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// NT = 2 (with the x planes only): the f16x2 flavour - two fp16 terms per operand, products hi*hi + hi*lo + lo*hi
// (SplitF16x2's), twice the images, three MFMAs per product.
// YB: grad_y is a bf16 matrix [M][ldgy] (nf_rqs_bwd_max_bf16's output): 8-byte fetches, widened in the conversion step.
template <bool XP, int WT_K, int NT = 1, bool YB = false>
__global__ __launch_bounds__(512, 1) void wgrad_tr_kernel(const WgradSplitArgs g) {
    static_assert(NT == 1 || XP, "two terms: x comes from its planes");
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef short s4 __attribute__((ext_vector_type(4)));
    typedef short s8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s4* lds_s4;
    constexpr int NXI = WT_K / 128;                  // x images of a chunk
    constexpr int XL = WT_K / 64;                    // 16-byte x loads per thread and chunk (fp32 rows)
    constexpr int XG = WT_K / 4;                     // four-column groups of an x row
    constexpr int NJ = WT_K / 128;                   // 32-column blocks of a wave along the input width
    // LDS: XP ? [2 buffers][NT] grad_y images, then [3 buffers][NT][NXI] x images
    //         : [2 buffers of (grad_y image + NXI x images)]
    extern __shared__ __attribute__((aligned(16))) char wt_lds[];
    auto ybase = [&](int buf) { return wt_lds + (XP ? buf * NT * WT_IMG : buf * (1 + NXI) * WT_IMG); };
    auto xbase = [&](int buf) { return wt_lds + (XP ? (2 * NT + buf * NT * NXI) * WT_IMG : (buf * (1 + NXI) + 1) * WT_IMG); };
    const int slot = blockIdx.x >> 3;
    const int64_t w = ((int64_t)(slot / g.group) * 8 + (blockIdx.x & 7)) * g.group + slot % g.group;
    if (w >= (int64_t)g.nt * g.kt * g.splits) return;
    const int tk = (int)(w % g.kt), tn = g.nt - 1 - (int)((w / g.kt) % g.nt), split = (int)(w / ((int64_t)g.kt * g.nt));
    const int n0 = tn * WT_N, k0 = tk * WT_K;
    const bool do_bias = g.gb != nullptr && tk == 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;         // wave tile: 64 (out) x WT_K / 4 (in)
    // live K range of this tile's 128 output rows (forward table, 32-wide units): a wave whose input columns lie
    // outside it has only masked weights to produce
    int kb = 0, ke = g.in;
    if (g.ktab) { kb = g.ktab[2 * tn] * 32; ke = g.ktab[2 * tn + 1] * 32; }
    const int cw = wc * (WT_K / 4), kw0 = k0 + cw;
    const bool live = kw0 < min(ke, g.in) && kw0 + WT_K / 4 > kb;
    const bool any_live = k0 < min(ke, g.in) && k0 + WT_K > kb;
    if (!any_live && !do_bias) return;
    const int64_t per = ((g.M + g.splits - 1) / g.splits + 6 * WSB - 1) / (6 * WSB) * (6 * WSB);
    const int64_t b_begin = (int64_t)split * per, b_end = min(g.M, b_begin + per);
    if (b_begin >= b_end) return;
    float sy, iy, sx, ix;
    pow2_pair(*g.gy_max, sy, iy);
    pow2_pair(*g.x_max, sx, ix);

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fetch map: grad_y chunk = 32 rows x 32 four-column groups, thread = group (tid & 31) x rows (tid >> 5) + 16 e;
    // x chunk (fp32 rows) = 32 rows x XG groups, thread = group (tid % XG) x rows (tid / XG) + (512 / XG) e.  Columns past
    // the matrix edge are clamped (they only reach accumulator entries that are never stored), rows past the batch end
    // are clamped and grad_y's are scaled by 0.  Every load is unconditional: a load under a branch makes hipcc wait
    // vmcnt(0) at each use.
    const int ycg = tid & 31, yr0 = tid >> 5, xcg = tid % XG, xr0 = tid / XG;
    const float* ycol = g.gy + min(n0 + 4 * ycg, g.out - 4);
    const __bf16* ycol16 = reinterpret_cast<const __bf16*>(g.gy) + min(n0 + 4 * ycg, g.out - 4);
    const float* xcol = g.x + min(k0 + 4 * xcg, g.in - 4);
    // grad_y comes from HBM: three chunks of it in flight; x is the same for every workgroup of a batch split (L2): two
    constexpr int NY = WT_K > 256 ? 2 : 3;           // (the 512-column tile has 128 accumulator registers: one set fewer)
    float4 cy[YB ? 1 : NY][2], cx[XP ? 1 : 2][XL];
    uint2 cyb[YB ? NY : 1][2];
    auto load_y = [&](int c, int64_t b0) {
#ifndef NF_WT_KO_LOADS
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (YB) cyb[c][e] = *reinterpret_cast<const uint2*>(ycol16 + min(b0 + yr0 + 16 * e, b_end - 1) * g.ldgy);
            else cy[c][e] = *reinterpret_cast<const float4*>(ycol + min(b0 + yr0 + 16 * e, b_end - 1) * g.ldgy);
        }
#endif
    };
    auto load_x = [&](int c, int64_t b0) {
#ifndef NF_WT_KO_LOADS
        if (XP) return;
#pragma unroll
        for (int e = 0; e < XL; ++e)
            cx[c][e] = *reinterpret_cast<const float4*>(xcol + min(b0 + xr0 + (512 / XG) * e, b_end - 1) * g.ldx);
#endif
    };
    // XP: the chunk's NXI * 8 one-KB pieces (4 rows of an image each) are spread over the 8 waves, NXI per wave
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wt_lds;
    // (addresses recomputed per piece: a handful of integer operations against 4 registers per piece held for the loop)
    auto dma_x = [&](int buf, int64_t b0) {
#ifndef NF_WT_KO_LOADS
#pragma unroll
        for (int t = 0; t < NT * NXI; ++t) {
            const int piece = NT * NXI * wave + t, term = piece / (8 * NXI), img = (piece >> 3) % NXI, q = piece & 7;
            const int row = 4 * q + (lane >> 4);
            const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            wt_glds16(g.xh + term * g.xh_plane + min(k0 + img * 128 + 8 * ch, g.in - 8) + min(b0 + row, b_end - 1) * g.ldxh,
                      lds0 + (unsigned)((2 * NT + (buf * NT + term) * NXI + img) * WT_IMG + q * 1024));
        }
#endif
    };
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};            // in grad_y's scaled units: this thread's four columns
    auto convert = [&](int yc, int xc, int64_t b0, int buf) {
#ifndef NF_WT_KO_CONV
        char* yb = ybase(buf);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = yr0 + 16 * e;
            const float sc = b0 + r < b_end ? sy : 0.f;
            float4 yv;
            if (YB) {                                // bf16 = the high half of the fp32 pattern
                const uint2 u = cyb[yc][e];
                yv = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                 __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
            } else {
                yv = cy[yc][e];
            }
            const float v0 = yv.x * sc, v1 = yv.y * sc, v2 = yv.z * sc, v3 = yv.w * sc;
            bsum[0] += v0; bsum[1] += v1; bsum[2] += v2; bsum[3] += v3;
            h4 hv = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
            *reinterpret_cast<h4*>(yb + wt_off(r, ycg >> 1) + 8 * (ycg & 1)) = hv;
            if (NT == 2) {
                h4 lv = {(_Float16)(v0 - (float)hv[0]), (_Float16)(v1 - (float)hv[1]), (_Float16)(v2 - (float)hv[2]),
                         (_Float16)(v3 - (float)hv[3])};
                *reinterpret_cast<h4*>(yb + WT_IMG + wt_off(r, ycg >> 1) + 8 * (ycg & 1)) = lv;
            }
        }
        if (XP) return;
        char* xb = xbase(buf);
#pragma unroll
        for (int e = 0; e < XL; ++e) {
            const int r = xr0 + (512 / XG) * e;
            float v0 = cx[xc][e].x, v1 = cx[xc][e].y, v2 = cx[xc][e].z, v3 = cx[xc][e].w;
            if (g.relu_in) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            h4 hv = {(_Float16)(v0 * sx), (_Float16)(v1 * sx), (_Float16)(v2 * sx), (_Float16)(v3 * sx)};
            *reinterpret_cast<h4*>(xb + (xcg >> 5) * WT_IMG + wt_off(r, (xcg & 31) >> 1) + 8 * (xcg & 1)) = hv;
        }
#endif
    };
    // transposed operand reads: group gq = lane >> 4 of 16 lanes, lane 4q + p of it addresses row q, columns 4p..4p+3 of
    // its block and receives column (lane & 15); block = rows 16 ks + 8 (gq >> 1) + 4 t .. + 3, columns base + 16 (gq & 1)
    const int gq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    int aoff[2][2], boff[NJ][2];                     // [i or j][t]: byte offsets inside the grad_y image / x images, ks = 0
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 8 * (gq >> 1) + 4 * t + tq;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            aoff[i][t] = wt_off(row, (64 * wr + 32 * i + 16 * (gq & 1)) / 8 + (tp >> 1)) + 8 * (tp & 1);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = cw + 32 * j + 16 * (gq & 1);             // of the x chunk's WT_K columns
            boff[j][t] = (col / 128) * WT_IMG + wt_off(row, (col % 128) / 8 + (tp >> 1)) + 8 * (tp & 1);
        }
    }
    auto rd = [&](const char* base, int off) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(base + off));
    };
    auto mma = [&](int ybuf, int xbuf) {
#ifndef NF_WT_KO_MMA
        if (!live) return;                           // wave-uniform
        const char* yb = ybase(ybuf);
        const char* xb = xbase(xbuf);
#pragma unroll
        for (int ks = 0; ks < WSB / 16; ++ks) {
            f16x8 a[2][NT], b[NJ][NT];
            // (rows + 16 ks: 4096 bytes further, the swizzle does not involve that bit)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const s4 lo = rd(yb, t * WT_IMG + aoff[i][0] + 4096 * ks), hi = rd(yb, t * WT_IMG + aoff[i][1] + 4096 * ks);
                    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    a[i][t] = __builtin_bit_cast(f16x8, v);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const s4 lo = rd(xb, t * NXI * WT_IMG + boff[j][0] + 4096 * ks);
                    const s4 hi = rd(xb, t * NXI * WT_IMG + boff[j][1] + 4096 * ks);
                    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    b[j][t] = __builtin_bit_cast(f16x8, v);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (NT == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][NT - 1], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][NT - 1], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
#endif
    };
    // the step barrier orders LDS traffic only: __syncthreads() would also drain vmcnt, i.e. wait out the chunks
    // requested steps ahead on purpose.  XP: the DMA pieces of chunk s + 1, which the next step reads, must have landed
    // - after them this wave issued 2 grad_y loads in their own step (s - 1) and NT * NXI + 2 operations in this one (loads return
    // in order; hipcc's own counts for the grad_y registers only become stricter by the DMA operations it cannot see)
    auto lds_barrier = [] {
        if (XP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + NT * NXI + 2) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // Waves w and w + 4 share a SIMD and take the two halves of a step - the MFMAs of chunk s (reads buffer s & 1) and
    // the conversion of chunk s + 1 (writes the other buffer) - in opposite order: one's vector-ALU / LDS work runs
    // under the other's MFMAs instead of all eight waves queueing on the same pipe in lockstep.
    const bool mma_first = (wave >> 2) != 0;
    // step s (chunk s = rows b_begin + 32 s ..): convert chunk s + 1 (grad_y set (s + 1) % 3, x set (s + 1) & 1) into
    // LDS buffer (s + 1) & 1 - its last readers passed the previous barrier -, request x chunk s + 3 (XP: DMA chunk
    // s + 2 into x buffer (s + 2) % 3, last read in step s - 1) and then grad_y chunk s + 4 into the sets just freed
    // (x first: loads return in order, and the next step's wait for its x must not include a fresh HBM fetch), and run
    // the MFMAs of chunk s.  Unrolled by 6: every set / buffer index constant.
    // No exit inside the six steps (an early exit gives the loop header a second predecessor whose loads were issued
    // one step ago, and hipcc then waits for ALL loads in step 0): steps past the batch end multiply zeros.
    load_y(0, b_begin);
    load_x(0, b_begin);
    if (XP) { dma_x(0, b_begin); dma_x(1, b_begin + WSB); }
    load_y(1, b_begin + WSB);
    load_x(1, b_begin + WSB);
    if (NY == 3) load_y(2, b_begin + 2 * WSB);
    convert(0, 0, b_begin, 0);
    load_x(0, b_begin + 2 * WSB);
    load_y(0, b_begin + NY * WSB);
    if (XP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#define WT_STEP(S_)                                                                       \
    {                                                                                     \
        const int64_t bs_ = b0 + (S_) * WSB;                                              \
        if (mma_first) { mma((S_) & 1, XP ? (S_) % 3 : (S_) & 1); __builtin_amdgcn_sched_barrier(0); } \
        convert(((S_) + 1) % NY, ((S_) + 1) & 1, bs_ + WSB, ((S_) + 1) & 1);              \
        if (XP) dma_x(((S_) + 2) % 3, bs_ + 2 * WSB);                                     \
        else load_x(((S_) + 1) & 1, bs_ + 3 * WSB);                                       \
        load_y(((S_) + 1) % NY, bs_ + (1 + NY) * WSB);                                    \
        if (!mma_first) { __builtin_amdgcn_sched_barrier(0); mma((S_) & 1, XP ? (S_) % 3 : (S_) & 1); } \
        lds_barrier();                                                                    \
    }
    for (int64_t b0 = b_begin; b0 < b_end; b0 += 6 * WSB) {
        WT_STEP(0) WT_STEP(1) WT_STEP(2) WT_STEP(3) WT_STEP(4) WT_STEP(5)
    }
#undef WT_STEP
    if (XP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // DMA pieces still in flight target LDS
    if (do_bias) {           // the 16 threads that share a column group add up through LDS (fixed order), one atomic each
        __syncthreads();
        float* red = reinterpret_cast<float*>(wt_lds);           // [16][128]
#pragma unroll
        for (int q = 0; q < 4; ++q) red[yr0 * 128 + 4 * ycg + q] = bsum[q];
        __syncthreads();
        if (tid < 128 && n0 + tid < g.out) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += red[q * 128 + tid];
            atomicAdd(&g.gb[n0 + tid], s * iy);
        }
    }
#ifdef NF_WT_KO_EPI
    if (acc[0][0][0] != 12345.f) return;
#endif
    if (!live) return;
    const float oscale = iy * ix;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = kw0 + j * 32 + (lane & 31);
            if (k >= g.in) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n >= g.out) continue;
                const int64_t idx = (int64_t)n * g.in + k;
                float v = acc[i][j][r] * oscale;
                if (g.mask) v *= g.mask[idx];
                if (g.splits > 1) atomicAdd(&g.gw[idx], v);
                else g.gw[idx] += v;
            }
        }
}

// packed[tile][step][wave 0..3][j 0..1][term][lane 0..63][8]: the 8 elements lane `lane` feeds to the MFMA as
// the B operand of column block j of wave `wave` - row = wave*64 + j*32 + (lane & 31) of the tile,
// k = step*16 + (lane >> 5)*8 + e - i.e. term t of W[n][k] * mask * fold (* the scale slot's power of two)
template <class SP>
__global__ void pack_fused_split_kernel(const float* __restrict__ W, const float* __restrict__ mask,
                                        const float* __restrict__ bias, int out, int in, int src_rows, int kpad,
                                        int64_t total, int P, int n_logits, float logit_fold,
                                        const unsigned* __restrict__ wmax, typename SP::elem* __restrict__ packed,
                                        float* __restrict__ bias_packed, int transposed) {
    // transposed: W / mask are stored [in][out] (the layer's own arrays, packed here as the input-gradient GEMM's operand)
    constexpr int NT = SP::NT;
    const int nsteps = kpad / XS;
    float sw = 1.f, swi = 1.f;
    if (SP::SCALED) pow2_pair(*wmax, sw, swi);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i % 8);
        const int lane = (int)((i / 8) % 64);
        const int term = (int)((i / 512) % NT);
        const int jj = (int)((i / (512 * NT)) % 2);
        const int w = (int)((i / (1024 * NT)) % 4);
        const int step = (int)((i / (4096 * NT)) % nsteps);
        const int tn = (int)(i / ((int64_t)4096 * NT * nsteps));
        const int j = w * 64 + jj * 32 + (lane & 31);          // row of the 256-row tile
        const int k = step * XS + (lane >> 5) * 8 + e;
        const int64_t n = (int64_t)tn * src_rows + j;
        const bool live = j < src_rows && n < out;
        const float fold = (j % P) < n_logits ? logit_fold : 1.0f;
        float v = 0.0f;
        if (live && k < in) {
            const int64_t at = transposed ? (int64_t)k * out + n : n * in + k;
            v = W[at];
            if (mask) v *= mask[at];
            v *= fold;
        }
        typename SP::elem t[NT];
        SP::split1(v * sw, t);
        typename SP::elem o = t[0];
#pragma unroll
        for (int q = 1; q < NT; ++q) o = term == q ? t[q] : o;
        packed[i] = o;
        if (k == 0 && term == 0) bias_packed[tn * FN + j] = (live && bias) ? bias[n] * fold : 0.0f;
    }
}

// scale slot (after the K-range table): the bit pattern of max |W * mask * fold| (positive floats order like
// unsigned integers, so atomicMax works on the bits); consumers derive the power of two with pow2_pair.
__global__ __launch_bounds__(256) void maxabs_kernel(const float* __restrict__ W, const float* __restrict__ mask,
                                                     int64_t total, int P, int in, int n_logits, float logit_fold,
                                                     unsigned* __restrict__ slot) {
    const int out = (int)(total / in);
    float m = 0.f;
    for (int r = blockIdx.x; r < out; r += gridDim.x) {           // a workgroup per weight row: the fold is per row
        const float fold = (r % P) < n_logits ? logit_fold : 1.0f;
        const float* __restrict__ w = W + (int64_t)r * in;
        const float* __restrict__ k = mask ? mask + (int64_t)r * in : nullptr;
        float mr = 0.f;
        for (int c = threadIdx.x; c < in; c += 256) mr = fmaxf(mr, fabsf(k ? w[c] * k[c] : w[c]));
        m = fmaxf(m, mr * fabsf(fold));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float wm[4];                          // one atomic per workgroup: they all hit the same address
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f && m < INFINITY) atomicMax(slot, __float_as_uint(m));
    }
}

// largest finite |h| of h[M][K] into slot[0] (bit pattern, see above)
__global__ __launch_bounds__(256) void maxabs_rows_kernel(const float* __restrict__ h, int64_t ldh, int64_t M, int K,
                                                          unsigned* __restrict__ slot) {
    const int cpr = K / 4;
    float m = 0.f;
    auto take = [&](const float4 v) {
        const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));   // fmaxf drops NaN
        m = fmaxf(m, a < INFINITY ? a : 0.f);
    };
    if (K % 4 != 0 || ldh % 4 != 0 || ((uintptr_t)h & 15) != 0) {     // ragged / unaligned rows: scalar loads
        for (int64_t r = blockIdx.x; r < M; r += gridDim.x)
            for (int c = threadIdx.x; c < K; c += 256) {
                const float a = fabsf(h[r * ldh + c]);
                m = fmaxf(m, a < INFINITY ? a : 0.f);
            }
    } else if (ldh == K) {                                  // contiguous: one flat array, 4 independent loads in flight
        const int64_t total = M * cpr, stride = (int64_t)gridDim.x * 256;
        const float4* p = reinterpret_cast<const float4*>(h);
        int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < total; i += 4 * stride) {
            const float4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], e = p[i + 3 * stride];
            take(a); take(b); take(c); take(e);
        }
        for (; i < total; i += stride) take(p[i]);
    } else {                                         // strided rows: a workgroup walks 4 rows at a time
        for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < M; r0 += (int64_t)gridDim.x * 4)
            for (int c = threadIdx.x; c < cpr; c += 256) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = *reinterpret_cast<const float4*>(h + min(r0 + u, M - 1) * ldh + c * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) take(v[u]);
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float wm[4];                          // one atomic per workgroup: they all hit the same address
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(slot, __float_as_uint(m));
    }
}

// h[M][K] fp32 -> NT term planes [NT][M][kp16], kp16 = K rounded up to whole 16-wide steps (zero filled)
template <class SP>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ h, int64_t ldh, int64_t M, int K,
                                                           int kp16, const unsigned* __restrict__ hmax,
                                                           typename SP::elem* __restrict__ planes) {
    float hs = 1.f, hi = 1.f;
    if (SP::SCALED) pow2_pair(*hmax, hs, hi);
    const int cpr = kp16 / 8;                                  // 8-element chunks per row
    const int64_t total = M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int k = (int)(i - r * cpr) * 8;
        const bool live = k < K;                               // K % 8 == 0: whole chunk in or out
        const float* p = h + r * ldh + (live ? k : 0);
        typename SP::vec t[SP::NT];
        split_chunk<SP>(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), live, hs, t);
        typename SP::elem* o = planes + r * kp16 + k;
#pragma unroll
        for (int q = 0; q < SP::NT; ++q) *reinterpret_cast<typename SP::vec*>(o + q * M * kp16) = t[q];
    }
}


// ======================================================================================================
// Eight-wave flavour of the f16x2 fused layer (whole-layer launches with the hidden state pre-split).
//
// Why: in the four-wave kernel above a wave owns a 128 x 64 accumulator (244 VGPRs), so a CU holds two waves per
// SIMD and nothing is left to hide the K loop's per-step bubbles (LDS read latency and the barrier in front of every
// 768-cycle MFMA block: 1280 clocks per step measured with one workgroup per CU, 1540-1640 with two) or to run one
// workgroup's spline epilogue under another's MFMAs (matrix pipe 47 % busy).  Here a workgroup is 512 threads on the
// same 128 x 256 tile, a wave owns 64 x 64 (64 accumulator VGPRs, < 128 in all) and a CU holds FOUR waves per SIMD
// from two workgroups; both operands go global -> LDS by LDS-DMA (no staging registers, no ds_write, no VALU in the
// loop), three 24 KB stages deep, one raw s_barrier per 16-wide K step with a counted vmcnt in front of it:
//   wait own pieces of step k | barrier | issue pieces of step k+2 | 8 ds_read_b128 | 12 MFMA
// The LDS image of a stage is lane-linear per 1 KB piece (what an LDS-DMA instruction writes): the A pieces come
// from the "tiled planes" the pre-pass writes ([row tile][step][term][slot][8 halfs], the conflict-free half swap
// already applied), the B pieces are the packed weights as they are ([tile][step][wave][j][term][lane][8]).
constexpr int YT = 512, YM = 128;
constexpr int YA_BYTES = 2 * YM * XS * 2;          // A of one K step: 2 term planes x 128 rows x 16 halfs (8 KB)
constexpr int YB_BYTES = 2 * FN * XS * 2;          // B of one K step: 2 term planes x 256 columns x 16 halfs (16 KB)
constexpr int YSTAGE = YA_BYTES + YB_BYTES;
#ifndef NF_YNS
#define NF_YNS 3
#endif
#ifndef NF_YA_HINT
#define NF_YA_HINT ""
#endif
#ifndef NF_YB_HINT
#define NF_YB_HINT ""
#endif
constexpr int YNS = NF_YNS;

// one LDS-DMA wave-instruction: lane l's 16 (4) bytes at gsrc land at lds_dst + 16 l (4 l); lds_dst wave-uniform.
// M0 is compiler-reserved: saved and restored around the instruction.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" NF_YA_HINT "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16b(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" NF_YB_HINT "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int YFP = 4;                               // features per epilogue pass: waves = 4 features x (2 RT) row groups of 64
// RT = row tiles of 128 samples per workgroup.  RT = 1: 512 threads on 128 x 256 outputs, two workgroups per CU (one's
// epilogue under the other's K loop).  RT = 2 (round 3): 1024 threads on 256 x 256, ONE workgroup per CU, every B stage
// shared by two row tiles: 32 KB instead of 2 x 24 KB of operands through L2 -> LDS per K step and 256 rows.  Knock-out
// builds of the RT = 1 kernel (benchmarks/ab_fused.sh, profiles/r03_a_knockouts.jsonl) showed why: with the MFMAs AND
// the spline evaluation removed the launch still took 2.8 of its 5.65 ms - the LDS-DMA stream alone, 38 GB at the
// 13.7 TB/s this device moves from L2 into its CUs - and 3.8 ms with the DMA removed instead: the K loop was bound
// by the operand stream (24 KB per 768 MFMA cycles and CU), then by power.
__host__ __device__ inline int w8_stage_bytes(int rt) { return rt * YA_BYTES + YB_BYTES; }
__host__ __device__ inline int w8_ring_bytes(int P, int rt = 1) {
    const int tile = YFP * P * (YM * rt + 4) * 4;    // epilogue tile [4 P columns][rows + 4] (aliases the stages)
    const int ring = YNS * w8_stage_bytes(rt);
    return (tile > ring ? tile : ring + 15) / 16 * 16;
}
__host__ __device__ inline int w8_xs_bytes(int fpt, int rt = 1) { return (YM * rt * fpt + 63) / 64 * 256; }   // whole 64-element DMA pieces

__device__ __forceinline__ void wait_vmcnt(int n) {          // counted wait with a wave-uniform runtime count
    switch (n) {                                             // (a count above the last case waits for more than asked: safe)
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    }
}

template <int KT, bool INV, int RT>
__global__ __launch_bounds__(YT * RT, 4) void made_rqs_f16x2_w8_kernel(const FusedArgs g) {
    typedef SplitF16x2 SP;
    typedef SP::vec vec;
    constexpr int NTH = YT * RT, NW = NTH / 64;              // threads, waves (8 or 16)
    constexpr int ROWS = YM * RT, LDTR = ROWS + 4;           // samples per workgroup, stride of the epilogue tile
    constexpr int RG = 2 * RT;                               // row groups of 64 (= waves per feature in an epilogue pass)
    constexpr int STAGE = RT * YA_BYTES + YB_BYTES;
    constexpr int BPW = 16 / NW;                             // B pieces a wave moves per K step (2 or 1), + one A piece
    constexpr int LPS = 1 + BPW;                             // LDS-DMA instructions per wave and K step
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int fpt = g.fpt, P = g.c.P;
    const int ring = w8_ring_bytes(P, RT);
    float* tile = (float*)lds;                               // epilogue: [4 P][LDTR], aliases the stages
    float* xs = (float*)(lds + ring);                        // [ROWS][fpt] inputs, overwritten by outputs
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    int64_t tm;
    int tn;
    tile_of_block(g, tm, tn);
    const int64_t m0 = tm * ROWS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wq = w & 3;
    const int f0 = tn * fpt;
    const int fv = min(fpt, g.feat_limit - f0);
#ifdef NF_STAMP
    long long sp_[12], t_dma = 0, t_bar = 0;
    int spn = 0;
    const long long rt0_ = __builtin_amdgcn_s_memrealtime();
#define YSTAMP() { __builtin_amdgcn_sched_barrier(0); sp_[spn++] = clock64(); __builtin_amdgcn_sched_barrier(0); }
#else
#define YSTAMP()
#endif
    YSTAMP()

    const int ksteps = (g.K + XS - 1) / XS;
    const int kt_begin = g.ktab[2 * tn];
    const int kt_end = min(g.ktab[2 * tn + 1], ksteps);

    // per step wave w moves A piece w (of 8 RT: piece w & 7 of the workgroup's 128-row tile w >> 3; a tile past the
    // end of an odd batch re-reads the last one, its rows are never stored) and B pieces w (+ 8) of 16
    const int64_t tiles128 = (g.M + YM - 1) / YM;
    const int64_t at = min(tm * RT + (w >> 3), tiles128 - 1);
    const char* asrc = (const char*)g.hp + (at * (g.hp_ld / XS)) * YA_BYTES + (w & 7) * 1024 + lane * 16;
    const char* bsrc = (const char*)g.w + ((int64_t)tn * (g.kpad / XS)) * YB_BYTES + w * 1024 + lane * 16;
    auto issue = [&](int kt, int st) {
        if (NF_KO_LOADS) return;
        const unsigned dst = lds_base + st * STAGE + w * 1024;
        glds16(asrc + (int64_t)kt * YA_BYTES, dst);
        glds16b(bsrc + (int64_t)kt * YB_BYTES, dst + RT * YA_BYTES);
        if (BPW == 2) glds16b(bsrc + (int64_t)kt * YB_BYTES + 8192, dst + RT * YA_BYTES + 8192);
    };
#pragma unroll
    for (int q = 0; q < YNS - 1; ++q)
        if (kt_begin + q < kt_end) issue(kt_begin + q, q);
    const int pro = min(YNS - 1, max(kt_end - kt_begin, 0));       // K steps the prologue issued

    // inputs of the tile, straight into LDS (4-byte LDS-DMA, 64 elements per piece), BEHIND the first stages: they are
    // first read in the epilogue, so they must not hold up the first K steps (loads retire in order)
    int nxp = 0;                                                   // pieces this wave issued
    {
        const int nx = ROWS * fpt;
        for (int pc = w; pc * 64 < nx; pc += NW) {
            const int i = min(pc * 64 + lane, nx - 1);
            const int r = i / fpt, f = min(i - r * fpt, fv - 1);
            const int64_t row = min(m0 + r, g.M - 1);
            glds4(g.x + row * g.ldx + (g.feat_index ? g.feat_index[f0 + f] : f0 + f), lds_base + ring + pc * 256);
            ++nxp;
        }
    }
    // the bias joins in the epilogue (its load would otherwise sit in front of the first MFMA), the accumulators start at 0
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = g.bias[tn * FN + wq * 64 + j * 32 + (lane & 31)];

    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment offsets inside a stage: the A image is [128-row tile][term][slot = 2 row + (half ^ bit 3 of the row)][16 B]
    const int frow = lane & 31;
    const int arow = (wr & 1) * 64 + frow;                                                 // row inside its 128-row tile
    const int aoff = (wr >> 1) * YA_BYTES + (arow * 2 + ((lane >> 5) ^ ((frow >> 3) & 1))) * 16;   // + i * 1024 + p * 4096
    const int boff = RT * YA_BYTES + wq * 4096 + lane * 16;                                // + (j * 2 + p) * 1024
    YSTAMP()

    int st = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        // own pieces of step kt landed; what was issued after them may stay in flight: the later steps' pieces (LPS
        // each) and, while the prologue's steps are being consumed, the input pieces issued behind the prologue
        const int later = min(YNS - 2, kt_end - 1 - kt);
#ifdef NF_STAMP
        const long long q0 = clock64();
#endif
        wait_vmcnt(LPS * later + (kt - kt_begin < pro ? nxp : 0));
#ifdef NF_STAMP
        const long long q1 = clock64();
#endif
        // ... and everybody's: after the barrier stage st is complete and nobody reads stage st - 1 any more
        __builtin_amdgcn_s_barrier();
#ifdef NF_STAMP
        t_dma += q1 - q0;
        t_bar += clock64() - q1;
#endif
        if (kt + YNS - 1 < kt_end) issue(kt + YNS - 1, st == 0 ? YNS - 1 : st - 1);
        const char* sp = lds + st * STAGE;
        vec a[2][2], b[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p_ = 0; p_ < 2; ++p_) b[j][p_] = *reinterpret_cast<const vec*>(sp + boff + (j * 2 + p_) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p_ = 0; p_ < 2; ++p_) a[i][p_] = *reinterpret_cast<const vec*>(sp + aoff + i * 1024 + p_ * 4096);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!NF_KO_MMA) SP::fma_tile(acc[i][j], a[i], b[j]);
                else acc[i][j][0] += (float)a[i][0][0] + (float)b[j][1][1];          // (keeps the fragment reads alive)
            }
        st = st == YNS - 1 ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the inputs' DMA when there was no K step at all)
    YSTAMP()

    // ---- epilogue: passes of 4 features x all rows; every wave hands over the columns it holds of the pass's
    //      features, then evaluates ONE (feature, 64-row group) - the same amount of spline work in every wave ---------
    float ws = 1.f, wi = 1.f, hs = 1.f, hi = 1.f;
    pow2_pair(*g.wmax, ws, wi);
    pow2_pair(*g.hmax, hs, hi);
    const float oscale = wi * hi;
    const int fl = w / RG;                                   // wave-uniform feature slot of this wave in every pass
    const int r = (w % RG) * 64 + lane;                      // this lane's sample row in every pass
    float lsum = 0.f;
#pragma unroll 1
    for (int f_pass = 0; f_pass < fv; f_pass += YFP) {
        const int c0 = f_pass * P, c1 = min(f_pass + YFP, fpt) * P;
        __syncthreads();                                     // stages / previous pass fully consumed
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wq * 64 + j * 32 + (lane & 31);
            if (col >= c0 && col < c1) {
                float* tp = tile + (col - c0) * LDTR + wr * 64 + 4 * (lane >> 5);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                        v.x = v.x * oscale + bv[j]; v.y = v.y * oscale + bv[j];
                        v.z = v.z * oscale + bv[j]; v.w = v.w * oscale + bv[j];
                        *reinterpret_cast<float4*>(tp + i * 32 + 8 * q) = v;
                    }
            }
        }
        __syncthreads();
        const int f = f_pass + fl;
        if (f < fv) {
            float yv, ld;
            int bin;
            if (NF_KO_EVAL) { yv = tile[fl * P * LDTR + r]; ld = yv; bin = 0; }
            else rqs_eval<KT, INV, LDTR, true, true, const float*, false>(tile + fl * P * LDTR + r, xs[r * fpt + f], g.c, yv, ld, bin);
            xs[r * fpt + f] = yv;
            lsum += ld;
#ifndef NF_STAMP
            if (g.bins && m0 + r < g.M) g.bins[(m0 + r) * (int64_t)g.D + f0 + f] = bin;
#endif
        }
    }
    YSTAMP()
    __syncthreads();
    float* ldp = tile;                                       // [4 feature slots][ROWS]: the tile is dead now
    ldp[fl * ROWS + r] = lsum;
    __syncthreads();
    if (tid < ROWS) {
        const float s = ((ldp[tid] + ldp[ROWS + tid]) + ldp[2 * ROWS + tid]) + ldp[3 * ROWS + tid];      // fixed order
        if (m0 + tid < g.M) g.partial[(int64_t)tn * g.M + m0 + tid] = s;
    }
    float ymax = 0.f;
    for (int i = tid; i < ROWS * fpt; i += NTH) {
        const int rr = i / fpt, f = i - rr * fpt;
        if (m0 + rr < g.M && f < fv) {
            const float v = xs[i];
            g.y[(m0 + rr) * g.ldy + (g.feat_index ? g.feat_index[f0 + f] : f0 + f)] = v;
            const float a = fabsf(v);
            ymax = fmaxf(ymax, a < INFINITY ? a : 0.f);
        }
    }
    if (g.out_max) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
        if (lane == 0 && __float_as_uint(ymax) > __builtin_nontemporal_load(g.out_max)) atomicMax(g.out_max, __float_as_uint(ymax));
    }
#ifdef NF_STAMP
    YSTAMP()
    if (g.bins && lane == 0) {
        long long* dbg = (long long*)g.bins + ((int64_t)blockIdx.x * NW + w) * 16;
        for (int i = 0; i < 5; ++i) dbg[i] = sp_[i];
        dbg[8] = kt_end - kt_begin;
        dbg[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        dbg[10] = tn;
        dbg[11] = t_dma;
        dbg[12] = t_bar;
        dbg[13] = __builtin_amdgcn_s_memrealtime() - rt0_;
    }
#endif
#undef YSTAMP
}

// ======================================================================================================
// "t" flavours (round 3): the same fused layer with waves of 128 x 64 (128 accumulator registers per lane, two waves
// per SIMD) and - what the register budget of the 64 x 64-wave kernels above has no room for - TWO sets of operand
// fragments: the fragments of K step k+1 are read from LDS, and the LDS-DMA of step k+3 is issued, BETWEEN the MFMAs
// of step k, two items per accumulator tile, so that no MFMA waits on an LDS read and the issue slots of the loads sit
// under the matrix pipe's own cycles.  (In-kernel stamps of the kernels above, benchmarks/diag_stamp_w8.py: per K step
// a wave spends 1340-1400 ticks on issue + reads + its MFMAs and 370-900 at the barrier, against 768 / 1536 cycles of
// MFMA work per SIMD: after every barrier all waves of a workgroup issue their DMA pieces and fragment reads at the
// same time, and the matrix pipe idles until the first fragments are back.)
//   RT = 1 ("t4"): 256 threads, four waves side by side on 128 x 256 outputs, TWO workgroups per CU - every SIMD holds
//           one wave of each, so one workgroup's spline epilogue runs under the other's K loop;
//   RT = 2 ("t8"): 512 threads, 2 x 4 waves on 256 x 256, one workgroup per CU, every B stage shared by two row tiles
//           (32 KB instead of 48 KB through L2 -> LDS per K step and 256 rows).
// Stages: ring of 3 x (RT x 8 + 16) KB; the fragments of stage k are in registers before barrier k (lgkmcnt(0) in
// front of the barrier makes that formal), so the DMA of step k+3 goes into its slot right behind that barrier.
// Epilogue: passes of 4 features x all rows through the (idle) ring, two (feature, 64-row group) evaluations per wave
// as ONE basic block.  Results equal the kernels above bit for bit (same products in the same order per element).
template <int N>
__device__ __forceinline__ void wait_vmcnt_c() {
    static_assert(N == 4 || N == 6, "counts of the t kernels");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

template <int KT, bool INV, int RT>
__global__ __launch_bounds__(256 * RT, 2) void made_rqs_f16x2_t_kernel(const FusedArgs g) {
    typedef SplitF16x2 SP;
    typedef SP::vec vec;
    constexpr int NTH = 256 * RT, NW = 4 * RT, ROWS = YM * RT, LDTR = ROWS + 4;
    constexpr int STAGE = RT * YA_BYTES + YB_BYTES;
    constexpr int LPS = 2 + 16 / NW;                         // LDS-DMA instructions per wave and K step (6 or 4)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int fpt = g.fpt, P = g.c.P;
    const int ring = w8_ring_bytes(P, RT);
    float* tile = (float*)lds;                               // epilogue: [4 P][LDTR], aliases the stages
    float* xs = (float*)(lds + ring);                        // [ROWS][fpt] inputs, overwritten by outputs
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    int64_t tm;
    int tn;
    tile_of_block(g, tm, tn);
    const int64_t m0 = tm * ROWS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wq = w & 3;                       // rows wr * 128 .. + 128, columns wq * 64 .. + 64
    const int f0 = tn * fpt;
    const int fv = min(fpt, g.feat_limit - f0);
#ifdef NF_STAMP
    long long sp_[8];
    int spn = 0;
    const long long rt0_ = __builtin_amdgcn_s_memrealtime();
#define TSTAMP() { __builtin_amdgcn_sched_barrier(0); sp_[spn++] = clock64(); __builtin_amdgcn_sched_barrier(0); }
#else
#define TSTAMP()
#endif
    TSTAMP()

    const int ksteps = (g.K + XS - 1) / XS;
    const int kt_begin = g.ktab[2 * tn];
    const int n = max(min(g.ktab[2 * tn + 1], ksteps) - kt_begin, 0);          // K steps of this tile

    // per step a wave moves RT A pieces (RT = 2: piece w of both 128-row tiles; RT = 1: pieces w, w + 4 of its tile;
    // a tile past the end of an odd batch re-reads the last one, its rows are never stored) and B pieces
    // w, w + NW, ... of 16
    const int64_t tiles128 = (g.M + YM - 1) / YM;
    const int64_t at0 = min(tm * RT, tiles128 - 1), at1 = min(tm * RT + RT - 1, tiles128 - 1);
    const char* asrc0 = (const char*)g.hp + (at0 * (g.hp_ld / XS) + kt_begin) * YA_BYTES + w * 1024 + lane * 16;
    const char* asrc1 = (const char*)g.hp + (at1 * (g.hp_ld / XS) + kt_begin) * YA_BYTES + (RT == 2 ? w : w + 4) * 1024 + lane * 16;
    const char* bsrc = (const char*)g.w + ((int64_t)tn * (g.kpad / XS) + kt_begin) * YB_BYTES + w * 1024 + lane * 16;
    // first / second half of a step's pieces (they are issued between different MFMA groups)
    auto issue_a = [&](int k, int st) {
        if (NF_KO_LOADS) return;
        const unsigned dst = lds_base + st * STAGE;
        glds16(asrc0 + (int64_t)k * YA_BYTES, dst + w * 1024);
        glds16(asrc1 + (int64_t)k * YA_BYTES, dst + (RT == 2 ? YA_BYTES + w * 1024 : (w + 4) * 1024));
        if (RT == 1) glds16b(bsrc + (int64_t)k * YB_BYTES, dst + YA_BYTES + w * 1024);
    };
    auto issue_b = [&](int k, int st) {
        if (NF_KO_LOADS) return;
        const unsigned dst = lds_base + st * STAGE + RT * YA_BYTES + w * 1024;
        if (RT == 2) {
            glds16b(bsrc + (int64_t)k * YB_BYTES, dst);
            glds16b(bsrc + (int64_t)k * YB_BYTES + 8192, dst + 8192);
        } else {
            glds16b(bsrc + (int64_t)k * YB_BYTES + 4096, dst + 4096);
            glds16b(bsrc + (int64_t)k * YB_BYTES + 8192, dst + 8192);
            glds16b(bsrc + (int64_t)k * YB_BYTES + 12288, dst + 12288);
        }
    };
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < n) { issue_a(q, q); issue_b(q, q); }

    // inputs of the tile, straight into LDS (4-byte LDS-DMA, 64 elements per piece), BEHIND the first stages
    int nxp = 0;                                                   // pieces this wave issued
    {
        const int nx = ROWS * fpt;
        for (int pc = w; pc * 64 < nx; pc += NW) {
            const int i = min(pc * 64 + lane, nx - 1);
            const int r = (int)(((unsigned)i * g.fpt_magic) >> 20), f = min(i - r * fpt, fv - 1);
            const int64_t row = min(m0 + r, g.M - 1);
            glds4(g.x + row * g.ldx + (g.feat_index ? g.feat_index[f0 + f] : f0 + f), lds_base + ring + pc * 256);
            ++nxp;
        }
    }
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = g.bias[tn * FN + wq * 64 + j * 32 + (lane & 31)];

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment offsets inside a stage: A image [128-row tile][term][slot = 2 row + (half ^ bit 3 of the row)][16 B]
    const int frow = lane & 31;
    const int aoff = wr * YA_BYTES + (frow * 2 + ((lane >> 5) ^ ((frow >> 3) & 1))) * 16;   // + i * 1024 + p * 4096
    const int boff = RT * YA_BYTES + wq * 4096 + lane * 16;                                 // + (j * 2 + p) * 1024

#define T_SB() __builtin_amdgcn_sched_barrier(0);
#define T_RDB(B_, j_) { B_[j_][0] = *reinterpret_cast<const vec*>(spn_ + boff + ((j_) * 2) * 1024);                 \
                        B_[j_][1] = *reinterpret_cast<const vec*>(spn_ + boff + ((j_) * 2 + 1) * 1024); }
#define T_RDA(A_, i_) { A_[i_][0] = *reinterpret_cast<const vec*>(spn_ + aoff + (i_) * 1024);                       \
                        A_[i_][1] = *reinterpret_cast<const vec*>(spn_ + aoff + (i_) * 1024 + 4096); }
#define T_FT(A_, B_, i_, j_)                                                                               \
    T_SB()                                                                                                 \
    if (!NF_KO_MMA) SP::fma_tile(acc[i_][j_], A_[i_], B_[j_]);                                             \
    else acc[i_][j_][0] += (float)A_[i_][0][0] + (float)B_[j_][1][1];                                      \
    T_SB()
    // one K step: the fragments of step k are in (CA, CB).  Behind the barrier that publishes stage k + 1 (and frees
    // the slot of stage k): the DMA of step k + 3 and the fragment reads of step k + 1 into (NA, NB), two items
    // between consecutive accumulator tiles of step k (the six DMA instructions in two groups of three: one per tile,
    // spread over six tiles, measured 3 % slower).
#define T_STEP(CA, CB, NA, NB)                                                                             \
    {                                                                                                      \
        const bool more = k + 1 < n, dma = k + 3 < n;                                                      \
        const int sn = st == 2 ? 0 : st + 1;                                                               \
        const char* spn_ = lds + sn * STAGE;                                                               \
        if (more) {                                                                                        \
            if (k <= 1) wait_vmcnt((k + 2 < n ? LPS : 0) + nxp);                                           \
            else if (k + 2 < n) wait_vmcnt_c<LPS>();                                                       \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
            __builtin_amdgcn_s_barrier();                                                                  \
        }                                                                                                  \
        if (dma) issue_a(k + 3, st);                                                                       \
        T_FT(CA, CB, 0, 0)                                                                                 \
        if (dma) issue_b(k + 3, st);                                                                       \
        T_FT(CA, CB, 1, 0)                                                                                 \
        if (more) T_RDB(NB, 0)                                                                             \
        T_FT(CA, CB, 2, 0)                                                                                 \
        if (more) T_RDB(NB, 1)                                                                             \
        T_FT(CA, CB, 3, 0)                                                                                 \
        if (more) T_RDA(NA, 0)                                                                             \
        T_FT(CA, CB, 0, 1)                                                                                 \
        if (more) T_RDA(NA, 1)                                                                             \
        T_FT(CA, CB, 1, 1)                                                                                 \
        if (more) T_RDA(NA, 2)                                                                             \
        T_FT(CA, CB, 2, 1)                                                                                 \
        if (more) T_RDA(NA, 3)                                                                             \
        T_FT(CA, CB, 3, 1)                                                                                 \
        st = sn;                                                                                           \
        ++k;                                                                                               \
    }

    vec a0[4][2], b0[2][2], a1[4][2], b1[2][2];
    int k = 0, st = 0;
    TSTAMP()
    if (n > 0) {
        wait_vmcnt(LPS * min(2, n - 1) + nxp);                // own pieces of step 0
        __builtin_amdgcn_s_barrier();                         // ... and everybody's
        {
            const char* spn_ = lds;
            T_RDB(b0, 0) T_RDB(b0, 1) T_RDA(a0, 0) T_RDA(a0, 1) T_RDA(a0, 2) T_RDA(a0, 3)
        }
        while (k < n) {
            T_STEP(a0, b0, a1, b1)
            if (k < n) T_STEP(a1, b1, a0, b0)
        }
    }
#undef T_STEP
#undef T_FT
#undef T_RDA
#undef T_RDB
#undef T_SB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the inputs' DMA when there were fewer than two K steps)
    TSTAMP()
#if NF_KO_EPI
    {   // timing-only build: the K loop alone (the accumulators folded into one value that is never stored in practice)
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_ += acc[i][j][r];
        if (s_ == 12345.678f) g.y[tid] = s_ + bv[0] + bv[1];
        return;
    }
#endif

    // ---- epilogue: passes of 4 features x all rows; every wave hands over the columns it holds of the pass's
    //      features, then evaluates TWO (feature, 64-row group) items of one feature as one basic block ---------------
    float ws = 1.f, wi = 1.f, hs = 1.f, hi = 1.f;
    pow2_pair(*g.wmax, ws, wi);
    float osc[2];                                            // per 64 rows of this wave's 128
    if (g.hmax_rows) {
        const int64_t grp = (m0 >> 6) + 2 * wr;
        pow2_pair(g.hmax[grp], hs, hi);
        osc[0] = wi * hi;
        pow2_pair(g.hmax[grp + 1], hs, hi);
        osc[1] = wi * hi;
    } else {
        pow2_pair(*g.hmax, hs, hi);
        osc[0] = osc[1] = wi * hi;
    }
    // items of a pass: 4 features x 2 RT row groups of 64 = 2 per wave, both of one feature
    const int fl = RT == 2 ? w >> 1 : w;                                     // wave-uniform feature slot
    const int r0 = (RT == 2 ? (w & 1) * 128 : 0) + lane, r1 = r0 + 64;       // this lane's two sample rows
    float lsum0 = 0.f, lsum1 = 0.f;
    // kept parameters: columns lane and lane + 64 of a pass (YFP * P <= 128 of them), 1 or the un-fold factor
    float keep_scale[2] = {1.f, 1.f};
    if (g.params_out) {
        keep_scale[0] = (lane % P) < 2 * g.c.K ? g.unfold : 1.f;
        keep_scale[1] = ((lane + 64) % P) < 2 * g.c.K ? g.unfold : 1.f;
    }
#pragma unroll 1
    for (int f_pass = 0; f_pass < fv; f_pass += YFP) {
        const int c0 = f_pass * P, c1 = min(f_pass + YFP, fpt) * P;
        // (kept parameters: LDS-only barriers - __syncthreads() also drains vmcnt, i.e. would wait out the previous
        //  pass's parameter stores, a write latency per pass with nothing else to run)
        if (g.params_out) lds_only_barrier(); else
        __syncthreads();                                     // stages / previous pass fully consumed
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wq * 64 + j * 32 + (lane & 31);
            if (col >= c0 && col < c1) {
                float* tp = tile + (col - c0) * LDTR + wr * 128 + 4 * (lane >> 5);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float oscale = osc[i >> 1];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                        v.x = v.x * oscale + bv[j]; v.y = v.y * oscale + bv[j];
                        v.z = v.z * oscale + bv[j]; v.w = v.w * oscale + bv[j];
                        *reinterpret_cast<float4*>(tp + i * 32 + 8 * q) = v;
                    }
                }
            }
        }
        if (g.params_out) lds_only_barrier(); else
        __syncthreads();
        if (g.params_out) {
            // training forward (nf_made_rqs_fwd_params, f16x2 packs): the pass's parameters - whole features, c1 - c0
            // consecutive floats per row - leave as they lie in the tile, a wave per row and lanes along the row
            // (coalesced runs of up to 256 bytes), the logit columns un-folded (the pack multiplied them by log2(e) /
            // logit_div).  Issued in front of the evaluation, which does not wait for them.
            const int nc = min(c1, fv * P) - c0;
            for (int rr = w; rr < ROWS; rr += NW) {
                const int64_t row = m0 + rr;
                if (row < g.M) {
                    float* po = g.params_out + row * g.ldp + (int64_t)(f0 + f_pass) * P;
                    if (lane < nc) po[lane] = tile[lane * LDTR + rr] * keep_scale[0];
                    if (lane + 64 < nc) po[lane + 64] = tile[(lane + 64) * LDTR + rr] * keep_scale[1];
                }
            }
        }
        const int f = f_pass + fl;
#if NF_EPI_PRIO
        __builtin_amdgcn_s_setprio(NF_EPI_PRIO);             // the evaluation's VALU stream ahead of the partner workgroup's issue
#endif
        if (f < fv) {
            float yv0, ld0, yv1, ld1;
            int bin0, bin1;
            if (NF_KO_EVAL) { yv0 = tile[fl * P * LDTR + r0]; ld0 = yv0; bin0 = 0; yv1 = tile[fl * P * LDTR + r1]; ld1 = yv1; bin1 = 0; }
            else {
                rqs_eval<KT, INV, LDTR, true, true, const float*, false>(tile + fl * P * LDTR + r0, xs[r0 * fpt + f], g.c, yv0, ld0, bin0);
                rqs_eval<KT, INV, LDTR, true, true, const float*, false>(tile + fl * P * LDTR + r1, xs[r1 * fpt + f], g.c, yv1, ld1, bin1);
            }
            xs[r0 * fpt + f] = yv0;
            xs[r1 * fpt + f] = yv1;
            lsum0 += ld0;
            lsum1 += ld1;
#ifndef NF_STAMP
            if (g.bins) {
                if (m0 + r0 < g.M) g.bins[(m0 + r0) * (int64_t)g.D + f0 + f] = bin0;
                if (m0 + r1 < g.M) g.bins[(m0 + r1) * (int64_t)g.D + f0 + f] = bin1;
            }
#endif
        }
#if NF_EPI_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
    TSTAMP()
    __syncthreads();
    float* ldp = tile;                                       // [4 feature slots][ROWS]: the tile is dead now
    ldp[fl * ROWS + r0] = lsum0;
    ldp[fl * ROWS + r1] = lsum1;
    __syncthreads();
    if (tid < ROWS) {
        const float s = ((ldp[tid] + ldp[ROWS + tid]) + ldp[2 * ROWS + tid]) + ldp[3 * ROWS + tid];      // fixed order
        if (m0 + tid < g.M) g.partial[(int64_t)tn * g.M + m0 + tid] = s;
    }
    float ymax = 0.f;
    for (int i = tid; i < ROWS * fpt; i += NTH) {
        const int rr = (int)(((unsigned)i * g.fpt_magic) >> 20), f = i - rr * fpt;
        if (m0 + rr < g.M && f < fv) {
            const float v = xs[i];
            g.y[(m0 + rr) * g.ldy + (g.feat_index ? g.feat_index[f0 + f] : f0 + f)] = v;
            const float a = fabsf(v);
            ymax = fmaxf(ymax, a < INFINITY ? a : 0.f);
        }
    }
    if (g.out_max) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
        if (lane == 0 && __float_as_uint(ymax) > __builtin_nontemporal_load(g.out_max)) atomicMax(g.out_max, __float_as_uint(ymax));
    }
#ifdef NF_STAMP
    TSTAMP()
    if (g.bins && lane == 0) {
        long long* dbg = (long long*)g.bins + ((int64_t)blockIdx.x * NW + w) * 16;
        for (int i = 0; i < 5; ++i) dbg[i] = sp_[i];
        dbg[8] = n;
        dbg[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        dbg[10] = tn;
        dbg[11] = 0;
        dbg[12] = 0;
        dbg[13] = __builtin_amdgcn_s_memrealtime() - rt0_;
    }
#endif
#undef TSTAMP
}

// h[M][K] fp32 -> the eight-wave kernel's A stream: [row tile of 128][K step][term][slot][8 halfs], slot = 2 * row +
// (k half ^ bit 3 of the row): one K step of one row tile is 8 KB, contiguous, already in its LDS image order
template <class SP>
__global__ __launch_bounds__(256) void split_planes_tiled_kernel(const float* __restrict__ h, int64_t ldh, int64_t M,
                                                                 int K, int ksteps, const unsigned* __restrict__ hmax,
                                                                 typename SP::elem* __restrict__ planes) {
    float hs = 1.f, hi = 1.f;
    if (SP::SCALED) pow2_pair(*hmax, hs, hi);
    const int cpr = ksteps * 2;                                // 8-element chunks per row
    const int64_t rows = (M + YM - 1) / YM * YM;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int c = (int)(i - row * cpr);
        const int kt = c >> 1, half = c & 1, k = c * 8;
        const bool live = row < M && k < K;                    // K % 8 == 0: a chunk is wholly in or out
        const float* p = h + (live ? row * ldh + k : 0);
        typename SP::vec t[SP::NT];
        split_chunk<SP>(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), live, hs, t);
        const int rl = (int)(row & (YM - 1));
        const int slot = rl * 2 + (half ^ ((rl >> 3) & 1));
        typename SP::elem* o = planes + (((row >> 7) * ksteps + kt) * SP::NT * (2 * YM) + slot) * 8;
#pragma unroll
        for (int q = 0; q < SP::NT; ++q) *reinterpret_cast<typename SP::vec*>(o + q * (2 * YM) * 8) = t[q];
    }
}

inline int split_terms(int dtype) { return dtype == NF_DTYPE_BF16X3 ? 3 : dtype == NF_DTYPE_F16X2 ? 2 : 0; }
inline int kp16_of(int in) { return (in + XS - 1) / XS * XS; }
// (rows rounded up to whole 128-row tiles: the tiled layout of the eight-wave kernel)
inline size_t planes_bytes(int64_t B, int in, int dtype) {
    return (size_t)split_terms(dtype) * (size_t)((B + YM - 1) / YM * YM) * kp16_of(in) * 2;
}

inline size_t lds_bytes_split(int fpt, int P, int nt) {
    const size_t tile = sizeof(float) * (size_t)fpt * P * XLDT;
    const size_t stages = (size_t)2 * 2 * nt * XA;
    return (tile > stages ? tile : stages) + sizeof(float) * ((size_t)XM * fpt + (size_t)XM * (XT / 64));
}

template <class SP, int KT, bool INV, bool PRE>
int launch_one_split(const FusedArgs& g, int grid, size_t lds, hipStream_t s) {
    static unsigned long long cap_done = 0;
    const int rc = nf_raise_lds_cap((const void*)made_rqs_split_kernel<SP, KT, INV, PRE>, &cap_done);
    if (rc != NF_OK) return rc;
    hipLaunchKernelGGL((made_rqs_split_kernel<SP, KT, INV, PRE>), dim3(grid), dim3(XT), lds, s, g);
    return nf_launch_status();
}

template <class SP, bool INV, bool PRE>
int launch_k_split(const FusedArgs& g, int grid, size_t lds, hipStream_t s) {
    switch (g.c.K) {
        case 4: return launch_one_split<SP, 4, INV, PRE>(g, grid, lds, s);
        case 8: return launch_one_split<SP, 8, INV, PRE>(g, grid, lds, s);
        case 10: return launch_one_split<SP, 10, INV, PRE>(g, grid, lds, s);
        default: return launch_one_split<SP, 0, INV, PRE>(g, grid, lds, s);
    }
}

template <int KT, bool INV, int RT>
int launch_one_w8(const FusedArgs& g, int grid, size_t lds, hipStream_t s) {
    static unsigned long long cap_done = 0;
    const int rc = nf_raise_lds_cap((const void*)made_rqs_f16x2_w8_kernel<KT, INV, RT>, &cap_done);
    if (rc != NF_OK) return rc;
    hipLaunchKernelGGL((made_rqs_f16x2_w8_kernel<KT, INV, RT>), dim3(grid), dim3(YT * RT), lds, s, g);
    return nf_launch_status();
}

template <bool INV, int RT>
int launch_k_w8(const FusedArgs& g, int grid, size_t lds, hipStream_t s) {
    switch (g.c.K) {
        case 4: return launch_one_w8<4, INV, RT>(g, grid, lds, s);
        case 8: return launch_one_w8<8, INV, RT>(g, grid, lds, s);
        case 10: return launch_one_w8<10, INV, RT>(g, grid, lds, s);
        default: return launch_one_w8<0, INV, RT>(g, grid, lds, s);
    }
}

template <int KT, bool INV, int RT>
int launch_one_t(const FusedArgs& g, int grid, size_t lds, hipStream_t s) {
    static unsigned long long cap_done = 0;
    const int rc = nf_raise_lds_cap((const void*)made_rqs_f16x2_t_kernel<KT, INV, RT>, &cap_done);
    if (rc != NF_OK) return rc;
    hipLaunchKernelGGL((made_rqs_f16x2_t_kernel<KT, INV, RT>), dim3(grid), dim3(256 * RT), lds, s, g);
    return nf_launch_status();
}

template <bool INV, int RT>
int launch_k_t(const FusedArgs& g, int grid, size_t lds, hipStream_t s) {
    switch (g.c.K) {
        case 4: return launch_one_t<4, INV, RT>(g, grid, lds, s);
        case 8: return launch_one_t<8, INV, RT>(g, grid, lds, s);
        case 10: return launch_one_t<10, INV, RT>(g, grid, lds, s);
        default: return launch_one_t<0, INV, RT>(g, grid, lds, s);
    }
}

// Which flavour of the f16x2 whole-layer kernel runs: 1 = "w8", eight waves of 64 x 64 on 128 x 256 tiles (two
// workgroups per CU); 2 = "w16", sixteen such waves on 256 x 256; 3 = "t8", eight waves of 128 x 64 on 256 x 256 with
// register-pipelined fragments; 4 = "t4", four such waves on 128 x 256, two workgroups per CU.
// NF_FUSED_F16X2_KERNEL=w8|w16|t8|t4 in the environment overrides the rule (A/B runs and the tests that compare the
// flavours: they give bit-identical results - the same products in the same order per element).
inline int w8_flavour(int64_t B, int P, int fpt) {
    int fl = B > 512 ? NF_F16X2_DEFAULT_FLAVOUR : 1;
    if (const char* e = getenv("NF_FUSED_F16X2_KERNEL"))
        fl = !strcmp(e, "w8") ? 1 : !strcmp(e, "w16") ? 2 : !strcmp(e, "t8") ? 3 : !strcmp(e, "t4") ? 4 : fl;
    if ((fl == 2 || fl == 3) && w8_ring_bytes(P, 2) + w8_xs_bytes(fpt, 2) > 160 * 1024) fl = 1;
    return fl;
}

// the split pre-pass (when the workspace has room for the planes and the launch covers the layer) + the kernel
inline int maxabs_rows(const float* h, int64_t ld, int64_t B, int cols, unsigned* slot, hipStream_t s) {
    hipError_t me = hipMemsetAsync(slot, 0, sizeof(unsigned), s);
    if (me != hipSuccess) return (int)me;
    const int64_t q4 = B * ((cols + 3) / 4);
    const int mgrid = (int)((q4 + 255) / 256 < 2048 ? (q4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(maxabs_rows_kernel, dim3(mgrid), dim3(256), 0, s, h, ld, B, cols, slot);
    return nf_launch_status();
}

template <class SP>
int run_split(FusedArgs& g, int grid, size_t lds, bool pre, bool own_max, const float* h, int64_t h_row_stride,
              int64_t B, int in, int inverse, hipStream_t s) {
    if (SP::SCALED && own_max) {                       // no producer left max |h| in the slot: find it
        const int rc = maxabs_rows(h, h_row_stride, B, in, const_cast<unsigned*>(g.hmax), s);
        if (rc != NF_OK) return rc;
    }
    if (pre && SP::NT == 2 && w8_ring_bytes(g.c.P) + w8_xs_bytes(g.fpt) <= 160 * 1024) {
        // f16x2, whole layer: tiled planes + the eight-wave kernel (sixteen waves on 256-row tiles for real batches)
        const int ksteps = (int)(g.hp_ld / XS);
        const int64_t chunks = (B + YM - 1) / YM * YM * ksteps * 2;
        const int sgrid = (int)((chunks + 255) / 256 < 65536 ? (chunks + 255) / 256 : 65536);
        hipLaunchKernelGGL(split_planes_tiled_kernel<SplitF16x2>, dim3(sgrid), dim3(256), 0, s, h, h_row_stride, B, in,
                           ksteps, g.hmax, (_Float16*)g.hp);
        const int flavour = w8_flavour(B, g.c.P, g.fpt);
        const int rt = flavour == 1 || flavour == 4 ? 1 : 2;
        const size_t lds8 = (size_t)w8_ring_bytes(g.c.P, rt) + w8_xs_bytes(g.fpt, rt);
        if (rt == 2) {
            g.m_tiles = (B + 2 * YM - 1) / (2 * YM);
            g.group_m = 4;
            if (const char* e = getenv("NF_W8_GROUP_M")) g.group_m = atoi(e) > 0 ? atoi(e) : 4;
            const int64_t grid2 = g.m_tiles * g.n_tiles;
            if (flavour == 3)
                return inverse ? launch_k_t<true, 2>(g, (int)grid2, lds8, s) : launch_k_t<false, 2>(g, (int)grid2, lds8, s);
            return inverse ? launch_k_w8<true, 2>(g, (int)grid2, lds8, s) : launch_k_w8<false, 2>(g, (int)grid2, lds8, s);
        }
        if (flavour == 4)
            return inverse ? launch_k_t<true, 1>(g, grid, lds8, s) : launch_k_t<false, 1>(g, grid, lds8, s);
        return inverse ? launch_k_w8<true, 1>(g, grid, lds8, s) : launch_k_w8<false, 1>(g, grid, lds8, s);
    }
    if (pre) {
        const int64_t chunks = B * (g.hp_ld / 8);
        const int sgrid = (int)((chunks + 255) / 256 < 65536 ? (chunks + 255) / 256 : 65536);
        hipLaunchKernelGGL(split_planes_kernel<SP>, dim3(sgrid), dim3(256), 0, s, h, h_row_stride, B, in, (int)g.hp_ld,
                           g.hmax, (typename SP::elem*)g.hp);
        return inverse ? launch_k_split<SP, true, true>(g, grid, lds, s) : launch_k_split<SP, false, true>(g, grid, lds, s);
    }
    return inverse ? launch_k_split<SP, true, false>(g, grid, lds, s) : launch_k_split<SP, false, false>(g, grid, lds, s);
}

inline int kpad_of(int in) { return (in + KPAD - 1) / KPAD * KPAD; }
inline int fpt_of(int P) { return FN / P; }
inline bool fused_supported(int K, int tails) {   // (and the hidden width must be a multiple of 8)
    const int P = tails ? 3 * K - 1 : 3 * K + 1;
    return K >= 4 && K <= 32 && P <= FN;
}
inline size_t lds_bytes(int fpt, int P) {
    const size_t staging = sizeof(float) * (size_t)(FM + FN) * LDA;        // >= the bf16 staging size
    const size_t tile = sizeof(float) * (size_t)fpt * P * LDT;
    return (tile > staging ? tile : staging) + sizeof(float) * ((size_t)FM * fpt + (size_t)FM * (FT / 64));
}

template <bool BF16, int KT, bool INV>
int launch_one(const FusedArgs& g, int grid, size_t lds, hipStream_t s) {
    static unsigned long long cap_done[2] = {0, 0};     // dynamic-LDS cap: once per instantiation and device
    if (!INV && g.params_out) {
        const int rc = nf_raise_lds_cap((const void*)made_rqs_kernel<BF16, KT, false, true>, &cap_done[1]);
        if (rc != NF_OK) return rc;
        hipLaunchKernelGGL((made_rqs_kernel<BF16, KT, false, true>), dim3(grid), dim3(FT), lds, s, g);
        return nf_launch_status();
    }
    const int rc = nf_raise_lds_cap((const void*)made_rqs_kernel<BF16, KT, INV>, &cap_done[0]);
    if (rc != NF_OK) return rc;
    hipLaunchKernelGGL((made_rqs_kernel<BF16, KT, INV>), dim3(grid), dim3(FT), lds, s, g);
    return nf_launch_status();
}

template <bool BF16, bool INV>
int launch_k(const FusedArgs& g, int grid, size_t lds, hipStream_t s) {
    switch (g.c.K) {
        case 4: return launch_one<BF16, 4, INV>(g, grid, lds, s);
        case 8: return launch_one<BF16, 8, INV>(g, grid, lds, s);
        case 10: return launch_one<BF16, 10, INV>(g, grid, lds, s);
        case 16: return launch_one<BF16, 16, INV>(g, grid, lds, s);
        default: return launch_one<BF16, 0, INV>(g, grid, lds, s);
    }
}

}  // namespace

extern "C" int nf_made_rqs_supported(int K, int tails) { return fused_supported(K, tails) ? 1 : 0; }

extern "C" int nf_made_rqs_tiles(int features, int K, int tails) {
    if (!fused_supported(K, tails) || features <= 0) return 0;
    const int fpt = fpt_of(tails ? 3 * K - 1 : 3 * K + 1);
    return (features + fpt - 1) / fpt;
}

extern "C" size_t nf_made_rqs_packed_bytes(int features, int in, int K, int tails, int dtype) {
    const int nt = nf_made_rqs_tiles(features, K, tails);
    if (nt == 0 || in <= 0) return 0;
    return (size_t)nt * FN * kpad_of(in) * (dtype == NF_DTYPE_BF16 ? 2 : split_terms(dtype) ? 2 * split_terms(dtype) : 4);
}

extern "C" size_t nf_made_rqs_workspace_bytes(int64_t B, int features, int K, int tails) {
    const int nt = nf_made_rqs_tiles(features, K, tails);
    return nt == 0 || B <= 0 ? 0 : sizeof(float) * (size_t)nt * (size_t)B;
}

extern "C" size_t nf_made_rqs_workspace_bytes_dtype(int64_t B, int features, int in, int K, int tails, int dtype) {
    const size_t base = nf_made_rqs_workspace_bytes(B, features, K, tails);
    if (base == 0 || !split_terms(dtype) || in <= 0) return base;
    return (base + 255) / 256 * 256 + 256 + planes_bytes(B, in, dtype);   // + scale slot + term planes       // + the hidden state as three bf16 term planes
}

extern "C" int nf_made_rqs_pack_scaled(const float* W, const float* mask, const float* bias, int features, int in,
                                       int K, int tails, int dtype, double logit_div, void* packed, int32_t* ktab,
                                       float* bias_packed, void* stream) {
    if (!W || !packed || !bias_packed) return NF_ERR_NULL;     // ktab == NULL: keep the caller's table
    if (features <= 0 || in <= 0 || !fused_supported(K, tails) || !(logit_div > 0.0)) return NF_ERR_SHAPE;
    const float logit_fold = (float)(1.4426950408889634 / logit_div);
    if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_BF16 && !split_terms(dtype)) return NF_ERR_DTYPE;
    const int P = tails ? 3 * K - 1 : 3 * K + 1;
    const int fpt = fpt_of(P), nt = (features + fpt - 1) / fpt, out = features * P, kpad = kpad_of(in);
    const int64_t total = (int64_t)nt * FN * kpad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == NF_DTYPE_BF16X3) {
        hipLaunchKernelGGL(pack_fused_split_kernel<SplitBf16x3>, dim3(grid), dim3(256), 0, s, W, mask, bias, out, in,
                           fpt * P, kpad, 3 * total, P, 2 * K, logit_fold, (const unsigned*)nullptr, (__bf16*)packed,
                           bias_packed, 0);
    } else if (dtype == NF_DTYPE_F16X2) {
        if (!ktab) return NF_ERR_NULL;                 // the scale pair lives after the table
        unsigned* slot = (unsigned*)(ktab + 2 * nt);
        hipError_t me = hipMemsetAsync(slot, 0, 2 * sizeof(unsigned), s);
        if (me != hipSuccess) return (int)me;
        const int64_t nw = (int64_t)out * in;
        const int mgrid = out < 1024 ? out : 1024;
        hipLaunchKernelGGL(maxabs_kernel, dim3(mgrid), dim3(256), 0, s, W, mask, nw, P, in, 2 * K, logit_fold, slot);
        hipLaunchKernelGGL(pack_fused_split_kernel<SplitF16x2>, dim3(grid), dim3(256), 0, s, W, mask, bias, out, in,
                           fpt * P, kpad, 2 * total, P, 2 * K, logit_fold, (const unsigned*)slot, (_Float16*)packed,
                           bias_packed, 0);
    } else if (dtype == NF_DTYPE_F32)
        hipLaunchKernelGGL(pack_fused_kernel<float>, dim3(grid), dim3(256), 0, s, W, mask, bias, out, in, fpt * P, kpad,
                           total, P, 2 * K, logit_fold, (float*)packed, bias_packed);
    else
        hipLaunchKernelGGL(pack_fused_kernel<__bf16>, dim3(grid), dim3(256), 0, s, W, mask, bias, out, in, fpt * P, kpad,
                           total, P, 2 * K, logit_fold, (__bf16*)packed, bias_packed);
    if (ktab) launch_ktab_fused(mask, out, in, fpt * P, nt, ktab, 0, s);
    return nf_launch_status();
}

extern "C" int nf_made_rqs_pack(const float* W, const float* mask, const float* bias, int features, int in, int K,
                                int tails, int dtype, void* packed, int32_t* ktab, float* bias_packed, void* stream) {
    return nf_made_rqs_pack_scaled(W, mask, bias, features, in, K, tails, dtype, 1.0, packed, ktab, bias_packed, stream);
}

static int fused_fwd_impl(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                          const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                          int64_t y_row_stride, const int32_t* feat_index, int row_width, float* logabsdet,
                          int accumulate, int32_t* bin_idx, float* workspace, size_t workspace_bytes, int64_t B, int in,
                          int features, int K, int tails, double tail_bound, double min_bin_width,
                          double min_bin_height, double min_derivative, int inverse, int dtype, int tile_begin,
                          int tile_count, int feat_limit, int reduce, const unsigned* h_max, unsigned* y_max,
                          void* stream, float* params_out = nullptr, int64_t params_row_stride = 0,
                          double logit_div = 1.0, const void* planes_in = nullptr, const unsigned* row_scale = nullptr,
                          int params_bf16 = 0) {
    if (B < 0 || in <= 0 || features <= 0 || !fused_supported(K, tails)) return NF_ERR_SHAPE;
    if (planes_in) {
        // the hidden state arrives as tiled fp16 term planes with one scale per 64 rows (nf_made_hidden_planes_fwd)
        if (dtype != NF_DTYPE_F16X2) return NF_ERR_DTYPE;
        if (!row_scale) return NF_ERR_NULL;
        if (((uintptr_t)planes_in & 15) != 0 || feat_index || params_out) return NF_ERR_LAYOUT;
        h = (const float*)planes_in;                 // (only so that the null / alignment checks below pass: never read)
        h_row_stride = (in + 3) / 4 * 4;
    }
    if (params_out && ((split_terms(dtype) && (dtype != NF_DTYPE_F16X2 || params_bf16)) || inverse || feat_index ||
                       params_row_stride < (int64_t)features * (tails ? 3 * K - 1 : 3 * K + 1)))
        return NF_ERR_LAYOUT;                       // (the f32 / bf16 forward kernel of whole autoregressive layers, or
                                                    //  the f16x2 t4 kernel with the planes in the workspace - below)
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_BF16 && !split_terms(dtype)) return NF_ERR_DTYPE;
    if (y_max) {
        hipError_t me = hipMemsetAsync(y_max, 0, sizeof(unsigned), (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
    }
    if (B == 0) return NF_OK;
    if (!h || !packed_w || !ktab || !bias_packed || !x || !y || !logabsdet || !workspace) return NF_ERR_NULL;
    if (h_row_stride < in || x_row_stride < row_width || y_row_stride < row_width) return NF_ERR_LAYOUT;
    // the hidden state is staged with 16-byte loads only (it comes from nf_masked_linear_fwd)
    if (in % 8 != 0 || h_row_stride % 4 != 0 || ((uintptr_t)h & 15) != 0) return NF_ERR_LAYOUT;
    if (workspace_bytes < nf_made_rqs_workspace_bytes(B, features, K, tails)) return NF_ERR_SCRATCH;

    FusedArgs g;
    g.c = nf_make_rqs_const(K, tails, tail_bound, min_bin_width, min_bin_height, min_derivative, 1.0);
    g.h = h; g.ldh = h_row_stride; g.w = packed_w; g.kpad = kpad_of(in); g.ktab = ktab; g.bias = bias_packed;
    g.x = x; g.ldx = x_row_stride; g.y = y; g.ldy = y_row_stride; g.partial = workspace; g.bins = bin_idx; g.feat_index = feat_index; g.hp = nullptr; g.hp_ld = 0; g.hmax = nullptr; g.hmax_rows = 0; g.residual = nullptr; g.ldr = 0; g.add2 = nullptr; g.ldadd2 = 0; g.N = 0; g.relu_out = 0; g.out_max = y_max; g.vec_out = 0; g.k_reverse = 0;
    g.M = B; g.K = in; g.D = features; g.fpt = fpt_of(g.c.P);
    g.fpt_magic = ((1u << 20) + g.fpt - 1) / g.fpt;
    g.params_out = params_out; g.ldp = params_row_stride; g.unfold = (float)(logit_div * 0.6931471805599453);
    g.params_bf16 = params_bf16;
    const int all_tiles = (features + g.fpt - 1) / g.fpt;
    if (tile_begin < 0 || tile_count < 0 || tile_begin + tile_count > all_tiles) return NF_ERR_SHAPE;
    if (feat_limit > features || (tile_count > 0 && feat_limit <= (tile_begin + tile_count - 1) * g.fpt)) return NF_ERR_SHAPE;
    g.n_tiles = tile_count;
    g.tn0 = tile_begin;
    g.feat_limit = feat_limit;
    const bool x3 = split_terms(dtype) != 0;
    g.wmax = (const unsigned*)(ktab + 2 * all_tiles);
    g.m_tiles = x3 ? (B + XM - 1) / XM : (B + FM - 1) / FM;
    g.group_m = x3 ? 8 : 16;
    const int64_t grid = g.m_tiles * g.n_tiles;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    const size_t lds = x3 ? lds_bytes_split(g.fpt, g.c.P, split_terms(dtype)) : lds_bytes(g.fpt, g.c.P);
    if (lds > 160 * 1024) return NF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    int rc = NF_OK;
    if (grid > 0) {
        if (x3) {
            // whole-layer launches split the hidden state once (72 column tiles re-read it); the windowed
            // launches of the incremental inverse split it on the way into LDS instead
            const size_t part = (nf_made_rqs_workspace_bytes(B, features, K, tails) + 255) / 256 * 256;
            if (workspace_bytes < part + 256) return NF_ERR_SCRATCH;      // the hidden-state scale slot
            g.hmax = h_max ? h_max : (const unsigned*)((const char*)workspace + part);
            if (planes_in) {
                if (tile_count != all_tiles || feat_limit != features) return NF_ERR_SHAPE;
                if (w8_ring_bytes(g.c.P) + w8_xs_bytes(g.fpt) > 160 * 1024) return NF_ERR_SHAPE;
                g.hp = planes_in;
                g.hp_ld = kp16_of(in);
                g.hmax = row_scale;
                g.hmax_rows = 1;
                const size_t lds8 = (size_t)w8_ring_bytes(g.c.P, 1) + w8_xs_bytes(g.fpt, 1);
                rc = inverse ? launch_k_t<true, 1>(g, (int)grid, lds8, s) : launch_k_t<false, 1>(g, (int)grid, lds8, s);
                if (rc != NF_OK || !reduce) return rc;
                hipLaunchKernelGGL(rowsum_kernel, dim3(nf_ceil_div(B, kRowsumThreads)), dim3(kRowsumThreads), 0, s, workspace, all_tiles, B,
                                   logabsdet, accumulate);
                return nf_launch_status();
            }
            const bool pre = tile_count == all_tiles && workspace_bytes >= part + 256 + planes_bytes(B, in, dtype);
            if (pre) {
                g.hp = (const char*)workspace + part + 256;
                g.hp_ld = kp16_of(in);
            }
            if (params_out) {
                // the kept-parameters forward exists in the t4 kernel only: whole layer, planes in the workspace
                // (nf_made_rqs_workspace_bytes_dtype), ring within LDS - else the caller's two-launch route
                if (!pre || feat_limit != features || w8_ring_bytes(g.c.P) + w8_xs_bytes(g.fpt) > 160 * 1024)
                    return NF_ERR_SCRATCH;
                if (!h_max) {
                    const int rm = maxabs_rows(h, h_row_stride, B, in, const_cast<unsigned*>(g.hmax), s);
                    if (rm != NF_OK) return rm;
                }
                const int ksteps = (int)(g.hp_ld / XS);
                const int64_t chunks = (B + YM - 1) / YM * YM * ksteps * 2;
                const int sgrid = (int)((chunks + 255) / 256 < 65536 ? (chunks + 255) / 256 : 65536);
                hipLaunchKernelGGL(split_planes_tiled_kernel<SplitF16x2>, dim3(sgrid), dim3(256), 0, s, h, h_row_stride, B,
                                   in, ksteps, g.hmax, (_Float16*)g.hp);
                const size_t lds8 = (size_t)w8_ring_bytes(g.c.P, 1) + w8_xs_bytes(g.fpt, 1);
                rc = launch_k_t<false, 1>(g, (int)grid, lds8, s);
            } else
            rc = dtype == NF_DTYPE_BF16X3 ? run_split<SplitBf16x3>(g, (int)grid, lds, pre, h_max == nullptr, h, h_row_stride, B, in, inverse, s)
                                          : run_split<SplitF16x2>(g, (int)grid, lds, pre, h_max == nullptr, h, h_row_stride, B, in, inverse, s);
        }
        else if (dtype == NF_DTYPE_BF16)
            rc = inverse ? launch_k<true, true>(g, (int)grid, lds, s) : launch_k<true, false>(g, (int)grid, lds, s);
        else
            rc = inverse ? launch_k<false, true>(g, (int)grid, lds, s) : launch_k<false, false>(g, (int)grid, lds, s);
    }
    if (rc != NF_OK || !reduce) return rc;
    hipLaunchKernelGGL(rowsum_kernel, dim3(nf_ceil_div(B, kRowsumThreads)), dim3(kRowsumThreads), 0, s, workspace, all_tiles, B, logabsdet,
                       accumulate);
    return nf_launch_status();
}

extern "C" int nf_made_rqs_fwd_tiles(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                                     const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                                     int64_t y_row_stride, float* logabsdet, int accumulate, int32_t* bin_idx,
                                     float* workspace, size_t workspace_bytes, int64_t B, int in, int features, int K,
                                     int tails, double tail_bound, double min_bin_width, double min_bin_height,
                                     double min_derivative, int inverse, int dtype, int tile_begin, int tile_count,
                                     int feat_limit, int reduce, void* stream) {
    return fused_fwd_impl(h, h_row_stride, packed_w, ktab, bias_packed, x, x_row_stride, y, y_row_stride, nullptr,
                          features, logabsdet, accumulate, bin_idx, workspace, workspace_bytes, B, in, features, K, tails,
                          tail_bound, min_bin_width, min_bin_height, min_derivative, inverse, dtype, tile_begin,
                          tile_count, feat_limit, reduce, nullptr, nullptr, stream);
}

extern "C" int nf_coupling_rqs_fwd(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                                   const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                                   int64_t y_row_stride, const int32_t* feat_index, int row_width, float* logabsdet,
                                   int accumulate, int32_t* bin_idx, float* workspace, size_t workspace_bytes,
                                   int64_t B, int in, int features, int K, int tails, double tail_bound,
                                   double min_bin_width, double min_bin_height, double min_derivative, int inverse,
                                   int dtype, void* stream) {
    if (!feat_index) return NF_ERR_NULL;
    if (row_width < features) return NF_ERR_SHAPE;
    return fused_fwd_impl(h, h_row_stride, packed_w, ktab, bias_packed, x, x_row_stride, y, y_row_stride, feat_index,
                          row_width, logabsdet, accumulate, bin_idx, workspace, workspace_bytes, B, in, features, K,
                          tails, tail_bound, min_bin_width, min_bin_height, min_derivative, inverse, dtype, 0,
                          nf_made_rqs_tiles(features, K, tails), features, 1, nullptr, nullptr, stream);
}

// nf_made_rqs_fwd with the producer's max |h| slot (split-operand dtypes: skips the max pre-pass)
extern "C" int nf_made_rqs_fwd_hmax(const float* h, int64_t h_row_stride, const uint32_t* h_max, const void* packed_w,
                                    const int32_t* ktab, const float* bias_packed, const float* x, int64_t x_row_stride,
                                    float* y, int64_t y_row_stride, uint32_t* y_max, float* logabsdet, int accumulate, int32_t* bin_idx,
                                    float* workspace, size_t workspace_bytes, int64_t B, int in, int features, int K,
                                    int tails, double tail_bound, double min_bin_width, double min_bin_height,
                                    double min_derivative, int inverse, int dtype, void* stream) {
    return fused_fwd_impl(h, h_row_stride, packed_w, ktab, bias_packed, x, x_row_stride, y, y_row_stride, nullptr,
                          features, logabsdet, accumulate, bin_idx, workspace, workspace_bytes, B, in, features, K, tails,
                          tail_bound, min_bin_width, min_bin_height, min_derivative, inverse, dtype, 0,
                          nf_made_rqs_tiles(features, K, tails), features, 1, (const unsigned*)h_max, (unsigned*)y_max, stream);
}

// nf_made_rqs_fwd on a hidden state that nf_made_hidden_planes_fwd left as tiled fp16 term planes with one scale per
// 64 rows (NF_DTYPE_F16X2 only): no max pass, no split pass, no fp32 hidden state in HBM at all
extern "C" int nf_made_rqs_fwd_planes(const void* planes, const uint32_t* row_scale, const void* packed_w,
                                      const int32_t* ktab, const float* bias_packed, const float* x, int64_t x_row_stride,
                                      float* y, int64_t y_row_stride, uint32_t* y_max, float* logabsdet, int accumulate,
                                      int32_t* bin_idx, float* workspace, size_t workspace_bytes, int64_t B, int in,
                                      int features, int K, int tails, double tail_bound, double min_bin_width,
                                      double min_bin_height, double min_derivative, int inverse, void* stream) {
    if (!planes || !row_scale) return NF_ERR_NULL;
    return fused_fwd_impl(nullptr, 0, packed_w, ktab, bias_packed, x, x_row_stride, y, y_row_stride, nullptr, features,
                          logabsdet, accumulate, bin_idx, workspace, workspace_bytes, B, in, features, K, tails,
                          tail_bound, min_bin_width, min_bin_height, min_derivative, inverse, NF_DTYPE_F16X2, 0,
                          nf_made_rqs_tiles(features, K, tails), features, 1, nullptr, (unsigned*)y_max, stream, nullptr,
                          0, 1.0, planes, (const unsigned*)row_scale);
}

extern "C" int nf_made_rqs_fwd(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                               const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                               int64_t y_row_stride, float* logabsdet, int accumulate, int32_t* bin_idx,
                               float* workspace, size_t workspace_bytes, int64_t B, int in, int features, int K,
                               int tails, double tail_bound, double min_bin_width, double min_bin_height,
                               double min_derivative, int inverse, int dtype, void* stream) {
    return nf_made_rqs_fwd_tiles(h, h_row_stride, packed_w, ktab, bias_packed, x, x_row_stride, y, y_row_stride,
                                 logabsdet, accumulate, bin_idx, workspace, workspace_bytes, B, in, features, K, tails,
                                 tail_bound, min_bin_width, min_bin_height, min_derivative, inverse, dtype, 0,
                                 nf_made_rqs_tiles(features, K, tails), features, 1, stream);
}

// nf_made_rqs_fwd that also KEEPS the conditioner output: params[B, features * P] as nf_masked_linear_fwd would have
// written it (logits unfolded with `logit_div`, the value given to nf_made_rqs_pack_scaled; 1 for nf_made_rqs_pack).
// The training forward: one launch instead of GEMM + spline kernel, and `params` is not read back in between.
// NF_DTYPE_F32 / NF_DTYPE_BF16, forward direction; NF_DTYPE_F16X2 (round 3) on the t4 kernel, which needs the workspace
// of nf_made_rqs_workspace_bytes_dtype (logdet partials + scale slot + term planes; NF_ERR_SCRATCH otherwise).
// Does nf_made_rqs_fwd_params take packs of this dtype for this spline shape?  (f16x2: the t4 kernel's ring of stages
// and its parameter tile must fit LDS.)
extern "C" int nf_made_rqs_fwd_params_supported(int K, int tails, int dtype) {
    if (!fused_supported(K, tails)) return 0;
    if (dtype == NF_DTYPE_F32 || dtype == NF_DTYPE_BF16) return 1;
    if (dtype != NF_DTYPE_F16X2) return 0;
    const int P = tails ? 3 * K - 1 : 3 * K + 1;
    return w8_ring_bytes(P) + w8_xs_bytes(fpt_of(P)) <= 160 * 1024 ? 1 : 0;
}

extern "C" int nf_made_rqs_fwd_params(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                                      const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                                      int64_t y_row_stride, float* logabsdet, int accumulate, float* workspace,
                                      size_t workspace_bytes, int64_t B, int in, int features, int K, int tails,
                                      double tail_bound, double min_bin_width, double min_bin_height,
                                      double min_derivative, int dtype, double logit_div, float* params_out,
                                      int64_t params_row_stride, void* stream) {
    if (!params_out) return NF_ERR_NULL;
    return fused_fwd_impl(h, h_row_stride, packed_w, ktab, bias_packed, x, x_row_stride, y, y_row_stride, nullptr,
                          features, logabsdet, accumulate, nullptr, workspace, workspace_bytes, B, in, features, K, tails,
                          tail_bound, min_bin_width, min_bin_height, min_derivative, 0, dtype, 0,
                          nf_made_rqs_tiles(features, K, tails), features, 1, nullptr, nullptr, stream, params_out,
                          params_row_stride, logit_div);
}

// nf_made_rqs_fwd_params that keeps the conditioner output as bf16 [B, features * P] (row stride in elements): for the
// bf16 training mode, whose backward rounds everything derived from it to bf16 anyway - half the bytes stored here and
// half the bytes the spline backward (nf_rqs_bwd_max_bf16p) reads.  y / logabsdet are computed from the fp32
// accumulators as in nf_made_rqs_fwd_params.
extern "C" int nf_made_rqs_fwd_params_bf16(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                                           const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                                           int64_t y_row_stride, uint32_t* y_max, float* logabsdet, int accumulate,
                                           float* workspace, size_t workspace_bytes, int64_t B, int in, int features,
                                           int K, int tails, double tail_bound, double min_bin_width,
                                           double min_bin_height, double min_derivative, int dtype, double logit_div,
                                           void* params_out_bf16, int64_t params_row_stride, void* stream) {
    if (!params_out_bf16) return NF_ERR_NULL;
    if (((uintptr_t)params_out_bf16 & 15) != 0) return NF_ERR_LAYOUT;
    return fused_fwd_impl(h, h_row_stride, packed_w, ktab, bias_packed, x, x_row_stride, y, y_row_stride, nullptr,
                          features, logabsdet, accumulate, nullptr, workspace, workspace_bytes, B, in, features, K, tails,
                          tail_bound, min_bin_width, min_bin_height, min_derivative, 0, dtype, 0,
                          nf_made_rqs_tiles(features, K, tails), features, 1, nullptr, (unsigned*)y_max, stream,
                          (float*)params_out_bf16, params_row_stride, logit_div, nullptr, nullptr, 1);
}

// ---- split-operand hidden-layer GEMM ------------------------------------------------------------------
extern "C" int nf_linear_split_tiles(int out) { return out <= 0 ? 0 : (out + FN - 1) / FN; }

// terms per operand of the hidden-layer GEMM: the split modes' two / three, or NF_DTYPE_BF16's single bf16 term
static inline int linear_terms(int dtype) { return dtype == NF_DTYPE_BF16 ? 1 : split_terms(dtype); }

extern "C" size_t nf_linear_split_packed_bytes(int out, int in, int dtype) {
    if (out <= 0 || in <= 0 || !linear_terms(dtype)) return 0;
    return (size_t)nf_linear_split_tiles(out) * FN * kpad_of(in) * 2 * linear_terms(dtype);
}

static int linear_split_pack_impl(const float* W, const float* mask, const float* bias, int out, int in, int dtype,
                                  void* packed, int32_t* ktab, float* bias_packed, bool table, void* stream,
                                  int transposed = 0) {
    if (!W || !packed || !ktab || !bias_packed) return NF_ERR_NULL;
    if (out <= 0 || in <= 0) return NF_ERR_SHAPE;
    if (!linear_terms(dtype)) return NF_ERR_DTYPE;
    const int nt = nf_linear_split_tiles(out), kpad = kpad_of(in), terms = linear_terms(dtype);
    const int64_t total = (int64_t)terms * nt * FN * kpad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    unsigned* slot = (unsigned*)(ktab + 2 * nt);
    // the scale pair behind the table: rewritten per pack by the f16x2 flavour only (atomic max into zeros); the other
    // flavours never touch it again after the first pack - no 8-byte memset launch per layer and optimizer step
    if (dtype == NF_DTYPE_F16X2 || table) {
        hipError_t me = hipMemsetAsync(slot, 0, 2 * sizeof(unsigned), s);
        if (me != hipSuccess) return (int)me;
    }
    if (dtype == NF_DTYPE_F16X2) {
        const int64_t nw = (int64_t)out * in;
        const int mgrid = out < 1024 ? out : 1024;
        // (the largest |W * mask| does not depend on the orientation: scanned along the rows as stored)
        hipLaunchKernelGGL(maxabs_kernel, dim3(transposed ? (in < 1024 ? in : 1024) : mgrid), dim3(256), 0, s, W, mask, nw,
                           1, transposed ? out : in, 0, 1.0f, slot);
        hipLaunchKernelGGL(pack_fused_split_kernel<SplitF16x2>, dim3(grid), dim3(256), 0, s, W, mask, bias, out, in, FN,
                           kpad, total, 1, 0, 1.0f, (const unsigned*)slot, (_Float16*)packed, bias_packed, transposed);
    } else if (dtype == NF_DTYPE_BF16) {
        hipLaunchKernelGGL(pack_fused_split_kernel<SplitBf16x1>, dim3(grid), dim3(256), 0, s, W, mask, bias, out, in, FN,
                           kpad, total, 1, 0, 1.0f, (const unsigned*)nullptr, (__bf16*)packed, bias_packed, transposed);
    } else {
        hipLaunchKernelGGL(pack_fused_split_kernel<SplitBf16x3>, dim3(grid), dim3(256), 0, s, W, mask, bias, out, in, FN,
                           kpad, total, 1, 0, 1.0f, (const unsigned*)nullptr, (__bf16*)packed, bias_packed, transposed);
    }
    if (table) launch_ktab_fused(mask, out, in, FN, nt, ktab, transposed, s);
    return nf_launch_status();
}

extern "C" int nf_linear_split_pack(const float* W, const float* mask, const float* bias, int out, int in, int dtype,
                                    void* packed, int32_t* ktab, float* bias_packed, void* stream) {
    return linear_split_pack_impl(W, mask, bias, out, in, dtype, packed, ktab, bias_packed, true, stream);
}

extern "C" int nf_linear_split_repack(const float* W, const float* mask, const float* bias, int out, int in, int dtype,
                                      void* packed, int32_t* ktab, float* bias_packed, void* stream) {
    return linear_split_pack_impl(W, mask, bias, out, in, dtype, packed, ktab, bias_packed, false, stream);
}

// The pack of the TRANSPOSED layer - the weight operand of grad_x = grad_y @ (W * mask), a logical [in, out] matrix -
// read from the layer's own [out, in] arrays (no transposed copies of W and the mask per optimizer step).  Buffer
// sizes: nf_linear_split_packed_bytes(in, out, dtype), 2 * nf_linear_split_tiles(in) + 2 table entries, 256 *
// nf_linear_split_tiles(in) bias floats (zeroed: the input gradient has no bias).  fresh = 0: the mask is unchanged
// since the last call on these buffers, keep the K-range table.
extern "C" int nf_linear_split_pack_t(const float* W, const float* mask, int out, int in, int dtype, void* packed_t,
                                      int32_t* ktab_t, float* bias_zero, int fresh, void* stream) {
    return linear_split_pack_impl(W, mask, nullptr, in, out, dtype, packed_t, ktab_t, bias_zero, fresh != 0, stream, 1);
}

extern "C" int nf_maxabs(const float* x, int64_t x_row_stride, int64_t B, int cols, uint32_t* slot, void* stream) {
    if (B < 0 || cols <= 0) return NF_ERR_SHAPE;
    if (!slot) return NF_ERR_NULL;
    if (B == 0) return (int)hipMemsetAsync(slot, 0, sizeof(uint32_t), (hipStream_t)stream);
    if (!x) return NF_ERR_NULL;
    if (x_row_stride < cols) return NF_ERR_LAYOUT;
    return maxabs_rows(x, x_row_stride, B, cols, (unsigned*)slot, (hipStream_t)stream);
}

template <class SP>
static int launch_linear_split(const FusedArgs& g, int grid, size_t lds, int relu_in, hipStream_t s) {
    static unsigned long long cap_done[2] = {0, 0};
    const int rc = relu_in ? nf_raise_lds_cap((const void*)masked_linear_split_kernel<SP, true>, &cap_done[1])
                           : nf_raise_lds_cap((const void*)masked_linear_split_kernel<SP, false>, &cap_done[0]);
    if (rc != NF_OK) return rc;
    if (relu_in) hipLaunchKernelGGL((masked_linear_split_kernel<SP, true>), dim3(grid), dim3(XT), lds, s, g);
    else hipLaunchKernelGGL((masked_linear_split_kernel<SP, false>), dim3(grid), dim3(XT), lds, s, g);
    return nf_launch_status();
}

static int linear_split_impl(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                             const void* packed_w, const int32_t* ktab, const float* bias_packed,
                             const float* residual, int64_t res_row_stride, float* y, int64_t y_row_stride,
                             uint32_t* y_max, int64_t B, int in, int out, int relu_in, int relu_out,
                             int dtype, int k_reverse, void* stream, const void* x_plane = nullptr,
                             int64_t x_plane_stride = 0, const float* add2 = nullptr, int64_t add2_row_stride = 0) {
    if (B < 0 || in <= 0 || out <= 0) return NF_ERR_SHAPE;
    if (!linear_terms(dtype)) return NF_ERR_DTYPE;
    if (add2 && (relu_out != 2 || add2_row_stride < out)) return NF_ERR_LAYOUT;     // (joins behind a gate only)
    // (a bf16 plane of the A operand: whole 16-wide K steps, 16-byte rows, no ReLU - NF_DTYPE_BF16 only)
    if (x_plane && (dtype != NF_DTYPE_BF16 || relu_in || in % XS != 0 || x_plane_stride < in || x_plane_stride % 8 != 0 ||
                    ((uintptr_t)x_plane & 15) != 0))
        return NF_ERR_LAYOUT;
    hipStream_t s = (hipStream_t)stream;
    if (y_max) {
        hipError_t me = hipMemsetAsync(y_max, 0, sizeof(uint32_t), s);
        if (me != hipSuccess) return (int)me;
    }
    if (B == 0) return NF_OK;
    if (!x || !packed_w || !ktab || !bias_packed || !y || (dtype == NF_DTYPE_F16X2 && !x_max)) return NF_ERR_NULL;
    if (relu_out < 0 || relu_out > 2 || (relu_out == 2 && !residual)) return NF_ERR_NULL;
    if (!x_plane && (x_row_stride < in || in % 8 != 0 || x_row_stride % 4 != 0 || ((uintptr_t)x & 15) != 0)) return NF_ERR_LAYOUT;
    if (y_row_stride < out || (residual && res_row_stride < out)) return NF_ERR_LAYOUT;
    FusedArgs g = {};
    const int nt = nf_linear_split_tiles(out);
    g.h = x; g.ldh = x_row_stride; g.w = packed_w; g.kpad = kpad_of(in); g.ktab = ktab; g.bias = bias_packed;
    g.y = y; g.ldy = y_row_stride; g.M = B; g.K = in; g.N = out; g.n_tiles = nt; g.tn0 = 0;
    g.m_tiles = (B + XM - 1) / XM; g.group_m = 8; g.k_reverse = k_reverse;
    g.wmax = (const unsigned*)(ktab + 2 * nt); g.hmax = (const unsigned*)x_max;
    g.residual = residual; g.ldr = res_row_stride; g.relu_out = relu_out; g.out_max = (unsigned*)y_max;
    g.add2 = add2; g.ldadd2 = add2_row_stride;
    const int64_t grid = g.m_tiles * nt;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    g.vec_out = out % 4 == 0 && y_row_stride % 4 == 0 && ((uintptr_t)y & 15) == 0 &&
                (!residual || (res_row_stride % 4 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                (!add2 || (add2_row_stride % 4 == 0 && ((uintptr_t)add2 & 15) == 0));
    const size_t stages = (size_t)2 * 2 * linear_terms(dtype) * XA, otile = sizeof(float) * (XT / 64) * 32 * XOT;
    const size_t lds = g.vec_out && otile > stages ? otile : stages;
    if (x_plane) {
        g.hp = x_plane; g.hp_ld = x_plane_stride;
        static unsigned long long pre_cap = 0;
        if (const int rc = nf_raise_lds_cap((const void*)masked_linear_split_kernel<SplitBf16x1, false, true>, &pre_cap)) return rc;
        hipLaunchKernelGGL((masked_linear_split_kernel<SplitBf16x1, false, true>), dim3((unsigned)grid), dim3(XT), lds, s, g);
        return nf_launch_status();
    }
    return dtype == NF_DTYPE_BF16X3 ? launch_linear_split<SplitBf16x3>(g, (int)grid, lds, relu_in, s)
           : dtype == NF_DTYPE_BF16 ? launch_linear_split<SplitBf16x1>(g, (int)grid, lds, relu_in, s)
                                    : launch_linear_split<SplitF16x2>(g, (int)grid, lds, relu_in, s);
}

static int wgrad_split_impl(const float* x, int64_t x_row_stride, const uint32_t* x_max, const float* grad_y,
                            int64_t grad_y_row_stride, const uint32_t* grad_y_max, const float* mask,
                            const int32_t* ktab, float* grad_w, float* grad_b, int64_t B, int in, int out, int relu_in,
                            int terms, void* stream, void* workspace = nullptr, size_t workspace_bytes = 0,
                            bool gy_bf16 = false) {
    if (B < 0 || in <= 0 || out <= 0) return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!x || !grad_y || !grad_w || !x_max || !grad_y_max) return NF_ERR_NULL;
    if (x_row_stride < in || grad_y_row_stride < out) return NF_ERR_LAYOUT;
    if (x_row_stride > (1 << 23) || grad_y_row_stride > (1 << 23)) return NF_ERR_LAYOUT;     // 31-bit byte offsets within a chunk
    WgradSplitArgs g;
    g.x = x; g.ldx = x_row_stride; g.gy = grad_y; g.ldgy = grad_y_row_stride; g.x_max = (const unsigned*)x_max;
    g.gy_max = (const unsigned*)grad_y_max; g.mask = mask; g.ktab = ktab; g.gw = grad_w; g.gb = grad_b; g.M = B; g.out = out; g.in = in;
    g.relu_in = relu_in; g.xh = nullptr; g.ldxh = 0; g.xh_plane = 0;
    const int tw = out >= 2048 ? 2 : 1;                                  // wide layers: 256-row tiles, one workgroup per CU
    const int nt = (out + 128 * tw - 1) / (128 * tw), kt = (in + 127) / 128;
    int64_t splits = 3072 / tw / ((int64_t)nt * kt) + 1;
    if (splits < 1) splits = 1;
    const int64_t max_splits = (B + 8 * WSB - 1) / (8 * WSB);
    if (splits > max_splits) splits = max_splits;
    g.splits = (int)splits;
    g.nt = nt; g.kt = kt;
    const int resident = 64 / tw;                                        // workgroups one XCD holds at a time
    g.group = kt * (resident / kt < 1 ? 1 : (resident / kt < nt ? resident / kt : nt));
    const int64_t work = (int64_t)nt * kt * splits;
    const int64_t grid = (work + g.group - 1) / g.group * g.group;      // whole groups, then whole rounds of 8 XCDs
    const int64_t blocks = (grid / g.group + 7) / 8 * 8 * g.group;
    if (blocks > 0x7fffffff) return NF_ERR_SHAPE;
    static unsigned long long cap_done[4] = {0, 0, 0, 0};
    const void* kern = terms == 2 ? (tw == 2 ? (const void*)wgrad_split_kernel<2, SplitF16x2> : (const void*)wgrad_split_kernel<1, SplitF16x2>)
                                  : (tw == 2 ? (const void*)wgrad_split_kernel<2, SplitF16x1> : (const void*)wgrad_split_kernel<1, SplitF16x1>);
    const int rc = nf_raise_lds_cap(kern, &cap_done[(terms == 2 ? 0 : 2) + (tw == 2 ? 1 : 0)]);
    if (rc != NF_OK) return rc;
    const size_t lds = (size_t)2 * terms * (128 * tw + 128) * WSLD * sizeof(_Float16);   // 2 buffers x terms
    const dim3 gr((unsigned)blocks), bl(256 * tw);
    hipStream_t s = (hipStream_t)stream;
    // everything 16-byte addressable: the transposed-read kernel (row-wise 16-byte fetches) - one term always, two terms
    // when x goes through its planes (wide layer + workspace)
    const bool tr_ok = in % 4 == 0 && out % 4 == 0 && in >= 4 && out >= 4 && x_row_stride % 4 == 0 &&
                       grad_y_row_stride % 4 == 0 && (((uintptr_t)x | (uintptr_t)grad_y) & 15) == 0;
    const bool planes_ok = workspace && (terms == 2 || (int64_t)((out + WT_N - 1) / WT_N) * ((in + 255) / 256) > 128) && in % 8 == 0 &&
                           ((uintptr_t)workspace & 15) == 0 &&
                           workspace_bytes >= (size_t)terms * (size_t)B * in * sizeof(_Float16);
    if (gy_bf16 && !(tr_ok && planes_ok && terms == 1 && ((uintptr_t)grad_y & 7) == 0)) return NF_ERR_LAYOUT;
    if (tr_ok && (terms == 1 || planes_ok)) {
        g.nt = (out + WT_N - 1) / WT_N;
        // a wide layer with a caller's workspace: x goes through its fp16 plane (every workgroup of a batch split
        // re-reads the x chunk: half the bytes, no conversion in the loop); a layer of a few tiles reads x too
        // rarely for the extra pass to pay.  (A 512-column tile - grad_y fetched once for this model's wide layers -
        // was tried on top of the plane: 128 accumulator registers + the six unrolled steps' addresses spill 120
        // bytes even with two grad_y sets, 1.41 ms against 1.07.)
        const bool xp = planes_ok;
        const int wk = 256;
        g.kt = (in + wk - 1) / wk;
        // batch splits: whole rounds of 256 workgroups (one per CU).  A wide layer takes about five (measured on
        // 18032 x 512: 2.01 / 1.57 / 1.40 / 1.40 / 1.43 ms at 1 / 2 / 4 / 5 / 8 rounds); a layer of a few tiles takes one:
        // each further round adds a full pass of fp32 atomics over grad_W (512 x 512: 71 / 101 / 130 us at 1 / 2 / 4)
        const int64_t tiles = (int64_t)g.nt * g.kt;
        const int64_t target = (int64_t)g.nt * ((in + 255) / 256) <= 128 ? 256 : 5 * 256;
        int64_t sp = (target + tiles / 2) / tiles;
        if (sp < 1) sp = 1;
        if (sp > max_splits) sp = max_splits;
        g.splits = (int)sp;
        g.group = g.kt * (32 / g.kt < 1 ? 1 : (32 / g.kt < g.nt ? 32 / g.kt : g.nt));
        const int64_t wkn = (int64_t)g.nt * g.kt * sp;
        const int64_t gd = (wkn + g.group - 1) / g.group * g.group;
        const int64_t bk = (gd / g.group + 7) / 8 * 8 * g.group;
        if (bk > 0x7fffffff) return NF_ERR_SHAPE;
        static unsigned long long tr_cap[3] = {0, 0, 0};
        if (xp) {
            g.xh = (const _Float16*)workspace; g.ldxh = in; g.xh_plane = B * in;
            const int64_t quads = B * (in / 4);
            const dim3 pg((unsigned)(quads / 256 < 4096 ? quads / 256 + 1 : 4096));
            if (terms == 2) {
                hipLaunchKernelGGL(half_plane_kernel<2>, pg, dim3(256), 0, s, x, x_row_stride, B, in, g.x_max, relu_in,
                                   (_Float16*)workspace);
                if (const int e = nf_raise_lds_cap((const void*)wgrad_tr_kernel<true, 256, 2>, &tr_cap[2])) return e;
                hipLaunchKernelGGL((wgrad_tr_kernel<true, 256, 2>), dim3((unsigned)bk), dim3(512), (4 + 3 * 4) * WT_IMG, s, g);
                return nf_launch_status();
            }
            hipLaunchKernelGGL(half_plane_kernel<1>, pg, dim3(256), 0, s, x, x_row_stride, B, in, g.x_max, relu_in,
                               (_Float16*)workspace);
            if (gy_bf16) {
                static unsigned long long yb_cap = 0;
                if (const int e = nf_raise_lds_cap((const void*)wgrad_tr_kernel<true, 256, 1, true>, &yb_cap)) return e;
                hipLaunchKernelGGL((wgrad_tr_kernel<true, 256, 1, true>), dim3((unsigned)bk), dim3(512), (2 + 3 * 2) * WT_IMG, s, g);
                return nf_launch_status();
            }
            if (const int e = nf_raise_lds_cap((const void*)wgrad_tr_kernel<true, 256>, &tr_cap[1])) return e;
            hipLaunchKernelGGL((wgrad_tr_kernel<true, 256>), dim3((unsigned)bk), dim3(512), (2 + 3 * 2) * WT_IMG, s, g);
            return nf_launch_status();
        }
        if (const int e = nf_raise_lds_cap((const void*)wgrad_tr_kernel<false, 256>, &tr_cap[0])) return e;
        hipLaunchKernelGGL((wgrad_tr_kernel<false, 256>), dim3((unsigned)bk), dim3(512), 2 * (1 + 2) * WT_IMG, s, g);
        return nf_launch_status();
    }
    if (terms == 2 && tw == 2) hipLaunchKernelGGL((wgrad_split_kernel<2, SplitF16x2>), gr, bl, lds, s, g);
    else if (terms == 2) hipLaunchKernelGGL((wgrad_split_kernel<1, SplitF16x2>), gr, bl, lds, s, g);
    else if (tw == 2) hipLaunchKernelGGL((wgrad_split_kernel<2, SplitF16x1>), gr, bl, lds, s, g);
    else hipLaunchKernelGGL((wgrad_split_kernel<1, SplitF16x1>), gr, bl, lds, s, g);
    return nf_launch_status();
}

extern "C" int nf_masked_linear_split_fwd(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                          const void* packed_w, const int32_t* ktab, const float* bias_packed,
                                          const float* residual, int64_t res_row_stride, float* y, int64_t y_row_stride,
                                          uint32_t* y_max, int64_t B, int in, int out, int relu_in, int relu_out,
                                          int dtype, void* stream) {
    return linear_split_impl(x, x_row_stride, x_max, packed_w, ktab, bias_packed, residual, res_row_stride, y,
                             y_row_stride, y_max, B, in, out, relu_in, relu_out, dtype, 0, stream);
}

// Input-gradient GEMM on split operands: nf_masked_linear_split_fwd on the TRANSPOSED pack (x = grad_y, `gate` = the
// forward input whose sign gates the result, NULL for none), with each tile's live K range walked from its END down:
// under MADE's transposed masks the column tiles' ranges share their end, so the column tiles of one row tile -
// scheduled back to back on one XCD - pass the same grad_y columns together and all but one find them in L2.
extern "C" int nf_masked_linear_split_bwd_input(const float* grad_y, int64_t grad_y_row_stride, const uint32_t* grad_y_max,
                                                const void* packed_wt, const int32_t* ktab_t, const float* bias_zero,
                                                const float* gate, int64_t gate_row_stride, float* grad_x,
                                                int64_t grad_x_row_stride, uint32_t* grad_x_max, int64_t B, int out, int in,
                                                int dtype, void* stream) {
    return linear_split_impl(grad_y, grad_y_row_stride, grad_y_max, packed_wt, ktab_t, bias_zero, gate, gate_row_stride,
                             grad_x, grad_x_row_stride, grad_x_max, B, out, in, 0, gate ? 2 : 0, dtype, 1, stream);
}

// nf_masked_linear_split_bwd_input whose result joins a second gradient: grad_x = gate(grad_y @ (W * mask)) + add
// (`add` [B, in]: the gradient that reaches the same tensor along a skip connection - a residual block's input).  One
// launch instead of the GEMM, an add pass and a max pass over the sum (grad_x_max is that of the sum).
extern "C" int nf_masked_linear_split_bwd_input_add(const float* grad_y, int64_t grad_y_row_stride,
                                                    const uint32_t* grad_y_max, const void* packed_wt,
                                                    const int32_t* ktab_t, const float* bias_zero, const float* gate,
                                                    int64_t gate_row_stride, const float* add, int64_t add_row_stride,
                                                    float* grad_x, int64_t grad_x_row_stride, uint32_t* grad_x_max,
                                                    int64_t B, int out, int in, int dtype, void* stream) {
    if (!gate || !add) return NF_ERR_NULL;
    return linear_split_impl(grad_y, grad_y_row_stride, grad_y_max, packed_wt, ktab_t, bias_zero, gate, gate_row_stride,
                             grad_x, grad_x_row_stride, grad_x_max, B, out, in, 0, 2, dtype, 1, stream, nullptr, 0, add,
                             add_row_stride);
}

extern "C" int nf_masked_linear_split_bwd_weight(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                                 const float* grad_y, int64_t grad_y_row_stride,
                                                 const uint32_t* grad_y_max, const float* mask, const int32_t* ktab,
                                                 float* grad_w, float* grad_b, int64_t B, int in, int out, int relu_in,
                                                 void* stream) {
    return wgrad_split_impl(x, x_row_stride, x_max, grad_y, grad_y_row_stride, grad_y_max, mask, ktab, grad_w, grad_b, B,
                            in, out, relu_in, 2, stream);
}

// nf_masked_linear_split_bwd_input (NF_DTYPE_BF16) with grad_y given as a bf16 matrix (nf_rqs_bwd_max_bf16's output):
// the A operand goes from memory to LDS as it is - half the bytes, no conversion.  out % 16 == 0, 16-byte rows.
extern "C" int nf_masked_linear_split_bwd_input_bf16(const void* grad_y_bf16, int64_t grad_y_row_stride,
                                                     const void* packed_wt, const int32_t* ktab_t, const float* bias_zero,
                                                     const float* gate, int64_t gate_row_stride, float* grad_x,
                                                     int64_t grad_x_row_stride, uint32_t* grad_x_max, int64_t B, int out,
                                                     int in, void* stream) {
    if (!grad_y_bf16) return NF_ERR_NULL;
    return linear_split_impl((const float*)grad_y_bf16, grad_y_row_stride, nullptr, packed_wt, ktab_t, bias_zero, gate,
                             gate_row_stride, grad_x, grad_x_row_stride, grad_x_max, B, out, in, 0, gate ? 2 : 0,
                             NF_DTYPE_BF16, 1, stream, grad_y_bf16, grad_y_row_stride);
}

// nf_masked_linear_half_bwd_weight with grad_y given as a bf16 matrix (nf_rqs_bwd_max_bf16's output; row stride in
// elements, 8-byte aligned rows): wide layers only (the workspace path), else NF_ERR_LAYOUT.
extern "C" int nf_masked_linear_half_bwd_weight_bf16(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                                     const void* grad_y_bf16, int64_t grad_y_row_stride,
                                                     const uint32_t* grad_y_max, const float* mask, const int32_t* ktab,
                                                     float* grad_w, float* grad_b, int64_t B, int in, int out,
                                                     int relu_in, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_split_impl(x, x_row_stride, x_max, (const float*)grad_y_bf16, grad_y_row_stride, grad_y_max, mask, ktab,
                            grad_w, grad_b, B, in, out, relu_in, 1, stream, workspace, workspace_bytes, true);
}

// The same with a caller-owned workspace (2 * B * in * 2 bytes, 16-byte aligned): the transposed-read kernel with act(x)
// pre-split into two fp16 planes fetched by LDS-DMA.
extern "C" int nf_masked_linear_split_bwd_weight_ws(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                                    const float* grad_y, int64_t grad_y_row_stride,
                                                    const uint32_t* grad_y_max, const float* mask, const int32_t* ktab,
                                                    float* grad_w, float* grad_b, int64_t B, int in, int out,
                                                    int relu_in, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_split_impl(x, x_row_stride, x_max, grad_y, grad_y_row_stride, grad_y_max, mask, ktab, grad_w, grad_b, B,
                            in, out, relu_in, 2, stream, workspace, workspace_bytes);
}

// The same weight gradient with ONE scaled fp16 term per operand (11 significant bits relative to each tensor's
// largest element, fp32 accumulation): the bf16 training mode's weight gradient - a third of the MFMAs.
extern "C" int nf_masked_linear_half_bwd_weight(const float* x, int64_t x_row_stride, const uint32_t* x_max,
                                                const float* grad_y, int64_t grad_y_row_stride,
                                                const uint32_t* grad_y_max, const float* mask, const int32_t* ktab,
                                                float* grad_w, float* grad_b, int64_t B, int in, int out, int relu_in,
                                                void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_split_impl(x, x_row_stride, x_max, grad_y, grad_y_row_stride, grad_y_max, mask, ktab, grad_w, grad_b, B,
                            in, out, relu_in, 1, stream, workspace, workspace_bytes);
}

// ---- the one entry point that allocates nothing and takes no workspace ---------------------------------
// nf_made_rqs_fwd for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that
// forbid hipMalloc): the per-tile logdet partials (and, for the split-operand dtypes, the hidden state's scale
// slot and - when they fit - its term planes) live in a file-scope device array instead of a caller workspace.
// Every GPU the library runs on holds its own copy of the array at its own address until the process ends; the
// address is resolved once per device with that device current (hipGetSymbolAddress) and only that pointer is
// ever handed to kernels - the array's name in host code is a host-side stand-in, not a device address.
#define NF_NOALLOC_SCRATCH_BYTES (8u << 20)
static __device__ __align__(256) unsigned char nf_noalloc_scratch[NF_NOALLOC_SCRATCH_BYTES];
#define NF_NOALLOC_MAX_DEVICES 64
static void* nf_noalloc_ptr[NF_NOALLOC_MAX_DEVICES];       // per device ordinal, NULL until resolved

static int nf_noalloc_scratch_of_current_device(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= NF_NOALLOC_MAX_DEVICES) return NF_ERR_SHAPE;
    void* p = __atomic_load_n(&nf_noalloc_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(nf_noalloc_scratch));       // this device's copy of the array
        if (e != hipSuccess) return (int)e;
        if (!p) return NF_ERR_NULL;
        __atomic_store_n(&nf_noalloc_ptr[dev], p, __ATOMIC_RELEASE);       // (racing first calls resolve the same address)
    }
    *out = p;
    return NF_OK;
}

extern "C" int nf_made_rqs_fwd_noalloc(const float* h, int64_t h_row_stride, const void* packed_w, const int32_t* ktab,
                                       const float* bias_packed, const float* x, int64_t x_row_stride, float* y,
                                       int64_t y_row_stride, float* logabsdet, int accumulate, int32_t* bin_idx,
                                       int64_t B, int in, int features, int K, int tails, double tail_bound,
                                       double min_bin_width, double min_bin_height, double min_derivative, int inverse,
                                       int dtype, void* stream) {
    void* scratch = nullptr;
    const int rc = nf_noalloc_scratch_of_current_device(&scratch);          // also on B == 0: a warm-up call resolves it
    if (rc != NF_OK) return rc;
    if (B < 0 || in <= 0 || features <= 0 || !fused_supported(K, tails)) return NF_ERR_SHAPE;
    // the partials (+ the split dtypes' 256-byte scale slot) must fit the static array: NF_ERR_SCRATCH otherwise
    size_t need = nf_made_rqs_workspace_bytes(B, features, K, tails);
    if (split_terms(dtype)) need = (need + 255) / 256 * 256 + 256;
    if (need > NF_NOALLOC_SCRATCH_BYTES) return NF_ERR_SCRATCH;
    return fused_fwd_impl(h, h_row_stride, packed_w, ktab, bias_packed, x, x_row_stride, y, y_row_stride, nullptr,
                          features, logabsdet, accumulate, bin_idx, (float*)scratch, NF_NOALLOC_SCRATCH_BYTES, B, in,
                          features, K, tails, tail_bound, min_bin_width, min_bin_height, min_derivative, inverse, dtype, 0,
                          nf_made_rqs_tiles(features, K, tails), features, 1, nullptr, nullptr, stream);
}
