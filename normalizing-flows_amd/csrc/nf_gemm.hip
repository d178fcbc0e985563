// MADE masked-linear layers on the gfx950 matrix cores.
//   y[B,out] = act_out( act_in(x * in_scale + in_shift)[B,in] @ (W*mask)^T + bias (+ residual) )
// Reference: models/made.py:150-152 (F.linear(x, W*mask, b) with the mask re-multiplied on every
// call), :196-207 and :268-284 (the ReLU / BatchNorm / skip around it).
//
// Here the mask is folded into the weights ONCE (nf_made_pack): the packed matrix is zero-padded
// to 128-row x 32-column tiles, and a per-row-tile table records which 32-wide K tiles hold any
// unmasked weight, so the GEMM never loads or multiplies a fully masked tile (the autoregressive
// masks are triangular: SURVEY.md 2.3).
//
// f32 path: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).  Block tile 128x128x32,
// 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator registers).  Operands are
// staged global -> registers -> LDS (rows padded to 36 floats: conflict-free ds_read_b128), the next
// K tile's global loads are issued before the current tile's 64 MFMAs.  A lane reads 4 consecutive
// k per ds_read_b128; within an 8-wide K chunk lanes 0-31 own k 0..3 and lanes 32-63 own k 4..7, for
// A and B alike, which is a reordering of the K sum the instruction is indifferent to.
#include "nf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_LD = BK + 4;          // padded row, floats
constexpr int GT = 256;                 // threads
constexpr int GROUP_M = 8;              // row tiles that share a weight tile back to back (L2 reuse)

struct GemmArgs {
    const float* x;
    int64_t ldx;
    const float* w;          // packed [n_tiles*128][kpad]
    int kpad;
    const int32_t* ktab;     // [n_tiles][2] = first / one-past-last K tile with unmasked weights
    const float* bias;       // nullable
    const float* res;        // nullable
    int64_t ldr;
    const float* in_scale;   // nullable (eval-mode BatchNorm folded to scale/shift on the input)
    const float* in_shift;
    float* y;
    int64_t ldy;
    const float* gate;       // nullable [M, N]: outputs are zeroed where gate <= 0 (ReLU backward)
    int64_t ldg;
    int64_t M;
    int N, K;
    int relu_in, relu_out;
    int n_tiles;             // column tiles computed by this launch ...
    int tn0;                 // ... starting at this one (tile window; 0 / all for a full layer)
    int64_t m_tiles;
    int vec_a;               // x rows 16-byte aligned and K % 4 == 0
    int vec_out;             // y / residual / gate / bias take 16-byte accesses (LDS-staged epilogue)
    int k_reverse;           // walk the live K range from its END down (see nf_masked_linear_bwd_input)
};

// 16-byte epilogue of the 128 x 128 block kernels: each wave drops 16 rows x its 64 columns of accumulators into LDS
// (row stride 72 floats: conflict-free for the MFMA C/D layout), reads them back four columns per lane, adds bias /
// residual, applies ReLU / gate and stores float4.  The 4-byte form issued 64 residual loads per wave, each waiting
// in front of its store; here the four 16-byte loads of a chunk are in flight together.  `ot`: 4 x 16 x 72 floats.
constexpr int OT_LD = 72;
__device__ __forceinline__ void staged_epilogue(const GemmArgs& g, float* ot, const f32x16 (&acc)[2][2], int64_t m0,
                                                int n0, int wr, int wc, int lane) {
    __syncthreads();                            // every wave is done with the operand tiles this aliases
    ot += (wr * 2 + wc) * 16 * OT_LD;
    const int c0 = n0 + wc * 64 + (lane & 15) * 4;
    const bool cok = c0 < g.N;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias && cok) bv = *reinterpret_cast<const float4*>(g.bias + c0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    ot[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * OT_LD + j * 32 + (lane & 31)] = acc[i][j][8 * c + r];
            float4 rv[4], gv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row = m0 + wr * 64 + i * 32 + c * 16 + q * 4 + (lane >> 4);
                rv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                gv[q] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (row < g.M && cok) {
                    if (g.res) rv[q] = *reinterpret_cast<const float4*>(g.res + row * g.ldr + c0);
                    if (g.gate) gv[q] = *reinterpret_cast<const float4*>(g.gate + row * g.ldg + c0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row = m0 + wr * 64 + i * 32 + c * 16 + q * 4 + (lane >> 4);
                float4 v = *reinterpret_cast<const float4*>(ot + (q * 4 + (lane >> 4)) * OT_LD + (lane & 15) * 4);
                v.x = v.x + bv.x + rv[q].x; v.y = v.y + bv.y + rv[q].y; v.z = v.z + bv.z + rv[q].z; v.w = v.w + bv.w + rv[q].w;
                if (g.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (g.gate) {
                    v.x = gv[q].x > 0.f ? v.x : 0.f; v.y = gv[q].y > 0.f ? v.y : 0.f;
                    v.z = gv[q].z > 0.f ? v.z : 0.f; v.w = gv[q].w > 0.f ? v.w : 0.f;
                }
                if (row < g.M && cok) *reinterpret_cast<float4*>(g.y + row * g.ldy + c0) = v;
            }
        }
}

__device__ __forceinline__ void tile_of_block(const GemmArgs& g, int64_t& tm, int& tn) {
    // XCD-aware order: blocks that share an XCD (id % 8) get one contiguous run of work items; inside
    // the run GROUP_M row tiles sweep a weight tile together, so the 4 MiB L2 of that XCD sees each
    // operand panel a handful of times instead of once per block.
    const int64_t nwg = (int64_t)gridDim.x;
    const int64_t bid = blockIdx.x;
    const int64_t q = nwg / 8, r = nwg % 8;
    const int64_t xcd = bid % 8;
    const int64_t wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    const int64_t per_group = (int64_t)GROUP_M * g.n_tiles;
    const int64_t grp = wg / per_group;
    const int64_t first_m = grp * GROUP_M;
    const int64_t gm = min((int64_t)GROUP_M, g.m_tiles - first_m);
    const int64_t in = wg - grp * per_group;
    tm = first_m + in % gm;
    // heaviest column tiles first: with triangular masks the K extent grows with the tile index, and
    // starting the long workgroups early leaves the short ones to fill the tail of the launch
    tn = g.n_tiles - 1 - (int)(in / gm) + g.tn0;
}

// load_one only issues the loads; everything that consumes a loaded value lives in store_one so that
// no s_waitcnt lands between the prefetch and the MFMAs it is meant to hide under.  The staged
// registers are eight named float4 variables: kept in arrays, hipcc (ROCm 7.2) leaves them in scratch.
template <bool VEC_A>
__device__ __forceinline__ void load_one(const GemmArgs& g, const float* arow, const float* brow, int k, int kt,
                                         float4& a, float4& b) {
    if (VEC_A) {
        a = *reinterpret_cast<const float4*>(arow + min(k, g.K - 4));
    } else {
        a.x = arow[min(k, g.K - 1)];
        a.y = arow[min(k + 1, g.K - 1)];
        a.z = arow[min(k + 2, g.K - 1)];
        a.w = arow[min(k + 3, g.K - 1)];
    }
    b = *reinterpret_cast<const float4*>(brow + kt * BK);
}

template <bool IN_AFFINE, bool RELU_IN>
__device__ __forceinline__ void store_one(const GemmArgs& g, float* As, float* Bs, int k, float4 sc, float4 sh,
                                          float4 a, float4 b) {
    if (IN_AFFINE) {
        a.x = a.x * sc.x + sh.x;
        a.y = a.y * sc.y + sh.y;
        a.z = a.z * sc.z + sh.z;
        a.w = a.w * sc.w + sh.w;
    }
    if (RELU_IN) {
        a.x = fmaxf(a.x, 0.f);
        a.y = fmaxf(a.y, 0.f);
        a.z = fmaxf(a.z, 0.f);
        a.w = fmaxf(a.w, 0.f);
    }
    // zero the K tail (the packed weights are zero there too, but 0 * inf/NaN must not appear)
    a.x = k < g.K ? a.x : 0.f;
    a.y = k + 1 < g.K ? a.y : 0.f;
    a.z = k + 2 < g.K ? a.z : 0.f;
    a.w = k + 3 < g.K ? a.w : 0.f;
    *reinterpret_cast<float4*>(As) = a;
    *reinterpret_cast<float4*>(Bs) = b;
}

// VEC_A: x rows are 16-byte aligned and K % 4 == 0 (one 16-byte load per staged float4).
// IN_AFFINE: per-input-feature scale/shift (eval-mode BatchNorm) applied while staging.
// Loads are unconditional on clamped addresses and masked with selects afterwards: a branch around
// a load makes hipcc drain vmcnt(0) after each one, which serialises the whole prefetch.
template <bool VEC_A, bool IN_AFFINE, bool RELU_IN>
__global__ __launch_bounds__(GT, 2) void masked_linear_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

    int64_t tm;
    int tn;
    tile_of_block(g, tm, tn);
    const int64_t m0 = tm * BM;
    const int n0 = tn * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int kt_begin = g.ktab[2 * tn], kt_end = g.ktab[2 * tn + 1];

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging map: float4 index v = tid + 256*i -> row v>>3 (= srow + 32*i), k-offset 4*(v&7)
    const int srow = tid >> 3, skc = (tid & 7) * 4;
    // rows past M re-read the last row; they are never stored
    const float* ar0 = g.x + min(m0 + srow, g.M - 1) * g.ldx;
    const float* ar1 = g.x + min(m0 + srow + 32, g.M - 1) * g.ldx;
    const float* ar2 = g.x + min(m0 + srow + 64, g.M - 1) * g.ldx;
    const float* ar3 = g.x + min(m0 + srow + 96, g.M - 1) * g.ldx;
    const float* br0 = g.w + (int64_t)(n0 + srow) * g.kpad + skc;
    const float* br1 = br0 + (int64_t)32 * g.kpad;
    const float* br2 = br0 + (int64_t)64 * g.kpad;
    const float* br3 = br0 + (int64_t)96 * g.kpad;
    float* as0 = &As[srow * LDS_LD + skc];
    float* bs0 = &Bs[srow * LDS_LD + skc];
    float4 a0, a1, a2, a3, b0, b1, b2, b3;

#define NF_LOAD_TILE(KT)                                                  \
    do {                                                                  \
        const int kt_ = (KT);                                             \
        const int k_ = kt_ * BK + skc;                                    \
        load_one<VEC_A>(g, ar0, br0, k_, kt_, a0, b0);                    \
        load_one<VEC_A>(g, ar1, br1, k_, kt_, a1, b1);                    \
        load_one<VEC_A>(g, ar2, br2, k_, kt_, a2, b2);                    \
        load_one<VEC_A>(g, ar3, br3, k_, kt_, a3, b3);                    \
    } while (0)

    const int kn = kt_end - kt_begin, kdir = g.k_reverse ? -1 : 1, kfirst = g.k_reverse ? kt_end - 1 : kt_begin;
    if (kn > 0) NF_LOAD_TILE(kfirst);
    for (int ki = 0; ki < kn; ++ki) {
        const int kt = kfirst + kdir * ki;
        __syncthreads();             // everyone done reading the previous tile
        {
            const int k = kt * BK + skc;
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (IN_AFFINE) {
                const int k0 = min(k, g.K - 1), k1 = min(k + 1, g.K - 1), k2 = min(k + 2, g.K - 1), k3 = min(k + 3, g.K - 1);
                sc = make_float4(g.in_scale[k0], g.in_scale[k1], g.in_scale[k2], g.in_scale[k3]);
                sh = make_float4(g.in_shift[k0], g.in_shift[k1], g.in_shift[k2], g.in_shift[k3]);
            }
            store_one<IN_AFFINE, RELU_IN>(g, as0, bs0, k, sc, sh, a0, b0);
            store_one<IN_AFFINE, RELU_IN>(g, as0 + 32 * LDS_LD, bs0 + 32 * LDS_LD, k, sc, sh, a1, b1);
            store_one<IN_AFFINE, RELU_IN>(g, as0 + 64 * LDS_LD, bs0 + 64 * LDS_LD, k, sc, sh, a2, b2);
            store_one<IN_AFFINE, RELU_IN>(g, as0 + 96 * LDS_LD, bs0 + 96 * LDS_LD, k, sc, sh, a3, b3);
        }
        __syncthreads();
        NF_LOAD_TILE(ki + 1 < kn ? kt + kdir : kt);   // in flight under the MFMAs below (the last one is a re-read)
        __builtin_amdgcn_sched_barrier(0);       // keep the prefetch issue ABOVE the MFMAs

        const int frow = lane & 31, fk = (lane >> 5) * 4;
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4*>(&As[(wr * 64 + i * 32 + frow) * LDS_LD + kc * 8 + fk]);
                b[i] = *reinterpret_cast<const float4*>(&Bs[(wc * 64 + i * 32 + frow) * LDS_LD + kc * 8 + fk]);
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

#undef NF_LOAD_TILE

    if (g.vec_out) {
        staged_epilogue(g, As, acc, m0, n0, wr, wc, lane);
        return;
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.res) v += g.res[row * g.ldr + col];
                if (g.relu_out) v = fmaxf(v, 0.0f);
                if (g.gate) v = g.gate[row * g.ldg + col] > 0.0f ? v : 0.0f;
                g.y[row * g.ldy + col] = v;
            }
        }
}

// ---- bf16 matrix-core path (2e-2 contract) ------------------------------------------------------
// Same block structure with v_mfma_f32_32x32x16_bf16: x is read as fp32 and rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) while it is staged, the packed weights are bf16, accumulation and the epilogue
// stay fp32.  K tile = 64; LDS rows are 64 bf16 + 8 pad = 144 bytes, the same conflict-free stride as
// the fp32 kernel.  Lane l of a fragment holds A[row l&31][k = 8(l>>5) .. +7] (cdna guide, sec. 3).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BK16 = 64;
constexpr int LDS_LD16 = BK16 + 8;      // bf16 elements per padded row

template <bool VEC_A>
__device__ __forceinline__ void load_one16(const GemmArgs& g, const float* arow, const __bf16* brow, int k, int kt,
                                           float4& a_lo, float4& a_hi, bf16x8& b) {
    if (VEC_A) {
        const int kk = min(k, g.K - 8);
        a_lo = *reinterpret_cast<const float4*>(arow + kk);
        a_hi = *reinterpret_cast<const float4*>(arow + kk + 4);
    } else {
        a_lo.x = arow[min(k, g.K - 1)];
        a_lo.y = arow[min(k + 1, g.K - 1)];
        a_lo.z = arow[min(k + 2, g.K - 1)];
        a_lo.w = arow[min(k + 3, g.K - 1)];
        a_hi.x = arow[min(k + 4, g.K - 1)];
        a_hi.y = arow[min(k + 5, g.K - 1)];
        a_hi.z = arow[min(k + 6, g.K - 1)];
        a_hi.w = arow[min(k + 7, g.K - 1)];
    }
    b = *reinterpret_cast<const bf16x8*>(brow + kt * BK16);
}

template <bool IN_AFFINE, bool RELU_IN>
__device__ __forceinline__ void store_one16(const GemmArgs& g, __bf16* As, __bf16* Bs, int k, const float (&sc)[8],
                                            const float (&sh)[8], float4 a_lo, float4 a_hi, bf16x8 b) {
    float v[8] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_hi.x, a_hi.y, a_hi.z, a_hi.w};
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = v[j];
        if (IN_AFFINE) t = t * sc[j] + sh[j];
        if (RELU_IN) t = fmaxf(t, 0.f);
        t = k + j < g.K ? t : 0.f;
        o[j] = (__bf16)t;
    }
    *reinterpret_cast<bf16x8*>(As) = o;
    *reinterpret_cast<bf16x8*>(Bs) = b;
}

template <bool VEC_A, bool IN_AFFINE, bool RELU_IN>
__global__ __launch_bounds__(GT, 2) void masked_linear_bf16_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) __bf16 As[BM * LDS_LD16];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[BN * LDS_LD16];

    int64_t tm;
    int tn;
    tile_of_block(g, tm, tn);
    const int64_t m0 = tm * BM;
    const int n0 = tn * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ktab counts 32-wide K tiles; this kernel walks 64-wide ones
    const int kt_begin = g.ktab[2 * tn] >> 1, kt_end = (g.ktab[2 * tn + 1] + 1) >> 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging map: 16-byte (8 x bf16) chunk v = tid + 256*i -> row v>>3 (= srow + 32*i), k-offset 8*(v&7)
    const int srow = tid >> 3, skc = (tid & 7) * 8;
    const float* ar0 = g.x + min(m0 + srow, g.M - 1) * g.ldx;
    const float* ar1 = g.x + min(m0 + srow + 32, g.M - 1) * g.ldx;
    const float* ar2 = g.x + min(m0 + srow + 64, g.M - 1) * g.ldx;
    const float* ar3 = g.x + min(m0 + srow + 96, g.M - 1) * g.ldx;
    const __bf16* br0 = reinterpret_cast<const __bf16*>(g.w) + (int64_t)(n0 + srow) * g.kpad + skc;
    const __bf16* br1 = br0 + (int64_t)32 * g.kpad;
    const __bf16* br2 = br0 + (int64_t)64 * g.kpad;
    const __bf16* br3 = br0 + (int64_t)96 * g.kpad;
    __bf16* as0 = &As[srow * LDS_LD16 + skc];
    __bf16* bs0 = &Bs[srow * LDS_LD16 + skc];
    float4 a0l, a0h, a1l, a1h, a2l, a2h, a3l, a3h;
    bf16x8 b0, b1, b2, b3;

#define NF_LOAD_TILE16(KT)                                                \
    do {                                                                  \
        const int kt_ = (KT);                                             \
        const int k_ = kt_ * BK16 + skc;                                  \
        load_one16<VEC_A>(g, ar0, br0, k_, kt_, a0l, a0h, b0);            \
        load_one16<VEC_A>(g, ar1, br1, k_, kt_, a1l, a1h, b1);            \
        load_one16<VEC_A>(g, ar2, br2, k_, kt_, a2l, a2h, b2);            \
        load_one16<VEC_A>(g, ar3, br3, k_, kt_, a3l, a3h, b3);            \
    } while (0)

    const int kn = kt_end - kt_begin, kdir = g.k_reverse ? -1 : 1, kfirst = g.k_reverse ? kt_end - 1 : kt_begin;
    if (kn > 0) NF_LOAD_TILE16(kfirst);
    for (int ki = 0; ki < kn; ++ki) {
        const int kt = kfirst + kdir * ki;
        __syncthreads();
        {
            const int k = kt * BK16 + skc;
            float sc[8], sh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc[j] = IN_AFFINE ? g.in_scale[min(k + j, g.K - 1)] : 1.f;
                sh[j] = IN_AFFINE ? g.in_shift[min(k + j, g.K - 1)] : 0.f;
            }
            store_one16<IN_AFFINE, RELU_IN>(g, as0, bs0, k, sc, sh, a0l, a0h, b0);
            store_one16<IN_AFFINE, RELU_IN>(g, as0 + 32 * LDS_LD16, bs0 + 32 * LDS_LD16, k, sc, sh, a1l, a1h, b1);
            store_one16<IN_AFFINE, RELU_IN>(g, as0 + 64 * LDS_LD16, bs0 + 64 * LDS_LD16, k, sc, sh, a2l, a2h, b2);
            store_one16<IN_AFFINE, RELU_IN>(g, as0 + 96 * LDS_LD16, bs0 + 96 * LDS_LD16, k, sc, sh, a3l, a3h, b3);
        }
        __syncthreads();
        NF_LOAD_TILE16(ki + 1 < kn ? kt + kdir : kt);
        __builtin_amdgcn_sched_barrier(0);

        const int frow = lane & 31, fk = (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(&As[(wr * 64 + i * 32 + frow) * LDS_LD16 + ks * 16 + fk]);
                b[i] = *reinterpret_cast<const bf16x8*>(&Bs[(wc * 64 + i * 32 + frow) * LDS_LD16 + ks * 16 + fk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#undef NF_LOAD_TILE16

    if (g.vec_out) {
        staged_epilogue(g, reinterpret_cast<float*>(As), acc, m0, n0, wr, wc, lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
            const float bv = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (g.res) v += g.res[row * g.ldr + col];
                if (g.relu_out) v = fmaxf(v, 0.0f);
                if (g.gate) v = g.gate[row * g.ldg + col] > 0.0f ? v : 0.0f;
                g.y[row * g.ldy + col] = v;
            }
        }
}

// ---- weight gradient ---------------------------------------------------------------------------------
// grad_W[n,k] += mask[n,k] * sum_b grad_y[b,n] * act(x[b,k])   (autograd of made.py:152 `W*mask`: the mask
// re-applied to the gradient).  A "TN" GEMM: both operands are row-major with the reduction index b as the
// slow one, so tiles of 32 batch rows are staged [b][n] / [b][k] (rows padded to 132 floats) and an MFMA lane
// reads A[n = lane&31][b = lane>>5] with one conflict-free ds_read_b32.  128 x 128 outputs per block, 4 waves
// as 2x2; the batch is split over gridDim.z and partial tiles are added with fp32 atomics when split.
// Output tiles with no unmasked weight (the forward K-tile table) exit immediately.
constexpr int WB = 32;                 // batch rows per staged tile
constexpr int WLD = 132;               // padded LDS row (floats)

struct WgradArgs {
    const float* x;
    int64_t ldx;
    const float* gy;
    int64_t ldgy;
    const float* mask;       // nullable [out, in]
    const int32_t* ktab;     // nullable: forward K-tile table, [out/128][2] in 32-wide units
    float* gw;               // [out, in], accumulated into
    int64_t M;               // batch rows
    int out, in;
    int relu_in;
    int splits;
};

template <bool VEC>
__device__ __forceinline__ float4 ld4_rows(const float* base, int64_t ld, int64_t row, int64_t M, int col, int ncols) {
    // rows past M and columns past ncols contribute zeros (clamped address, select afterwards)
    const int64_t r = min(row, M - 1);
    if (!VEC) {
        // ragged layers (widths not a multiple of 4, rows not 16-byte aligned): four scalar loads, each bounded
        const float* p = base + r * ld;
        float4 v;
        v.x = p[min(col, ncols - 1)];
        v.y = p[min(col + 1, ncols - 1)];
        v.z = p[min(col + 2, ncols - 1)];
        v.w = p[min(col + 3, ncols - 1)];
        const bool in = row < M;
        v.x = (in && col < ncols) ? v.x : 0.f;
        v.y = (in && col + 1 < ncols) ? v.y : 0.f;
        v.z = (in && col + 2 < ncols) ? v.z : 0.f;
        v.w = (in && col + 3 < ncols) ? v.w : 0.f;
        return v;
    }
    const int c = min(col, ncols - 4);
    float4 v = *reinterpret_cast<const float4*>(base + r * ld + c);
    const bool ok = row < M && col < ncols;
    v.x = ok ? v.x : 0.f;
    v.y = ok ? v.y : 0.f;
    v.z = ok ? v.z : 0.f;
    v.w = ok ? v.w : 0.f;
    return v;
}

template <bool VEC>
__global__ __launch_bounds__(GT, 2) void wgrad_f32_kernel(const WgradArgs g) {
    __shared__ __attribute__((aligned(16))) float Ys[WB * WLD];
    __shared__ __attribute__((aligned(16))) float Xs[WB * WLD];
    const int tn = blockIdx.x, tk = blockIdx.y;
    const int n0 = tn * 128, k0 = tk * 128;
    if (g.ktab) {
        const int kb = g.ktab[2 * tn] * 32, ke = g.ktab[2 * tn + 1] * 32;
        if (k0 >= ke || k0 + 128 <= kb) return;            // fully masked tile
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t per = ((g.M + g.splits - 1) / g.splits + WB - 1) / WB * WB;
    const int64_t b_begin = (int64_t)blockIdx.z * per, b_end = min(g.M, b_begin + per);
    if (b_begin >= b_end) return;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging: float4 v = tid + 256*i -> row v>>5 (= srow + 8*i), col 4*(v&31)
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    float4 y0, y1, y2, y3, x0, x1, x2, x3;
#define NF_WG_LOAD(B0)                                                              \
    do {                                                                            \
        const int64_t b_ = (B0) + srow;                                             \
        y0 = ld4_rows<VEC>(g.gy, g.ldgy, b_, b_end, n0 + scol, g.out);                   \
        y1 = ld4_rows<VEC>(g.gy, g.ldgy, b_ + 8, b_end, n0 + scol, g.out);               \
        y2 = ld4_rows<VEC>(g.gy, g.ldgy, b_ + 16, b_end, n0 + scol, g.out);              \
        y3 = ld4_rows<VEC>(g.gy, g.ldgy, b_ + 24, b_end, n0 + scol, g.out);              \
        x0 = ld4_rows<VEC>(g.x, g.ldx, b_, b_end, k0 + scol, g.in);                      \
        x1 = ld4_rows<VEC>(g.x, g.ldx, b_ + 8, b_end, k0 + scol, g.in);                  \
        x2 = ld4_rows<VEC>(g.x, g.ldx, b_ + 16, b_end, k0 + scol, g.in);                 \
        x3 = ld4_rows<VEC>(g.x, g.ldx, b_ + 24, b_end, k0 + scol, g.in);                 \
    } while (0)
    auto relu4 = [&](float4 v) {
        if (g.relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        return v;
    };

    NF_WG_LOAD(b_begin);
    for (int64_t b0 = b_begin; b0 < b_end; b0 += WB) {
        __syncthreads();
        *reinterpret_cast<float4*>(&Ys[srow * WLD + scol]) = y0;
        *reinterpret_cast<float4*>(&Ys[(srow + 8) * WLD + scol]) = y1;
        *reinterpret_cast<float4*>(&Ys[(srow + 16) * WLD + scol]) = y2;
        *reinterpret_cast<float4*>(&Ys[(srow + 24) * WLD + scol]) = y3;
        *reinterpret_cast<float4*>(&Xs[srow * WLD + scol]) = relu4(x0);
        *reinterpret_cast<float4*>(&Xs[(srow + 8) * WLD + scol]) = relu4(x1);
        *reinterpret_cast<float4*>(&Xs[(srow + 16) * WLD + scol]) = relu4(x2);
        *reinterpret_cast<float4*>(&Xs[(srow + 24) * WLD + scol]) = relu4(x3);
        __syncthreads();
        NF_WG_LOAD(min(b0 + WB, b_end - 1));        // prefetch (the last one re-reads, its rows are masked to zero)
        __builtin_amdgcn_sched_barrier(0);
        const int fr = lane & 31, fb = lane >> 5;
#pragma unroll
        for (int bb = 0; bb < WB; bb += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = Ys[(bb + fb) * WLD + wr * 64 + i * 32 + fr];
                b[i] = Xs[(bb + fb) * WLD + wc * 64 + i * 32 + fr];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#undef NF_WG_LOAD

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
                float v = acc[i][j][r];
                if (g.mask) v *= g.mask[idx];
                if (g.splits > 1) atomicAdd(&g.gw[idx], v);
                else g.gw[idx] += v;
            }
        }
}

// grad_b[n] += sum_b grad_y[b,n]: 64 columns x 4 row lanes per block, rows split over gridDim.y, fp32 atomics.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ gy, int64_t ldgy, int64_t M, int out,
                                                     float* __restrict__ gb) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int64_t per = (M + gridDim.y - 1) / gridDim.y;
    const int64_t b0 = (int64_t)blockIdx.y * per, b1 = min(M, b0 + per);
    float s = 0.f;
    if (n < out)
        for (int64_t b = b0 + rl; b < b1; b += 4) s += gy[b * ldgy + n];
    part[rl][c] = s;
    __syncthreads();
    if (rl == 0 && n < out) atomicAdd(&gb[n], part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

// ---- packing ---------------------------------------------------------------------------------
// Logical matrix element (n, k) lives at W[n*sn + k*sk]: (sn, sk) = (in, 1) packs W itself, (1, out_src)
// packs its transpose (the operand of the input-gradient GEMM).
__global__ void pack_f32_kernel(const float* __restrict__ W, const float* __restrict__ mask, int out, int in,
                                int64_t sn, int64_t sk, int kpad, int64_t total, float* __restrict__ packed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / kpad;
        const int k = (int)(i - n * kpad);
        float v = 0.0f;
        if (n < out && k < in) {
            v = W[n * sn + k * sk];
            if (mask) v *= mask[n * sn + k * sk];
        }
        packed[i] = v;
    }
}

__global__ void pack_bf16_kernel(const float* __restrict__ W, const float* __restrict__ mask, int out, int in,
                                 int64_t sn, int64_t sk, int kpad, int64_t total, __bf16* __restrict__ packed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / kpad;
        const int k = (int)(i - n * kpad);
        float v = 0.0f;
        if (n < out && k < in) {
            v = W[n * sn + k * sk];
            if (mask) v *= mask[n * sn + k * sk];
        }
        packed[i] = (__bf16)v;
    }
}

// K-tile table: [tile][first, one-past-last 32-wide K tile with an unmasked weight].  A function of the mask only.
// KTAB_SLICES workgroups per tile scan it (consecutive threads on consecutive addresses when the fast index has unit
// stride), raw column bounds meet in the table through atomics, a last launch converts them to tiles.
constexpr int KTAB_SLICES = 64;
__global__ void ktab_reset_kernel(int32_t* __restrict__ ktab, int nt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt) { ktab[2 * i] = 0x7fffffff; ktab[2 * i + 1] = -1; }
}

__global__ __launch_bounds__(256) void ktab_kernel(const float* __restrict__ mask, int out, int in, int64_t sn,
                                                   int64_t sk, int32_t* __restrict__ ktab) {
    const int tn = blockIdx.x, slice = blockIdx.y, t = threadIdx.x;
    const int rows = min(BN, out - tn * BN);
    int lo = 0x7fffffff, hi = -1;
    if (!mask) {
        if (rows > 0 && slice == 0) { lo = 0; hi = in - 1; }
    } else if (sk == 1) {                          // rows of the logical matrix are contiguous: threads along k
        for (int r = slice; r < rows; r += KTAB_SLICES) {
            const float* __restrict__ m = mask + (int64_t)(tn * BN + r) * sn;
#pragma unroll 4
            for (int k = t; k < in; k += 256)
                if (m[k] != 0.0f) { lo = min(lo, k); hi = max(hi, k); }
        }
    } else {                                       // transposed view: threads along the tile's rows
        for (int k = slice; k < in; k += KTAB_SLICES)
            for (int r = t; r < rows; r += 256)
                if (mask[(int64_t)(tn * BN + r) * sn + k * sk] != 0.0f) { lo = min(lo, k); hi = max(hi, k); }
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

__global__ void ktab_tiles_kernel(int32_t* __restrict__ ktab, int nt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const int lo = ktab[2 * i], hi = ktab[2 * i + 1];
    if (hi < 0) { ktab[2 * i] = 0; ktab[2 * i + 1] = 0; }
    else { ktab[2 * i] = lo / BK; ktab[2 * i + 1] = hi / BK + 1; }
}

inline int n_tiles_of(int out) { return (out + BN - 1) / BN; }
inline int kpad_of(int in) { return (in + BK16 - 1) / BK16 * BK16; }   // whole K tiles of either kernel

}  // namespace

extern "C" size_t nf_made_packed_bytes(int out, int in, int dtype) {
    if (out <= 0 || in <= 0) return 0;
    const size_t elems = (size_t)n_tiles_of(out) * BN * kpad_of(in);
    return dtype == NF_DTYPE_BF16 ? elems * 2 : elems * 4;
}

extern "C" int nf_made_ktab_len(int out) { return out <= 0 ? 0 : 2 * n_tiles_of(out); }

static int pack_impl(const float* W, const float* mask, int out, int in, int64_t sn, int64_t sk, int dtype,
                     void* packed, int32_t* ktab, void* stream) {
    if (!W || !packed) return NF_ERR_NULL;       // ktab == NULL: the mask is unchanged, keep the caller's table
    if (out <= 0 || in <= 0) return NF_ERR_SHAPE;
    if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_BF16) return NF_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    const int kpad = kpad_of(in);
    const int64_t total = (int64_t)n_tiles_of(out) * BN * kpad;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == NF_DTYPE_F32)
        hipLaunchKernelGGL(pack_f32_kernel, dim3(grid), dim3(256), 0, s, W, mask, out, in, sn, sk, kpad, total,
                           (float*)packed);
    else
        hipLaunchKernelGGL(pack_bf16_kernel, dim3(grid), dim3(256), 0, s, W, mask, out, in, sn, sk, kpad, total,
                           (__bf16*)packed);
    if (ktab) {
        const int nt = n_tiles_of(out);
        hipLaunchKernelGGL(ktab_reset_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, ktab, nt);
        hipLaunchKernelGGL(ktab_kernel, dim3(nt, KTAB_SLICES), dim3(256), 0, s, mask, out, in, sn, sk, ktab);
        hipLaunchKernelGGL(ktab_tiles_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, ktab, nt);
    }
    return nf_launch_status();
}

extern "C" int nf_made_pack(const float* W, const float* mask, int out, int in, int dtype, void* packed,
                            int32_t* ktab, void* stream) {
    return pack_impl(W, mask, out, in, in, 1, dtype, packed, ktab, stream);
}

// Packs the TRANSPOSE of (W*mask): a logical [in, out] matrix, the weight operand of grad_x = grad_y @ (W*mask).
// Buffer sizes: nf_made_packed_bytes(in, out, dtype) and nf_made_ktab_len(in).
extern "C" int nf_made_pack_t(const float* W, const float* mask, int out, int in, int dtype, void* packed_t,
                              int32_t* ktab_t, void* stream) {
    return pack_impl(W, mask, in, out, 1, in, dtype, packed_t, ktab_t, stream);
}

static int linear_impl(const float* x, int64_t x_row_stride, const void* packed_w, const int32_t* ktab,
                       const float* bias, const float* residual, int64_t residual_row_stride, const float* in_scale,
                       const float* in_shift, const float* gate, int64_t gate_row_stride, float* y,
                       int64_t y_row_stride, int64_t B, int in, int out, int relu_in, int relu_out, int dtype,
                       int n_tile_begin, int n_tile_count, void* stream, int k_reverse = 0) {
    if (B < 0 || in <= 0 || out <= 0) return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!x || !packed_w || !ktab || !y) return NF_ERR_NULL;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return NF_ERR_NULL;
    if (x_row_stride < in || y_row_stride < out || (residual && residual_row_stride < out)) return NF_ERR_LAYOUT;
    if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_BF16) return NF_ERR_DTYPE;
    GemmArgs g;
    g.x = x; g.ldx = x_row_stride; g.w = (const float*)packed_w; g.kpad = kpad_of(in); g.ktab = ktab;
    g.bias = bias; g.res = residual; g.ldr = residual_row_stride; g.in_scale = in_scale; g.in_shift = in_shift;
    g.y = y; g.ldy = y_row_stride; g.M = B; g.N = out; g.K = in; g.relu_in = relu_in; g.relu_out = relu_out;
    g.gate = gate; g.ldg = gate_row_stride;
    if (n_tile_begin < 0 || n_tile_count < 0 || n_tile_begin + n_tile_count > n_tiles_of(out)) return NF_ERR_SHAPE;
    if (n_tile_count == 0) return NF_OK;
    g.n_tiles = n_tile_count;
    g.tn0 = n_tile_begin;
    g.m_tiles = (B + BM - 1) / BM;
    const int vec = dtype == NF_DTYPE_BF16 ? 8 : 4;      // floats per staged chunk
    g.vec_a = (in % vec == 0) && (x_row_stride % 4 == 0) && (((uintptr_t)x & 15) == 0);
    auto a16 = [](const void* p, int64_t ld) { return p == nullptr || (ld % 4 == 0 && ((uintptr_t)p & 15) == 0); };
    g.k_reverse = k_reverse;
    g.vec_out = out % 4 == 0 && a16(y, y_row_stride) && a16(bias, 0) && a16(residual, residual_row_stride) &&
                a16(gate, gate_row_stride);
    const int64_t grid = g.m_tiles * g.n_tiles;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 gr((int)grid), bl(GT);
    const int sel = (g.vec_a ? 4 : 0) | (in_scale ? 2 : 0) | (relu_in ? 1 : 0);
    if (dtype == NF_DTYPE_BF16) {
        switch (sel) {
            case 0: hipLaunchKernelGGL((masked_linear_bf16_kernel<false, false, false>), gr, bl, 0, s, g); break;
            case 1: hipLaunchKernelGGL((masked_linear_bf16_kernel<false, false, true>), gr, bl, 0, s, g); break;
            case 2: hipLaunchKernelGGL((masked_linear_bf16_kernel<false, true, false>), gr, bl, 0, s, g); break;
            case 3: hipLaunchKernelGGL((masked_linear_bf16_kernel<false, true, true>), gr, bl, 0, s, g); break;
            case 4: hipLaunchKernelGGL((masked_linear_bf16_kernel<true, false, false>), gr, bl, 0, s, g); break;
            case 5: hipLaunchKernelGGL((masked_linear_bf16_kernel<true, false, true>), gr, bl, 0, s, g); break;
            case 6: hipLaunchKernelGGL((masked_linear_bf16_kernel<true, true, false>), gr, bl, 0, s, g); break;
            default: hipLaunchKernelGGL((masked_linear_bf16_kernel<true, true, true>), gr, bl, 0, s, g); break;
        }
        return nf_launch_status();
    }
    switch (sel) {
        case 0: hipLaunchKernelGGL((masked_linear_f32_kernel<false, false, false>), gr, bl, 0, s, g); break;
        case 1: hipLaunchKernelGGL((masked_linear_f32_kernel<false, false, true>), gr, bl, 0, s, g); break;
        case 2: hipLaunchKernelGGL((masked_linear_f32_kernel<false, true, false>), gr, bl, 0, s, g); break;
        case 3: hipLaunchKernelGGL((masked_linear_f32_kernel<false, true, true>), gr, bl, 0, s, g); break;
        case 4: hipLaunchKernelGGL((masked_linear_f32_kernel<true, false, false>), gr, bl, 0, s, g); break;
        case 5: hipLaunchKernelGGL((masked_linear_f32_kernel<true, false, true>), gr, bl, 0, s, g); break;
        case 6: hipLaunchKernelGGL((masked_linear_f32_kernel<true, true, false>), gr, bl, 0, s, g); break;
        default: hipLaunchKernelGGL((masked_linear_f32_kernel<true, true, true>), gr, bl, 0, s, g); break;
    }
    return nf_launch_status();
}

extern "C" int nf_masked_linear_fwd_tiles(const float* x, int64_t x_row_stride, const void* packed_w,
                                          const int32_t* ktab, const float* bias, const float* residual,
                                          int64_t residual_row_stride, const float* in_scale, const float* in_shift,
                                          float* y, int64_t y_row_stride, int64_t B, int in, int out, int relu_in,
                                          int relu_out, int dtype, int n_tile_begin, int n_tile_count, void* stream) {
    return linear_impl(x, x_row_stride, packed_w, ktab, bias, residual, residual_row_stride, in_scale, in_shift, nullptr,
                       0, y, y_row_stride, B, in, out, relu_in, relu_out, dtype, n_tile_begin, n_tile_count, stream);
}

// grad_x[B,in] = (grad_y[B,out] @ (W*mask)) zeroed where gate <= 0 (gate = the forward input when the layer
// applied ReLU to it, else NULL): the forward kernel on the transposed pack (nf_made_pack_t).
extern "C" int nf_masked_linear_bwd_input(const float* grad_y, int64_t grad_y_row_stride, const void* packed_wt,
                                          const int32_t* ktab_t, const float* gate, int64_t gate_row_stride,
                                          float* grad_x, int64_t grad_x_row_stride, int64_t B, int in, int out,
                                          int dtype, void* stream) {
    if (gate && gate_row_stride < in) return NF_ERR_LAYOUT;
    return linear_impl(grad_y, grad_y_row_stride, packed_wt, ktab_t, nullptr, nullptr, 0, nullptr, nullptr, gate,
                       gate_row_stride, grad_x, grad_x_row_stride, B, out, in, 0, 0, dtype, 0,
                       in > 0 ? n_tiles_of(in) : 0, stream, 1);
    // (k_reverse = 1: under MADE's transposed masks the live K ranges of the column tiles share their END, not their
    //  start; walked downwards, the column tiles of one row tile - scheduled back to back on one XCD - pass the same
    //  grad_y columns at the same time and all but one of them find them in L2.  The final layer's 2.4 GB of
    //  grad_params were otherwise fetched once per column tile.)
}

extern "C" int nf_masked_linear_fwd(const float* x, int64_t x_row_stride, const void* packed_w, const int32_t* ktab,
                                    const float* bias, const float* residual, int64_t residual_row_stride,
                                    const float* in_scale, const float* in_shift, float* y, int64_t y_row_stride,
                                    int64_t B, int in, int out, int relu_in, int relu_out, int dtype, void* stream) {
    return nf_masked_linear_fwd_tiles(x, x_row_stride, packed_w, ktab, bias, residual, residual_row_stride, in_scale,
                                      in_shift, y, y_row_stride, B, in, out, relu_in, relu_out, dtype, 0,
                                      out > 0 ? n_tiles_of(out) : 0, stream);
}

// grad_w[out,in] += mask * (grad_y^T @ act(x)), grad_b[out] += column sums of grad_y (either may be NULL).
// Both are ACCUMULATED (zero them before the first call); with the batch split across blocks the sums
// use fp32 atomics, so the last bits depend on arrival order.  `ktab` (nullable) is the forward K-tile table
// of this layer: output tiles without unmasked weights are skipped.  Any widths / alignment (ragged shapes take
// scalar loads).
extern "C" int nf_masked_linear_bwd_weight(const float* x, int64_t x_row_stride, const float* grad_y,
                                           int64_t grad_y_row_stride, const float* mask, const int32_t* ktab,
                                           float* grad_w, float* grad_b, int64_t B, int in, int out, int relu_in,
                                           int dtype, void* stream) {
    if (B < 0 || in <= 0 || out <= 0) return NF_ERR_SHAPE;
    if (dtype != NF_DTYPE_F32 && dtype != NF_DTYPE_BF16) return NF_ERR_DTYPE;
    if (B == 0) return NF_OK;
    if (!x || !grad_y || (!grad_w && !grad_b)) return NF_ERR_NULL;
    if (x_row_stride < in || grad_y_row_stride < out) return NF_ERR_LAYOUT;
    // 16-byte loads when both operands allow them, else the bounded scalar-load flavour (D = 2, K = 4 ...)
    const bool vec = !(in % 4 || out % 4 || x_row_stride % 4 || grad_y_row_stride % 4 ||
                       (((uintptr_t)x | (uintptr_t)grad_y) & 15));
    hipStream_t s = (hipStream_t)stream;
    if (grad_w) {
        WgradArgs g;
        g.x = x; g.ldx = x_row_stride; g.gy = grad_y; g.ldgy = grad_y_row_stride; g.mask = mask; g.ktab = ktab;
        g.gw = grad_w; g.M = B; g.out = out; g.in = in; g.relu_in = relu_in;
        const int nt = n_tiles_of(out), kt = (in + 127) / 128;
        int64_t splits = 1536 / ((int64_t)nt * kt);
        if (splits < 1) splits = 1;
        const int64_t max_splits = (B + 4 * WB - 1) / (4 * WB);
        if (splits > max_splits) splits = max_splits;
        if (splits > 65535) splits = 65535;
        g.splits = (int)splits;
        if (vec) hipLaunchKernelGGL(wgrad_f32_kernel<true>, dim3(nt, kt, (int)splits), dim3(GT), 0, s, g);
        else hipLaunchKernelGGL(wgrad_f32_kernel<false>, dim3(nt, kt, (int)splits), dim3(GT), 0, s, g);
    }
    if (grad_b) {
        int64_t rs = (B + 1023) / 1024;
        if (rs > 256) rs = 256;
        hipLaunchKernelGGL(colsum_kernel, dim3((out + 63) / 64, (int)rs), dim3(256), 0, s, grad_y, grad_y_row_stride, B,
                           out, grad_b);
    }
    return nf_launch_status();
}
