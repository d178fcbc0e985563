// One RQ-spline coupling layer in ONE launch (gfx950).
//
//   identity half -> ResidualNet (initial layer, residual blocks, final Linear)      (models/nn/resnet.py:90-100)
//   transformed half -> RQ spline with those parameters, widths / heights / sqrt(hidden)   (transforms.py:212-247)
//   outputs = both halves, logabsdet = sum over the transformed features                   (transforms.py:80-106)
//
// As two launches (nf_resnet_hidden_fwd + nf_coupling_rqs_fwd) the layer cost 1.24 ms per 2^20 rows at cfg-2 (D = 64,
// hidden 32, K = 8): the fused final-layer kernel is built for a deep K loop and evaluates the spline 32 samples x 11
// features at a time on 512 lanes (69 % of them busy), three workgroups per row tile each re-reading the hidden state.
// Here a 256-thread workgroup owns 64 whole rows: the rows are read once (coalesced) into LDS, the hidden stack runs
// there (fp32 MFMA, weights from L1/L2), then the final Linear is taken in chunks of 8 features (184 parameter columns,
// 12 MFMA tiles = 3 per wave) into a column-major LDS tile from which every lane evaluates TWO splines of its own
// sample as one basic block (lane = sample: conflict-free parameter reads, all 256 lanes busy); the outputs replace the
// inputs in the LDS rows, which leave as whole rows - identity columns included, no clone, no hidden state in HBM.
// 79 KB of LDS: two workgroups per CU, one's spline evaluation (VALU) under the other's GEMM chunks (matrix pipe).
#include <stdlib.h>
#include "nf_common.h"
#include "nf_rqs_eval.h"
#include "nf_resnet_layer.h"

namespace {

constexpr int CM = 64;                  // samples per workgroup
constexpr int CLDT = CM + 4;            // column stride of the parameter tile (floats): 16-byte aligned quads
constexpr int CMAXW = 128;              // widest conditioner layer / most identity features
constexpr int CMAXROW = 512;            // widest row (whole rows live in LDS)

struct CouplingArgs {
    const float* x;
    int64_t ldx;
    float* y;
    int64_t ldy;
    int row_width;
    const int32_t* in_index;     // [n_in] identity columns (conditioner inputs)
    int n_in, kp;
    const int32_t* feat_index;   // [F] transformed columns
    int F;
    const float* blob;           // hidden stack, layout of nf_resnet_blob_floats
    int hidden, num_blocks;
    const void* wf;              // final layer as three bf16 term planes in MFMA operand order, per chunk (see coupling_pack_kernel)
    const float* bf;             // [F * P], logit entries folded like the weights
    float* logabsdet;
    int accumulate;
    int32_t* bins;               // nullable [M, F]
    int64_t M;
    int fc, nct;                 // features and 32-column MFMA tiles per chunk
    int ko;                      // timing experiments (NF_COUPLING_KO): 1 no spline evaluation, 2 no final MFMAs, 4 no hidden stack
    RqsConst c;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float4& lo, const float4& hi, bf16x8 (&t)[3]) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 a = (__bf16)v[j];
        const float r1 = v[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        t[0][j] = a;
        t[1][j] = b;
        t[2][j] = (__bf16)(r1 - (float)b);
    }
}

// NKS = hidden / 16 K steps (2 or 4).  The final Linear runs on bf16 MFMA with every fp32 operand carried as three
// bf16 terms (8 + 8 + 8 bits: the six products kept are the fp32 product to 2^-22) - 192 matrix cycles per 16-wide K
// step of a 32 x 32 tile against 512 on the fp32 pipe, which at K = 32 was as much matrix time as the splines took
// vector time.  A workgroup's 64 samples need only 2 x NKS x 3 operand fragments per lane (a wave keeps ONE row tile):
// they stay in registers from the hidden stack to the last chunk.
template <int KT, bool INV, int NKS>
__global__ __launch_bounds__(RT, NKS <= 2 ? 3 : 2) void coupling_rqs_layer_kernel(const CouplingArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int rw1 = g.row_width | 1;                       // odd row stride: a column read by 64 lanes hits 64 banks
    const int wmax = max(g.kp, g.hidden);
    const int ld = wmax + 4;
    float* XY = smem;                                      // [64][rw1] whole rows: inputs, transformed columns overwritten
    float* H = XY + ((CM * rw1 + 3) & ~3);                 // [64][ld] hidden state (dead once the fragments are taken)
    float* tile = H;                                       // [32 nct][CLDT] parameter tile, over H and T
    float* T = H + CM * ld;                                // [64][ld]
    float* ldsum = H + max(32 * g.nct * CLDT, 2 * CM * ld);   // [4][64]
    int* idx = (int*)(ldsum + 4 * CM);                     // [n_in] identity columns, then [F] transformed columns
    float* bnl = (float*)(idx + ((g.n_in + g.F + 3) & ~3));   // [2 blocks][2][hidden] folded BatchNorm
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * CM;
    const int P = g.c.P;
    const int frow = lane & 31;
    const int hd = g.hidden;
    const int irow = wave & 1;                             // this wave's row tile in every chunk
    const int nchunks = (g.F + g.fc - 1) / g.fc;

    // B fragments of a chunk: packed [chunk][column tile][K step][term][lane][8 bf16]; requested a phase ahead.
    // Column tiles of a chunk go round the wave pairs: (0,1) take tile 0, (2,3) tile 1, (0,1) tile 2, ...
    bf16x8 wfr[2][NKS][3];
    float bfr[2];
    auto request = [&](int chunk) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = (wave >> 1) + 2 * u;
            if (j < g.nct) {
                const bf16x8* wp = (const bf16x8*)g.wf + ((int64_t)(chunk * g.nct + j) * NKS * 3) * 64 + lane;
                bfr[u] = g.bf[min(chunk * g.fc * P + j * 32 + frow, g.F * P - 1)];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int q = 0; q < 3; ++q) wfr[u][ks][q] = wp[(ks * 3 + q) * 64];
            }
        }
    };

    // whole rows in, coalesced (rows past the batch are zero), four rows of a wave in flight at once; the column
    // indices to LDS beside them.  (Gathering straight from global memory - index load, then the dependent element
    // load, per element - left a workgroup waiting out sixteen memory round trips before its first MFMA.)
    for (int c = tid; c < g.n_in + g.F; c += RT) idx[c] = c < g.n_in ? g.in_index[c] : g.feat_index[c - g.n_in];
    const int ntile = 2 * (hd / 32);
    const bool mine = wave < ntile;
    const int li = wave & 1, lj = wave >> 1;               // this wave's (row, column) tile of every layer
    const int fk = (lane >> 5) * 4;
    constexpr int NKQ = NKS * 2;                           // float4 per fragment over K = hidden (K <= 16 NKS)
    float4 wq[2][NKQ];
    float bq[2];
    auto fetch = [&](int slot, const float* W, int K, const float* bias) {
        if (!mine) return;
        const int col = lj * 32 + frow;
        bq[slot] = bias[col];
#pragma unroll
        for (int q = 0; q < NKQ; ++q) {
            if (q * 8 < K) {
                wq[slot][q] = *reinterpret_cast<const float4*>(W + (int64_t)col * K + fk + q * 8);
            }
        }
    };
    auto run = [&](int slot, const float* src, int K, const float* bn, float* dst, bool residual) {   // bn: LDS [2][hidden] scale, shift
        if (!mine) return;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bq[slot];
        const float* ap = src + (li * 32 + frow) * ld + fk;
#pragma unroll
        for (int q = 0; q < NKQ; ++q) {
            if (q * 8 < K) {
                float4 a = *reinterpret_cast<const float4*>(ap + q * 8);
                const float4 b = wq[slot][q];
                if (bn) {
                    const float4 s_ = *reinterpret_cast<const float4*>(bn + fk + q * 8);
                    const float4 h_ = *reinterpret_cast<const float4*>(bn + hd + fk + q * 8);
                    a.x = fmaxf(fmaf(a.x, s_.x, h_.x), 0.f);
                    a.y = fmaxf(fmaf(a.y, s_.y, h_.y), 0.f);
                    a.z = fmaxf(fmaf(a.z, s_.z, h_.z), 0.f);
                    a.w = fmaxf(fmaf(a.w, s_.w, h_.w), 0.f);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = li * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float* d = dst + row * ld + lj * 32 + frow;
            *d = residual ? *d + acc[r] : acc[r];
        }
    };
    const bool init_fast = g.kp <= 16 * NKS;               // (the fragment registers cover K <= hidden)
    if (init_fast) fetch(0, g.blob, g.kp, g.blob + (int64_t)g.hidden * g.kp);
    // folded BatchNorm scale / shift of every block layer to LDS: [2 blocks][scale, shift][hidden]
    for (int e = tid; e < g.num_blocks * 4 * hd; e += RT) {
        const int l = e / (2 * hd), o = e - l * 2 * hd;    // layer l of the blocks: its (s, t) pair sits in front of its weights
        const int64_t per = 2 * hd + (int64_t)hd * hd + hd;
        bnl[e] = g.blob[(int64_t)hd * g.kp + hd + l * per + o];
    }
    for (int c = lane; c < g.row_width; c += 64) {         // (all sixteen rows of the wave in flight at once)
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = g.x[min(m0 + wave * 16 + q, g.M - 1) * g.ldx + c];
#pragma unroll
        for (int q = 0; q < 16; ++q) XY[(wave * 16 + q) * rw1 + c] = m0 + wave * 16 + q < g.M ? v[q] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < CM * g.kp; e += RT) {
        const int r = e / g.kp, c = e - r * g.kp;
        T[r * ld + c] = c < g.n_in ? XY[r * rw1 + idx[c]] : 0.f;
    }
    __syncthreads();

    // hidden stack (nf_resnet_hidden_fwd's arithmetic).  A layer's weight fragments, bias and folded-BatchNorm
    // scale / shift are requested one layer ahead, before the barrier that publishes its input: as loads inside the
    // layer every one of the 1 + 2 blocks layers began with an exposed L2 round trip.
    // Output tile t of a layer (2 row tiles x hidden / 32 column tiles) belongs to wave t % 4; hidden 32 / 64 means at
    // most one tile per wave.
    const float* p = g.blob + (int64_t)hd * g.kp + hd;     // first block
    const int nb = (g.ko & 4) ? 0 : g.num_blocks;
    if (nb > 0) fetch(1, p + 2 * hd, hd, p + 2 * hd + (int64_t)hd * hd);
    if (init_fast) run(0, T, g.kp, nullptr, H, false);
    else layer<false, false>(T, ld, g.kp, g.blob, g.blob + (int64_t)hd * g.kp, nullptr, nullptr, H, hd, wave, lane);
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const float *Wa = p + 2 * hd, *ba = Wa + (int64_t)hd * hd;
        const float *Wb = ba + 3 * hd, *bb = Wb + (int64_t)hd * hd;
        fetch(0, Wb, hd, bb);
        run(1, H, hd, bnl + (2 * b) * 2 * hd, T, false);
        __syncthreads();
        p = bb + hd;
        if (b + 1 < nb) fetch(1, p + 2 * hd, hd, p + 2 * hd + (int64_t)hd * hd);
        run(0, T, hd, bnl + (2 * b + 1) * 2 * hd, H, true);
        __syncthreads();
    }
    request(0);
    // this lane's A fragments: row irow * 32 + lane % 32, k = 16 ks + 8 (lane / 32) .. + 7, as three bf16 terms
    bf16x8 afr[NKS][3];
    {
        const float* ap = H + (irow * 32 + frow) * ld + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            split3(*reinterpret_cast<const float4*>(ap + 16 * ks), *reinterpret_cast<const float4*>(ap + 16 * ks + 4), afr[ks]);
    }
    __syncthreads();                                       // H is dead: the parameter tile may overwrite it

    // final Linear + spline, `fc` features at a time
    float lsum = 0.f;
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int f0 = chunk * g.fc;
        // (the tile is free: the barrier behind the previous chunk's evaluation, or the one above)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = (wave >> 1) + 2 * u;
            if (j < g.nct && !(g.ko & 2)) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bfr[u];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][2], wfr[u][ks][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], wfr[u][ks][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][1], wfr[u][ks][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][1], wfr[u][ks][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], wfr[u][ks][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks][0], wfr[u][ks][0], acc, 0, 0, 0);
                }
                float* tp = tile + (j * 32 + frow) * CLDT + irow * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(tp + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            }
        }
        if (chunk + 1 < nchunks) request(chunk + 1);              // lands under the evaluation below
        __syncthreads();
        // lane = sample; wave w takes features w, w + 4, ... of the chunk, two at a time as one basic block
        const int nf = min(g.fc, g.F - f0);
        for (int fi = wave; fi < ((g.ko & 1) ? 0 : nf); fi += 2 * (RT / 64)) {
            const int fj = fi + RT / 64;
            const int ca = idx[g.n_in + f0 + fi];
            float ya, la;
            int ba_;
            if (fj < nf) {
                const int cb = idx[g.n_in + f0 + fj];
                float yb, lb;
                int bb_;
                rqs_eval<KT, INV, CLDT, true, true, const float*, false>(tile + fi * P * CLDT + lane, XY[lane * rw1 + ca], g.c, ya, la, ba_);
                rqs_eval<KT, INV, CLDT, true, true, const float*, false>(tile + fj * P * CLDT + lane, XY[lane * rw1 + cb], g.c, yb, lb, bb_);
                XY[lane * rw1 + cb] = yb;
                lsum += la;
                lsum += lb;
                if (g.bins && m0 + lane < g.M) g.bins[(m0 + lane) * (int64_t)g.F + f0 + fj] = bb_;
            } else {
                rqs_eval<KT, INV, CLDT, true, true, const float*, false>(tile + fi * P * CLDT + lane, XY[lane * rw1 + ca], g.c, ya, la, ba_);
                lsum += la;
            }
            XY[lane * rw1 + ca] = ya;
            if (g.bins && m0 + lane < g.M) g.bins[(m0 + lane) * (int64_t)g.F + f0 + fi] = ba_;
        }
        __syncthreads();
    }

    // logabsdet: the four waves' sums of a sample in fixed order (deterministic)
    ldsum[wave * CM + lane] = lsum;
    __syncthreads();
    if (tid < CM && m0 + tid < g.M) {
        const float s = ((ldsum[tid] + ldsum[CM + tid]) + ldsum[2 * CM + tid]) + ldsum[3 * CM + tid];
        g.logabsdet[m0 + tid] = g.accumulate ? g.logabsdet[m0 + tid] + s : s;
    }
    // whole rows out: identity columns as they came, transformed columns from the spline
    for (int r = wave; r < CM; r += RT / 64) {
        if (m0 + r >= g.M) break;
        float* yr = g.y + (m0 + r) * g.ldy;
        for (int c = lane; c < g.row_width; c += 64) yr[c] = XY[r * rw1 + c];
    }
}

// final layer -> [chunk][column tile][K step][term][lane][8 bf16] (lane l: column c0 + 32 j + l % 32 of the chunk that
// starts at parameter column c0 = chunk * fc * P, k = 16 ks + 8 (l / 32) .. + 7), the logit rows (widths, heights) folded
// with log2(e) / logit_div first: the spline's softmax then runs on bare exp2 instructions and needs no pre-divisor
__global__ void coupling_pack_kernel(const float* __restrict__ W, const float* __restrict__ b, int out, int in, int P,
                                     int K2, float fold, int fc, int nct, int nchunks, __bf16* __restrict__ Wp,
                                     float* __restrict__ bp) {
    const int nks = in / 16;
    const int64_t total = (int64_t)nchunks * nct * nks * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int ks = (int)(r % nks); r /= nks;
        const int j = (int)(r % nct);
        const int chunk = (int)(r / nct);
        const int col = chunk * fc * P + j * 32 + (lane & 31);
        const int k0 = 16 * ks + 8 * (lane >> 5);
        bf16x8 t[3];
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[q] = col < out ? W[(int64_t)col * in + k0 + q] : 0.f;
            if (col < out && (col % P) < K2) v[q] *= fold;
        }
        split3(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), t);
        bf16x8* o = (bf16x8*)Wp + (((int64_t)(chunk * nct + j) * nks + ks) * 3) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 3; ++q) o[q * 64] = t[q];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < out; i += gridDim.x * blockDim.x) {
        const float v = b ? b[i] : 0.f;
        bp[i] = (i % P) < K2 ? v * fold : v;
    }
}

inline int coupling_nct(int F, int P) {
    // MFMA column tiles per chunk: 3 (4 features at P = 23: 46 KB of LDS, three workgroups per CU) unless the
    // parameters of one feature need more; NF_COUPLING_NCT overrides (A/B runs)
    int want = 3;
    if (const char* e = getenv("NF_COUPLING_NCT")) want = atoi(e) > 0 ? atoi(e) : want;
    const int need = (F * P + 31) / 32, one = (P + 31) / 32;
    int nct = want > one ? want : one;
    if (nct > 4) nct = 4;                                  // (two column tiles per wave pair and chunk at most)
    return nct < need ? nct : need;
}

inline size_t coupling_lds(int row_width, int n_in, int features, int kp, int hidden, int nct, int num_blocks = 2) {
    const int rw1 = row_width | 1, ld = (kp > hidden ? kp : hidden) + 4;
    const size_t xy = (size_t)((CM * rw1 + 3) & ~3), ht = (size_t)2 * CM * ld;
    const size_t tile = (size_t)32 * nct * CLDT > ht ? (size_t)32 * nct * CLDT : ht;
    return sizeof(float) * (xy + tile + 4 * CM + (size_t)((n_in + features + 3) & ~3) + (size_t)num_blocks * 4 * hidden);
}

template <int KT, int NKS>
int launch_coupling_nks(const CouplingArgs& g, unsigned grid, size_t lds, int inverse, hipStream_t s) {
    static unsigned long long cap_done[2] = {0, 0};
    if (inverse) {
        const int rc = nf_raise_lds_cap((const void*)coupling_rqs_layer_kernel<KT, true, NKS>, &cap_done[1]);
        if (rc != NF_OK) return rc;
        hipLaunchKernelGGL((coupling_rqs_layer_kernel<KT, true, NKS>), dim3(grid), dim3(RT), lds, s, g);
    } else {
        const int rc = nf_raise_lds_cap((const void*)coupling_rqs_layer_kernel<KT, false, NKS>, &cap_done[0]);
        if (rc != NF_OK) return rc;
        hipLaunchKernelGGL((coupling_rqs_layer_kernel<KT, false, NKS>), dim3(grid), dim3(RT), lds, s, g);
    }
    return nf_launch_status();
}

template <int KT>
int launch_coupling(const CouplingArgs& g, unsigned grid, size_t lds, int inverse, hipStream_t s) {
    if (g.hidden == 32) return launch_coupling_nks<KT, 2>(g, grid, lds, inverse, s);
    return launch_coupling_nks<KT, 4>(g, grid, lds, inverse, s);
}

}  // namespace

extern "C" int nf_coupling_layer_supported(int row_width, int n_in, int hidden, int features, int K, int tails) {
    const int P = tails ? 3 * K - 1 : 3 * K + 1;
    if (row_width <= 0 || row_width > CMAXROW || n_in <= 0 || n_in > CMAXW || (hidden != 32 && hidden != 64) ||
        features <= 0 || K < 2 || P > 32 * 4)
        return 0;
    const int nct = coupling_nct(features, P);
    if (32 * nct < P) return 0;
    return coupling_lds(row_width, n_in, features, (n_in + 7) / 8 * 8, hidden, nct) <= 160 * 1024 ? 1 : 0;
}

extern "C" size_t nf_coupling_final_packed_bytes(int features, int hidden, int K, int tails) {
    if (features <= 0 || (hidden != 32 && hidden != 64) || K < 2) return 0;
    const int P = tails ? 3 * K - 1 : 3 * K + 1;
    const int nct = coupling_nct(features, P), fc = 32 * nct / P;
    if (fc < 1) return 0;
    const int nchunks = (features + fc - 1) / fc;
    return (size_t)nchunks * nct * (hidden / 16) * 3 * 1024;
}

extern "C" int nf_coupling_final_pack(const float* W, const float* bias, int features, int hidden, int K, int tails,
                                      double logit_div, void* w_packed, float* b_packed, void* stream) {
    if (!W || !w_packed || !b_packed) return NF_ERR_NULL;
    if (features <= 0 || (hidden != 32 && hidden != 64) || K < 2 || !(logit_div > 0.0)) return NF_ERR_SHAPE;
    if (((uintptr_t)w_packed & 15) != 0) return NF_ERR_LAYOUT;
    const int P = tails ? 3 * K - 1 : 3 * K + 1, out = features * P;
    const int nct = coupling_nct(features, P), fc = 32 * nct / P;
    if (fc < 1) return NF_ERR_SHAPE;
    const int nchunks = (features + fc - 1) / fc;
    const int64_t total = (int64_t)nchunks * nct * (hidden / 16) * 64;
    const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(coupling_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, bias, out, hidden, P, 2 * K,
                       (float)(1.4426950408889634 / logit_div), fc, nct, nchunks, (__bf16*)w_packed, b_packed);
    return nf_launch_status();
}

extern "C" int nf_coupling_rqs_layer_fwd(const float* x, int64_t x_row_stride, float* y, int64_t y_row_stride,
                                         int row_width, const int32_t* in_index, int n_in, const int32_t* feat_index,
                                         int features, const float* hidden_blob, int hidden, int num_blocks,
                                         const float* w_packed, const float* b_packed, float* logabsdet, int accumulate,
                                         int32_t* bin_idx, int64_t B, int K, int tails, double tail_bound,
                                         double min_bin_width, double min_bin_height, double min_derivative, int inverse,
                                         void* stream) {
    if (B < 0 || !nf_coupling_layer_supported(row_width, n_in, hidden, features, K, tails) || num_blocks < 0)
        return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    if (B == 0) return NF_OK;
    if (!x || !y || !in_index || !feat_index || !hidden_blob || !w_packed || !b_packed || !logabsdet) return NF_ERR_NULL;
    if (x_row_stride < row_width || y_row_stride < row_width || ((uintptr_t)hidden_blob & 15) != 0 ||
        ((uintptr_t)w_packed & 15) != 0)
        return NF_ERR_LAYOUT;
    CouplingArgs g;
    g.c = nf_make_rqs_const(K, tails, tail_bound, min_bin_width, min_bin_height, min_derivative, 1.0);
    g.x = x; g.ldx = x_row_stride; g.y = y; g.ldy = y_row_stride; g.row_width = row_width;
    g.in_index = in_index; g.n_in = n_in; g.kp = (n_in + 7) / 8 * 8; g.feat_index = feat_index; g.F = features;
    g.blob = hidden_blob; g.hidden = hidden; g.num_blocks = num_blocks; g.wf = w_packed; g.bf = b_packed;
    g.logabsdet = logabsdet; g.accumulate = accumulate; g.bins = bin_idx; g.M = B;
    g.nct = coupling_nct(features, g.c.P);
    g.ko = 0;
#ifdef NF_TIMING_KNOBS      // timing-only builds (benchmarks/build_fused_variant.sh): phases knocked out, results wrong
    if (const char* e = getenv("NF_COUPLING_KO")) g.ko = atoi(e);
#endif
    g.fc = 32 * g.nct / g.c.P;
    const size_t lds = coupling_lds(row_width, n_in, features, g.kp, hidden, g.nct, num_blocks);
    if (lds > 160 * 1024) return NF_ERR_SHAPE;
    const int64_t grid = (B + CM - 1) / CM;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 4: return launch_coupling<4>(g, (unsigned)grid, lds, inverse, s);
        case 8: return launch_coupling<8>(g, (unsigned)grid, lds, inverse, s);
        case 10: return launch_coupling<10>(g, (unsigned)grid, lds, inverse, s);
        case 16: return launch_coupling<16>(g, (unsigned)grid, lds, inverse, s);
        default: return launch_coupling<0>(g, (unsigned)grid, lds, inverse, s);
    }
}
