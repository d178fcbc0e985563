// Autoregressive inverse (sampling) of a rational-quadratic spline MAF layer, row-resident (gfx950).
//
// The reference inverts feature by feature with D full conditioner passes (models/transforms.py:493-500).
// The MADE masks (models/made.py:119-148) make that a recurrence with very little new work per feature:
// with hidden units sorted by degree (unit j has degree j+1, every hidden layer alike), output feature f
// reads hidden units 0..f-1 only, and input f reaches unit k only for k >= f.  So once x[f] is known
//   * unit f of every hidden layer becomes final (nothing later can change it), and
//   * the parameters of feature f+1 are one 32-column row block of the final layer over units 0..f.
// The samples never interact, so a workgroup keeps 32 samples from x[0] to x[H] without any grid-wide step:
//
//   hidden layers, "right-looking": thread k owns unit k of every hidden layer and keeps its running
//     pre-activation for all 32 samples in registers (NSTEP x 32 fp32).  When unit j becomes final its
//     activation (32 values) is broadcast through LDS and every thread adds its one weight times it.
//     Per feature and layer: one 2 KB weight row (transposed pack, coalesced, prefetched a pass ahead),
//     32 FMAs per thread, one barrier.  Total = the masked MACs of one forward pass.
//   final layer, "left-looking": the final hidden state so far lives in LDS ([unit][sample]); the P <= 32
//     parameter columns of feature f+1 are a 32 x 32 x f product on v_mfma_f32_32x32x2_f32, K split over the
//     8 waves (weights packed per feature in operand order, triangular: only the live K range is stored
//     and read), partial tiles summed in fixed order through LDS, then 32 lanes invert the spline
//     (nf_rqs_eval.h, the code the forward kernels use) and broadcast x[f+1].
//
// Features past H+1 depend on no further hidden unit: the kernel leaves the final hidden state in HBM and
// the caller finishes them with one windowed launch of the fused layer kernel (nf_made_rqs_fwd_tiles).
// Work per layer = about one forward pass (the reference: D forward passes; the windowed incremental inverse
// in transforms.py: one 128/256-column tile per layer per feature, ~10-100 x more MACs than needed).
#include "nf_common.h"
#include "nf_rqs_eval.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int IR = 32;            // samples per workgroup
constexpr int IT = 512;           // threads = hidden units a workgroup can own
constexpr int IW = IT / 64;
constexpr int MAXSTEP = 5;        // hidden vectors: initial layer + 2 per residual block (1 per feed-forward block)
constexpr int PSTR = 33;          // stride of the parameter tile [column][sample]
constexpr int SS = 8;             // hidden units per K super-step of the final layer (4 MFMAs of K = 2)
#ifndef NF_INV_SLOTS
#define NF_INV_SLOTS 2
#endif
constexpr int NB = NF_INV_SLOTS;  // B-operand register slots per wave
constexpr int MW = IW - 1;        // waves on the final-layer product (wave 0 inverts the spline meanwhile)
constexpr int NI = (IT / SS + MW - 1) / MW;   // super-steps per wave at most

constexpr int F_RELU_IN = 1, F_RESIDUAL = 2, F_RELU_OUT = 4;

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
    const float* wl;              // [H+1][32] wl[f][c] = W[f*P+c][f-1]: the newest hidden unit feature f reads
    const int32_t* off;           // [H+2] first super-step of each feature in wf
    const float4* wf;             // [super-step][lane]: lane (kh, col) holds W[f*P+col][8S+4kh .. +3]
    int flags[MAXSTEP];
    int64_t B;
    int H, HS, fpt, full_tiles;
    RqsConst c;
};

template <int NSTEP, int KT>
__global__ __launch_bounds__(IT, 1) void made_rqs_inverse_kernel(const InvArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hlT = smem;                              // [HS][IR] final hidden state, unit-major
    float* part = hlT + g.HS * IR;                  // [MW][IR][32] per-wave partial parameter tiles
    float* prm = part + MW * IR * 32;               // [32][PSTR]
    float* bc = prm + 32 * PSTR;                    // [NSTEP][IR] broadcast slots
    float* xs = bc + MAXSTEP * IR;                  // [IR] the feature just inverted

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * IR;
    const int H = g.H, HS = g.HS;
    const bool unit = tid < H;
    const int kcol = unit ? tid : 0;

    f32x2 acc[NSTEP][IR / 2];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const float b = unit ? g.bias[s][kcol] : 0.0f;
#pragma unroll
        for (int r = 0; r < IR / 2; ++r) acc[s][r] = (f32x2){b, b};
    }
    for (int e = tid; e < HS * IR; e += IT) hlT[e] = 0.0f;

    const bool ev = tid < IR;
    const int64_t row = (m0 + tid < g.B) ? m0 + tid : g.B - 1;
    const bool row_ok = ev && m0 + tid < g.B;
    float znext = ev ? g.z[row * g.ldz] : 0.0f;
    float ldsum = 0.0f;

    float wcur[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wcur[s] = g.wt[s][kcol];
    float4 bw[NB];                // B operands of this wave's first NB super-steps (the rest follow in place)
#pragma unroll
    for (int i = 0; i < NB; ++i) bw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int arow = lane & 31, kh = lane >> 5;
    // parameters of feature 0: the bias alone
    for (int e = tid; e < IR * 32; e += IT) prm[(e & 31) * PSTR + (e >> 5)] = g.bf[e & 31];
    const int rc = tid & 31, rr = tid >> 5;     // parameter sums: this thread owns column rc of samples rr, rr + 16
    float wl = g.wl[32 + rc], bfv = g.bf[32 + rc];
    __syncthreads();

#ifdef NF_INV_STAMP
    long long st[6] = {0, 0, 0, 0, 0, 0};
#define STAMP(i_) { const long long n_ = clock64(); st[i_] += n_ - tlast; tlast = n_; }
    long long tlast = clock64();
#else
#define STAMP(i_)
#endif
    for (int f = 0; f <= H; ++f) {
        if (wave == 0) {
            // ---- wave 0: invert feature f (its parameters were completed at the end of the previous pass)
            if (lane < IR) {
                float y, ld;
                int bin;
#ifdef NF_NOEVAL
                y = prm[tid] + znext; ld = prm[tid + PSTR]; bin = 0;
#else
                typedef const __attribute__((address_space(3))) float* lds_ptr;
                rqs_eval<KT, true, PSTR, false, false, lds_ptr>((lds_ptr)(prm + tid), znext, g.c, y, ld, bin);
#endif
                xs[tid] = y;
                if (row_ok) g.x[row * g.ldx + f] = y;
                ldsum += ld;
                if ((f + 1) % g.fpt == 0) {
                    const int t = f / g.fpt;
                    if (row_ok && t < g.full_tiles) g.partial[(int64_t)t * g.B + row] = ldsum;
                    ldsum = 0.0f;
                }
                if (f < H) znext = g.z[row * g.ldz + f + 1];
            }
        } else if (f < H) {
            // ---- waves 1..7 meanwhile: feature f+1 over the hidden units that are already final (0..f-1);
            //      unit f, which this pass is about to finish, is added as a rank-1 term at the end
            const int NS = (f + SS - 1) / SS;
            f32x16 d;
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = 0.0f;
            const int64_t fo = g.off[f + 1];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int S = wave - 1 + MW * i;
                if (S < NS) {
                    const float* ap = hlT + (SS * S + 4 * kh) * IR + arow;
                    const float a0 = ap[0], a1 = ap[IR], a2 = ap[2 * IR], a3 = ap[3 * IR];
                    const float4 b = bw[i % NB];
                    // the slot's next user (NB super-steps of this wave later) is loaded behind these MFMAs
                    if (i + NB < NI && S + NB * MW < NS) bw[i % NB] = g.wf[(fo + S + NB * MW) * 64 + lane];
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b.x, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b.y, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b.z, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b.w, d, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = (i & 3) + 8 * (i >> 2) + 4 * kh;
                part[((wave - 1) * IR + r) * 32 + arow] = d[i];
            }
        }
        STAMP(0)
        if (f == H) break;

        // operands of the next pass: issued now, used a whole hidden chain later
        if (wave > 0 && f + 1 < H) {
            const int NS1 = (f + SS) / SS;
            const int64_t o = g.off[f + 2];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int S = wave - 1 + MW * i;
                if (S < NS1) bw[i] = g.wf[(o + S) * 64 + lane];
            }
        }
        __syncthreads();
        STAMP(1)

        // ---- x[f] is known: unit j = f of every hidden layer becomes final, one layer after the other
        const int j = f;
        const bool live = wave * 64 + 63 >= j;          // units below j are final: their weights are masked
        const bool mine = tid == j;
        if (live) {
            const f32x2 w2 = (f32x2){wcur[0], wcur[0]};
            const f32x2* v2 = reinterpret_cast<const f32x2*>(xs);
#pragma unroll
            for (int r = 0; r < IR / 2; ++r) acc[0][r] = __builtin_elementwise_fma(w2, v2[r], acc[0][r]);
        }
        // (each weight register is reloaded for the next pass right after its use: a whole pass of latency cover)
        const int64_t wrow = (int64_t)(f + 1 < H ? f + 1 : H - 1) * HS + kcol;
        wcur[0] = g.wt[0][wrow];
#pragma unroll
        for (int s = 1; s < NSTEP; ++s) {
            if (mine) {
                f32x2* o2 = reinterpret_cast<f32x2*>(bc + s * IR);
                const bool relu = g.flags[s] & F_RELU_IN;
#pragma unroll
                for (int r = 0; r < IR / 2; ++r) {
                    f32x2 v = acc[s - 1][r];
                    if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); }
                    o2[r] = v;
                }
            }
            __syncthreads();
            if (live) {
                const f32x2 w2 = (f32x2){wcur[s], wcur[s]};
                const f32x2* v2 = reinterpret_cast<const f32x2*>(bc + s * IR);
#pragma unroll
                for (int r = 0; r < IR / 2; ++r) acc[s][r] = __builtin_elementwise_fma(w2, v2[r], acc[s][r]);
            }
            wcur[s] = g.wt[s][wrow];
            if (mine) {
                if (s >= 2 && (g.flags[s] & F_RESIDUAL)) {
#pragma unroll
                    for (int r = 0; r < IR / 2; ++r) acc[s][r] = acc[s][r] + acc[s >= 2 ? s - 2 : 0][r];
                }
                if (g.flags[s] & F_RELU_OUT) {
#pragma unroll
                    for (int r = 0; r < IR / 2; ++r) {
                        acc[s][r].x = fmaxf(acc[s][r].x, 0.0f);
                        acc[s][r].y = fmaxf(acc[s][r].y, 0.0f);
                    }
                }
            }
        }
        if (mine) {
            f32x2* o2 = reinterpret_cast<f32x2*>(hlT + j * IR);
#pragma unroll
            for (int r = 0; r < IR / 2; ++r) o2[r] = acc[NSTEP - 1][r];
        }
        STAMP(2)
        __syncthreads();
        STAMP(3)

        // ---- parameters of feature f+1: bias + the partial tiles (fixed order) + unit f's rank-1 term
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = rr + 16 * h;
            float v = bfv;
#pragma unroll
            for (int w = 0; w < MW; ++w) v += part[(w * IR + r) * 32 + rc];
            prm[rc * PSTR + r] = fmaf(hlT[j * IR + r], wl, v);
        }
        {
            const int fn = f + 2 <= H ? f + 2 : H;
            wl = g.wl[fn * 32 + rc];
            bfv = g.bf[fn * 32 + rc];
        }
        STAMP(4)
        __syncthreads();
        STAMP(5)
    }
#ifdef NF_INV_STAMP
    if (blockIdx.x == 0 && (tid & 63) == 0)
        printf("wave %d: work %lld | barC %lld | chain %lld | barD %lld | reduce %lld | barB %lld (clock64 ticks per pass)\n", wave,
               st[0] / H, st[1] / H, st[2] / H, st[3] / H, st[4] / H, st[5] / H);
#endif

    // final hidden state for the features past H (one windowed launch of the fused layer kernel)
    if (unit) {
#pragma unroll
        for (int r = 0; r < IR / 2; ++r) {
            if (m0 + 2 * r < g.B) g.hid[(m0 + 2 * r) * g.ldh + tid] = acc[NSTEP - 1][r].x;
            if (m0 + 2 * r + 1 < g.B) g.hid[(m0 + 2 * r + 1) * g.ldh + tid] = acc[NSTEP - 1][r].y;
        }
    }
}

inline int hs_of(int H) { return (H + 63) / 64 * 64; }
inline int64_t total_supersteps(int H) {
    int64_t t = 0;
    for (int f = 0; f <= H; ++f) t += (f + SS - 1) / SS;
    return t;
}
inline int off_len(int H) { return (H + 2 + 3) / 4 * 4; }

template <int NSTEP>
int launch_steps(const InvArgs& g, int K, unsigned grid, size_t lds, hipStream_t s) {
    auto go = [&](auto kern) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(IT), lds, s, g);
        return nf_launch_status();
    };
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
    return n_steps * (H + 1) * HS + 2 * (H + 1) * 32 + off_len(hidden) + (size_t)total_supersteps(hidden) * 256;
}

extern "C" int nf_made_rqs_inverse_rows(const float* blob, int hidden, int n_steps, const int* step_flags,
                                        const float* z, int64_t z_row_stride, float* x, int64_t x_row_stride,
                                        float* hidden_out, int64_t h_row_stride, float* workspace,
                                        size_t workspace_bytes, int64_t B, int features, int K, int tails,
                                        double tail_bound, double min_bin_width, double min_bin_height,
                                        double min_derivative, void* stream) {
    if (B < 0 || !nf_made_rqs_inverse_rows_supported(features, hidden, n_steps, K, tails)) return NF_ERR_SHAPE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_ERR_BINS;
    if (B == 0) return NF_OK;
    if (!blob || !step_flags || !z || !x || !hidden_out || !workspace) return NF_ERR_NULL;
    if (z_row_stride < features || x_row_stride < features || h_row_stride < hidden || ((uintptr_t)blob & 15) != 0)
        return NF_ERR_LAYOUT;
    InvArgs g;
    g.c = nf_make_rqs_const(K, tails, tail_bound, min_bin_width, min_bin_height, min_derivative, 1.0);
    g.fpt = 256 / g.c.P;
    const int all_tiles = (features + g.fpt - 1) / g.fpt;
    if (workspace_bytes < sizeof(float) * (size_t)all_tiles * (size_t)B) return NF_ERR_SCRATCH;
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
    g.wl = p; p += (size_t)(H + 1) * 32;
    g.off = reinterpret_cast<const int32_t*>(p); p += off_len(H);
    g.wf = reinterpret_cast<const float4*>(p);
    g.z = z; g.ldz = z_row_stride; g.x = x; g.ldx = x_row_stride; g.hid = hidden_out; g.ldh = h_row_stride;
    g.partial = workspace; g.B = B; g.H = H; g.HS = HS;
    g.full_tiles = (H + 1) / g.fpt;
    const int64_t grid = (B + IR - 1) / IR;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    const size_t lds = sizeof(float) * ((size_t)HS * IR + MW * IR * 32 + 32 * PSTR + MAXSTEP * IR + IR);
    hipStream_t st = (hipStream_t)stream;
    switch (n_steps) {
        case 2: return launch_steps<2>(g, K, (unsigned)grid, lds, st);
        case 3: return launch_steps<3>(g, K, (unsigned)grid, lds, st);
        case 4: return launch_steps<4>(g, K, (unsigned)grid, lds, st);
        case 5: return launch_steps<5>(g, K, (unsigned)grid, lds, st);
    }
    return NF_ERR_SHAPE;
}
