// Hidden stack of the coupling conditioner in one launch (gfx950).
//
//   h = L_init(x[:, in_index])                                        (models/nn/resnet.py:69-74, :96)
//   h = h + L1(relu(BN1(L0(relu(BN0(h))))))        x num_blocks        (resnet.py:38-57, BN in eval mode)
//
// The reference runs this as an index-select, five tiny GEMMs (hidden 32..128) and four elementwise
// passes, each a round trip of [B, hidden] through HBM.  Here a workgroup keeps 64 samples' hidden
// state in LDS from the gathered input columns to the last residual sum: HBM sees the input rows once
// and the final hidden state once (it feeds nf_coupling_rqs_fwd, the final Linear fused with the spline).
//
// Every layer is a 64 x hidden x K product on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): the A
// operand comes from LDS (ReLU and the folded BatchNorm applied as it is read), the B operand - the
// layer's weight rows, a few KB that live in L1/L2 - straight from global memory in the lane layout the
// instruction wants.  4 waves share the 2 x hidden/32 output tiles of a layer.
#include "nf_common.h"
#include "nf_resnet_layer.h"

namespace {

constexpr int RM = 64;          // samples per workgroup
constexpr int MAXW = 128;       // widest layer (inputs or hidden units)

struct ResnetArgs {
    const float* x;
    int64_t ldx;
    const int32_t* in_index;    // nullable: input column c of the stack is x[:, in_index[c]]
    int n_in, kp;               // inputs, padded to a multiple of 8
    const float* blob;          // W0[hidden][kp] b0[hidden] { s0 t0 Wa[hidden][hidden] ba s1 t1 Wb[hidden][hidden] bb } x blocks
    int hidden, num_blocks;
    float* h;
    int64_t ldh;
    int64_t M;
    float* copy_out;            // nullable: the gathered input columns are also written to copy_out[:, in_index[c]]
    int64_t ldc;
};

__global__ __launch_bounds__(RT) void resnet_hidden_kernel(const ResnetArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wmax = max(g.kp, g.hidden);
    const int ld = wmax + 4;                    // row stride = 4 mod 32 floats: conflict-free b128 operand reads
    float* H = smem;                            // [64][ld] hidden state
    float* T = smem + RM * ld;                  // [64][ld] inputs, then the block's inner activations
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * RM;

    // gather the input columns (rows past the batch and the K padding are zero)
    for (int e = tid; e < RM * g.kp; e += RT) {
        const int r = e / g.kp, c = e - r * g.kp;
        float v = 0.f;
        if (m0 + r < g.M && c < g.n_in) {
            const int col = g.in_index ? g.in_index[c] : c;
            v = g.x[(m0 + r) * g.ldx + col];
            if (g.copy_out) g.copy_out[(m0 + r) * g.ldc + col] = v;      // a coupling layer's identity half passes through
        }
        T[r * ld + c] = v;
    }
    __syncthreads();

    const int hd = g.hidden;
    const float* p = g.blob;
    layer<false, false>(T, ld, g.kp, p, p + (int64_t)hd * g.kp, nullptr, nullptr, H, hd, wave, lane);
    p += (int64_t)hd * g.kp + hd;
    __syncthreads();
    for (int b = 0; b < g.num_blocks; ++b) {
        const float *s0 = p, *t0 = p + hd, *Wa = p + 2 * hd, *ba = Wa + (int64_t)hd * hd;
        const float *s1 = ba + hd, *t1 = s1 + hd, *Wb = t1 + hd, *bb = Wb + (int64_t)hd * hd;
        layer<true, false>(H, ld, hd, Wa, ba, s0, t0, T, hd, wave, lane);
        __syncthreads();
        layer<true, true>(T, ld, hd, Wb, bb, s1, t1, H, hd, wave, lane);
        __syncthreads();
        p = bb + hd;
    }
    // hidden state out, rows contiguous
    const int q = hd / 4;
    for (int e = tid; e < RM * q; e += RT) {
        const int r = e / q, c = (e - r * q) * 4;
        if (m0 + r < g.M) *reinterpret_cast<float4*>(g.h + (m0 + r) * g.ldh + c) = *reinterpret_cast<const float4*>(H + r * ld + c);
    }
}

}  // namespace

extern "C" size_t nf_resnet_blob_floats(int n_in, int hidden, int num_blocks) {
    if (n_in <= 0 || hidden <= 0 || num_blocks < 0) return 0;
    const size_t kp = (size_t)(n_in + 7) / 8 * 8, h = (size_t)hidden;
    return h * kp + h + (size_t)num_blocks * (2 * (2 * h + h * h + h));
}

static int resnet_hidden_impl(const float* x, int64_t x_row_stride, const int32_t* in_index, int n_in,
                              int row_width, const float* blob, int hidden, int num_blocks, float* h,
                              int64_t h_row_stride, float* copy_out, int64_t copy_row_stride, int64_t B, void* stream) {
    if (B < 0 || n_in <= 0 || n_in > MAXW || hidden <= 0 || hidden > MAXW || hidden % 32 != 0 || num_blocks < 0 ||
        row_width < n_in)
        return NF_ERR_SHAPE;
    if (B == 0) return NF_OK;
    if (!x || !blob || !h) return NF_ERR_NULL;
    if (x_row_stride < row_width || h_row_stride < hidden || h_row_stride % 4 != 0 || ((uintptr_t)h & 15) != 0 ||
        ((uintptr_t)blob & 15) != 0)
        return NF_ERR_LAYOUT;
    ResnetArgs g;
    g.x = x; g.ldx = x_row_stride; g.in_index = in_index; g.n_in = n_in; g.kp = (n_in + 7) / 8 * 8;
    g.blob = blob; g.hidden = hidden; g.num_blocks = num_blocks; g.h = h; g.ldh = h_row_stride; g.M = B;
    g.copy_out = copy_out; g.ldc = copy_row_stride;
    const int wmax = g.kp > hidden ? g.kp : hidden;
    const size_t lds = sizeof(float) * 2 * RM * (size_t)(wmax + 4);
    const int64_t grid = (B + RM - 1) / RM;
    if (grid > 0x7fffffff) return NF_ERR_SHAPE;
    static unsigned long long cap_done = 0;
    const int rc = nf_raise_lds_cap((const void*)resnet_hidden_kernel, &cap_done);
    if (rc != NF_OK) return rc;
    hipLaunchKernelGGL(resnet_hidden_kernel, dim3((unsigned)grid), dim3(RT), lds, (hipStream_t)stream, g);
    return nf_launch_status();
}

extern "C" int nf_resnet_hidden_fwd(const float* x, int64_t x_row_stride, const int32_t* in_index, int n_in,
                                    int row_width, const float* blob, int hidden, int num_blocks, float* h,
                                    int64_t h_row_stride, int64_t B, void* stream) {
    return resnet_hidden_impl(x, x_row_stride, in_index, n_in, row_width, blob, hidden, num_blocks, h, h_row_stride,
                              nullptr, 0, B, stream);
}

// The same launch also copies the gathered columns into `copy_out` (same column positions): a coupling layer's identity
// half reaches the layer's output without the `outputs = inputs.clone()` pass (transforms.py:98-106 writes both halves).
extern "C" int nf_resnet_hidden_fwd_copy(const float* x, int64_t x_row_stride, const int32_t* in_index, int n_in,
                                         int row_width, const float* blob, int hidden, int num_blocks, float* h,
                                         int64_t h_row_stride, float* copy_out, int64_t copy_row_stride, int64_t B,
                                         void* stream) {
    if (!copy_out && B > 0) return NF_ERR_NULL;
    if (copy_row_stride < row_width) return NF_ERR_LAYOUT;
    return resnet_hidden_impl(x, x_row_stride, in_index, n_in, row_width, blob, hidden, num_blocks, h, h_row_stride,
                              copy_out, copy_row_stride, B, stream);
}
