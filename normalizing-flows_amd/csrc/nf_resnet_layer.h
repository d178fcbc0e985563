// One dense layer of the coupling conditioner's hidden stack on a 64-sample LDS tile (shared by nf_resnet.hip and
// nf_coupling.hip; models/nn/resnet.py:38-57, :96).  256 threads = 4 waves share the 2 x hidden/32 output tiles.
#pragma once
#include <hip/hip_runtime.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RT = 256;         // threads

// dst[64][hidden] (+)= act(src[64][K]) @ W[hidden][K]^T + bias; act(v) = relu(v * sc[k] + sh[k]) when ACT.
template <bool ACT, bool RESIDUAL>
__device__ __forceinline__ void layer(const float* __restrict__ src, int lds_ld, int K, const float* __restrict__ W,
                                      const float* __restrict__ bias, const float* __restrict__ sc,
                                      const float* __restrict__ sh, float* __restrict__ dst, int hidden, int wave,
                                      int lane) {
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    const int ntile = 2 * (hidden / 32);
    for (int t = wave; t < ntile; t += RT / 64) {
        const int i = t & 1, j = t >> 1;
        const int col = j * 32 + frow;
        const float bv = bias[col];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
        const float* ap = src + (i * 32 + frow) * lds_ld + fk;
        const float* wp = W + (int64_t)col * K + fk;
        for (int kc = 0; kc < K; kc += 8) {
            float4 a = *reinterpret_cast<const float4*>(ap + kc);
            const float4 b = *reinterpret_cast<const float4*>(wp + kc);
            if (ACT) {
                const float4 s = *reinterpret_cast<const float4*>(sc + kc + fk);
                const float4 h = *reinterpret_cast<const float4*>(sh + kc + fk);
                a.x = fmaxf(fmaf(a.x, s.x, h.x), 0.f);
                a.y = fmaxf(fmaf(a.y, s.y, h.y), 0.f);
                a.z = fmaxf(fmaf(a.z, s.z, h.z), 0.f);
                a.w = fmaxf(fmaf(a.w, s.w, h.w), 0.f);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float* d = dst + row * lds_ld + col;
            *d = RESIDUAL ? *d + acc[r] : acc[r];
        }
    }
}

}  // namespace
