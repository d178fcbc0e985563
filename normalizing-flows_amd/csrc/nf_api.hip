// Version / diagnostics entry points of libnf_hip.so (see include/nf_hip.h).
#include "nf_common.h"

extern "C" int nf_version(void) { return 100; }

extern "C" const char* nf_strerror(int code) {
    switch (code) {
        case NF_OK: return "ok";
        case NF_ERR_NULL: return "a required pointer is NULL";
        case NF_ERR_SHAPE: return "sizes inconsistent or outside the supported range";
        case NF_ERR_LAYOUT: return "stride or alignment not accepted";
        case NF_ERR_BINS: return "minimal bin width/height too large for the number of bins";
        case NF_ERR_DTYPE: return "unknown dtype or layout";
        case NF_ERR_SCRATCH:
            return "workspace too small (nf_made_rqs_fwd_noalloc: the batch exceeds its static scratch, tiles * B * 4 bytes "
                   "must fit 8 MiB)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

// ---- diagnostics: achievable peaks of this device (used by bench.py beside the nominal roofline peaks) ----
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_f32_loop_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0];
    if (s == 12345.678f) out[0] = s;           // keeps the loop alive, never true in practice
}

__global__ __launch_bounds__(256) void mfma_bf16_loop_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + 0.01f * j + threadIdx.x * 1e-3f); b[j] = (__bf16)(0.5f - 0.01f * j); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0];
    if (s == 12345.678f) out[0] = s;
}

// the same loop on v_mfma_f32_32x32x16_f16 with operands that differ per lane and change from one instruction to the
// next (four pseudo-random A and B vectors taken in rotation): what the fp16 matrix pipe sustains, and draws, when its
// operand bits toggle like real data (the constant-operand loops above read high: their datapath barely switches)
typedef _Float16 f16x8d __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_f16_random_loop_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8d a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            a[q][j] = (_Float16)(((int)(h >> 9) & 0xffff) * (1.0f / 32768.0f) - 1.0f);
            h = h * 1664525u + 1013904223u;
            b[q][j] = (_Float16)(((int)(h >> 9) & 0xffff) * (1.0f / 32768.0f) - 1.0f);
        }
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + j) & 3], b[(u * 3 + j) & 3], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += acc[j][0];
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace

// Launches `blocks` x 256 threads each issuing iters*4 back-to-back MFMAs (dtype 0: 32x32x2 f32 = 4096 FLOP,
// dtype 1: 32x32x16 bf16 = 32768 FLOP per wave instruction; dtype NF_DTYPE_F16X2: 32x32x16 f16 on pseudo-random,
// rotating operands, iters a multiple of 4).  FLOP = blocks*4 waves*iters*4*that.
extern "C" int nf_diag_mfma_loop(float* scratch, int blocks, int iters, int dtype, void* stream) {
    if (!scratch) return NF_ERR_NULL;
    if (blocks <= 0 || iters <= 0) return NF_ERR_SHAPE;
    if (dtype == NF_DTYPE_F32) hipLaunchKernelGGL(mfma_f32_loop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
    else if (dtype == NF_DTYPE_F16X2)
        hipLaunchKernelGGL(mfma_f16_random_loop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
    else hipLaunchKernelGGL(mfma_bf16_loop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
    return nf_launch_status();
}

// dst[0..bytes) = src[0..bytes) with 16-byte accesses (bytes % 16 == 0): the achievable HBM copy rate.
extern "C" int nf_diag_stream_copy(const void* src, void* dst, size_t bytes, void* stream) {
    if (!src || !dst) return NF_ERR_NULL;
    if (bytes % 16) return NF_ERR_LAYOUT;
    hipLaunchKernelGGL(stream_copy_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst,
                       (int64_t)(bytes / 16));
    return nf_launch_status();
}
