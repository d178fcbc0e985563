// Shared declarations for the gfx950 kernels behind include/nf_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Return codes of the C-ABI: 0 ok, negative = argument error (below), positive = hipError_t.
#define NF_OK 0
#define NF_ERR_NULL (-1)        // a required pointer is null
#define NF_ERR_SHAPE (-2)       // sizes inconsistent or out of the supported range
#define NF_ERR_LAYOUT (-3)      // stride / alignment the kernels do not accept
#define NF_ERR_BINS (-4)        // min_bin_width/height * num_bins > 1 (reference raises ValueError)
#define NF_ERR_DTYPE (-5)       // unknown dtype / layout enum
#define NF_ERR_SCRATCH (-6)     // workspace too small

#define NF_DTYPE_F32 0
#define NF_DTYPE_BF16 1
#define NF_DTYPE_F16X2 3          // fp32 operands as two fp16 terms (weights power-of-two scaled), three fp16 MFMAs per product
#define NF_DTYPE_BF16X3 2         // fp32 operands as three bf16 terms, six bf16 MFMAs per product (fused final layer only)

#define NF_AFFINE_INTERLEAVED 0  // params[b, 2d] = unconstrained scale, [2d+1] = shift (autoregressive)
#define NF_AFFINE_BLOCKED 1      // params[b, d] = shift, params[b, D+d] = unconstrained scale (coupling)

static inline int nf_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NF_OK : (int)e;
}

static inline int nf_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// MaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: raise it once per (kernel, device).
// `done` is the call site's bitmask over device ordinals 0..63 (atomic: entry points are thread-safe for distinct
// streams); ordinals past 63 set the attribute on every launch.  Not an allocation.
static inline int nf_raise_lds_cap(const void* kernel, unsigned long long* done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? 1ull << dev : 0ull;
    if (bit && (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit)) return NF_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (bit) __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
    return NF_OK;
}
