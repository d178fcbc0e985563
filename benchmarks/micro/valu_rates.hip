// Issue cost of the VALU instructions the spline evaluators lean on, in cycles per wave64 instruction on one SIMD
// (4 waves per SIMD keep the pipe full; dependent chains are broken by 8 independent accumulators per lane).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 4096;

#define KERNEL(NAME, DECL, BODY, SINK)                                                         \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed) {                      \
        DECL;                                                                                  \
        for (int i = 0; i < ITERS; ++i) {                                                      \
            BODY                                                                               \
        }                                                                                      \
        out[blockIdx.x * 256 + threadIdx.x] = SINK;                                            \
    }

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DECLF float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7
#define DECLD double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7
#define SUMF (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)

#define OP_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##i) : "v"(seed));
#define OP_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
#define OP_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(a##i));
#define OP_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
#define OP_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##i));
#define OP_ADD64(i) asm volatile("v_add_f64 %0, %0, %0" : "+v"(a##i));
#define OP_FMA64(i) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a##i));
#define OP_CVT(i) { double d_; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d_) : "v"(a##i)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a##i) : "v"(d_)); }
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(seed));
#define OP_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(seed));
#define OP_MULLO(i) { int t_ = __float_as_int(a##i); asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(t_)); a##i = __int_as_float(t_); }

KERNEL(k_add, DECLF, R8(OP_ADD), SUMF)
KERNEL(k_exp, DECLF, R8(OP_EXP), SUMF)
KERNEL(k_log, DECLF, R8(OP_LOG), SUMF)
KERNEL(k_rcp, DECLF, R8(OP_RCP), SUMF)
KERNEL(k_sqrt, DECLF, R8(OP_SQRT), SUMF)
KERNEL(k_add64, DECLD, R8(OP_ADD64), (float)SUMF)
KERNEL(k_fma64, DECLD, R8(OP_FMA64), (float)SUMF)
KERNEL(k_cvtpair, DECLF, R8(OP_CVT), SUMF)
KERNEL(k_cndmask, DECLF, R8(OP_CNDMASK), SUMF)
KERNEL(k_max3, DECLF, R8(OP_MAX3), SUMF)
#define OP_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a##i) : "v"(seed), "s"(msk));
KERNEL(k_cnd64, DECLF; unsigned long long msk = __ballot(threadIdx.x & 1), R8(OP_CND64), SUMF)
#define OP_CMPCND(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(seed) : "vcc");
KERNEL(k_cmpcnd, DECLF, R8(OP_CMPCND), SUMF)
#define OP_CMPCND64(i) { unsigned long long m_; asm volatile("v_cmp_gt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a##i), "=s"(m_) : "v"(seed)); }
KERNEL(k_cmpcnd64, DECLF, R8(OP_CMPCND64), SUMF)
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_fma, DECLF, R8(OP_FMA), SUMF)
#define OP_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_mul, DECLF, R8(OP_MUL), SUMF)
#define OP_ADD3(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a##i) : "v"(seed));
KERNEL(k_add_e64, DECLF, R8(OP_ADD3), SUMF)
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a##i) : "v"(seed));
KERNEL(k_mov, DECLF, R8(OP_MOV), SUMF)
KERNEL(k_mullo, DECLF, R8(OP_MULLO), SUMF)

__global__ __launch_bounds__(256) void k_pkmul(float* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0, a5 = a1, a6 = a2, a7 = a3;
    for (int i = 0; i < ITERS; ++i) {
#define OP_PK(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a##i));
        R8(OP_PK)
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <class K>
void run(const char* name, K kern, int per_iter, float* out) {
    const int grid = 256 * 4;          // 4 blocks of 4 waves per CU: 4 waves per SIMD
    hipEvent_t s, e;
    CHECK(hipEventCreate(&s)); CHECK(hipEventCreate(&e));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(s));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 1.0f);
    CHECK(hipEventRecord(e));
    CHECK(hipEventSynchronize(e));
    float ms; CHECK(hipEventElapsedTime(&ms, s, e));
    // wave-instructions per SIMD = 4 waves * ITERS * per_iter; cycles at the clock the run held (reported at 2.4 GHz)
    const double insts = 4.0 * ITERS * per_iter;
    printf("%-14s %8.3f ms  %6.2f ns per wave-instruction per SIMD  (= %5.1f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 4 * 256 * sizeof(float)));
    run("add_f32", k_add, 8, out);
    run("pk_mul", k_pkmul, 8, out);
    run("exp_f32", k_exp, 8, out);
    run("log_f32", k_log, 8, out);
    run("rcp_f32", k_rcp, 8, out);
    run("sqrt_f32", k_sqrt, 8, out);
    run("add_f64", k_add64, 8, out);
    run("fma_f64", k_fma64, 8, out);
    run("cvt pair", k_cvtpair, 16, out);
    run("cndmask", k_cndmask, 8, out);
    run("max3", k_max3, 8, out);
    run("cnd e64 sgpr", k_cnd64, 8, out);
    run("cmp+cnd vcc", k_cmpcnd, 16, out);
    run("cmp+cnd e64", k_cmpcnd64, 16, out);
    run("fma_f32", k_fma, 8, out);
    run("mul_f32", k_mul, 8, out);
    run("add_f32_e64", k_add_e64, 8, out);
    run("mov_b32", k_mov, 8, out);
    run("mul_lo_u32", k_mullo, 8, out);
    CHECK(hipFree(out));
    return 0;
}
