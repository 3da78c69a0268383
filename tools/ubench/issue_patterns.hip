// Micro-benchmark (round 5): how the vector instruction CLASSES of the pair loop share the issue slots of a SIMD when they are mixed.
// F = double-rate (v_mul_f32 / v_fma_f32: 2.6 cycles alone), S = full-rate (v_cmp / v_min / v_ffbl / v_alignbit / v_lshl_add: 4.4 alone),
// T = transcendental (v_rcp_f32: 8.4 alone).  Independent registers unless the name says "dep".  Cycles per BODY per SIMD, 6 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OUTER 64
#define DEFI(NAME, NI, BODY)                                                                              \
    __global__ void __launch_bounds__(64) NAME(float* sink) {                                             \
        float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f, e = 1.1f, f = 0.9f, g = 1.3f, h = 0.8f; \
        unsigned ia = threadIdx.x, ib = 3, ic = 5, id = 7;                                                \
        unsigned long long s0 = 1, s1 = 2;                                                                \
        _Pragma("unroll 1") for (int it = 0; it < OUTER; ++it) {                                          \
            asm volatile(".rept 32\n" BODY "\n.endr"                                                      \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id), "+s"(s0), "+s"(s1), \
                           "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "vcc", "scc");                           \
        }                                                                                                 \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h + ia + ib + ic + id + (float)(s0 + s1); \
    }                                                                                                     \
    static const int NAME##_n = NI;
// %0-%3, %10-%13 float; %4-%7 uint; %8, %9 sgpr pairs
#define F0 "v_mul_f32 %0,%0,%1\n"
#define F1 "v_mul_f32 %2,%2,%3\n"
#define F2 "v_fma_f32 %10,%10,%11,%11\n"
#define F3 "v_fma_f32 %12,%12,%13,%13\n"
#define S0 "v_cmp_lt_u32 %8,%4,%5\n"
#define S1 "v_min_f32 %1,%1,%3\n"
#define S2 "v_lshl_add_u32 %6,%6,5,%7\n"
#define S3 "v_alignbit_b32 %7,%7,%5,31\n"
#define T0 "v_rcp_f32 %11,%11\n"
#define T1 "v_sqrt_f32 %13,%13\n"
DEFI(p_FFFF, 4, F0 F1 F2 F3)
DEFI(p_SSSS, 4, S0 S1 S2 S3)
DEFI(p_SFSF, 4, S0 F0 S1 F1)
DEFI(p_SSFF, 4, S0 S1 F0 F1)
DEFI(p_SFFF, 4, S0 F0 F1 F2)
DEFI(p_SSSF, 4, S0 S1 S2 F0)
DEFI(p_SFF, 3, S0 F0 F1)
DEFI(p_SF, 2, S2 F0)
DEFI(p_align4, 4, "v_alignbit_b32 %4,%4,%5,31\n v_alignbit_b32 %4,%4,%6,31\n v_alignbit_b32 %4,%4,%7,31\n v_alignbit_b32 %4,%4,%5,31\n")
DEFI(p_align_dep_F, 8, "v_alignbit_b32 %4,%4,%5,31\n" F0 "v_alignbit_b32 %4,%4,%6,31\n" F1 "v_alignbit_b32 %4,%4,%7,31\n" F2 "v_alignbit_b32 %4,%4,%5,31\n" F3)
DEFI(p_TFFF, 4, T0 F0 F1 F3)
DEFI(p_TFFFFFFF, 8, T0 F0 F1 F3 F0 F1 F3 F0)
#define F4 F0 F1 F2 F3
DEFI(p_T_F15, 16, T0 F0 F1 F3 F4 F4 F4)
DEFI(p_T_F31, 32, T0 F0 F1 F3 F4 F4 F4 F4 F4 F4 F4)
DEFI(p_F32, 32, F4 F4 F4 F4 F4 F4 F4 F4)
DEFI(p_TTT_F44, 47, T0 T1 "v_rcp_f32 %1,%1\n" F4 F4 F4 F4 F4 F4 F4 F4 F4 F4 F4)
DEFI(p_T_F14_x3, 45, T0 F0 F1 F4 F4 F4 T1 F0 F1 F4 F4 F4 "v_rcp_f32 %1,%1\n" F0 F2 F4 F4 F4)
DEFI(p_TT, 2, T0 T1)
DEFI(p_TFTF, 4, T0 F0 T1 F1)
DEFI(p_TSFF, 4, T0 S0 F0 F1)
DEFI(p_TFFFTFFF_S, 12, T0 F0 F1 S0 F0 F1 T1 F0 S2 F1 F0 F1)
DEFI(p_T_nop_F, 3, T0 "s_nop 0\n" F0)
DEFI(p_FFFF_dep, 4, "v_mul_f32 %0,%0,%1\n v_mul_f32 %0,%0,%1\n v_mul_f32 %0,%0,%1\n v_mul_f32 %0,%0,%1\n")
DEFI(p_SSSS_dep, 4, "v_min_f32 %0,%0,%1\n v_min_f32 %0,%0,%1\n v_min_f32 %0,%0,%1\n v_min_f32 %0,%0,%1\n")
DEFI(p_F_salu, 4, F0 "s_and_b64 %8,%8,%9\n" F1 "s_or_b64 %9,%9,%8\n")
DEFI(p_S_salu, 4, S1 "s_and_b64 %8,%8,%9\n" S2 "s_or_b64 %9,%9,%8\n")

void run(const char* name, void (*kern)(float*), int ni, float* sink, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bodies_per_simd = (double)waves_per_simd * OUTER * 32;
    printf("%-16s %d waves/SIMD: %6.2f cycles per body of %2d  (%.2f per instruction) @2.4GHz\n", name, waves_per_simd,
           ms * 1e-3 * 2.4e9 / bodies_per_simd, ni, ms * 1e-3 * 2.4e9 / bodies_per_simd / ni);
}
#define R(K) run(#K, K, K##_n, sink, w)
int main() {
    float* sink; (void)hipMalloc(&sink, 1024 * 16 * 64 * 4);
    for (int w : {6, 2}) {
        R(p_FFFF); R(p_SSSS); R(p_SFSF); R(p_SSFF); R(p_SFFF); R(p_SSSF); R(p_SFF); R(p_SF); R(p_align4); R(p_align_dep_F);
        R(p_TFFF); R(p_TFFFFFFF); R(p_T_F15); R(p_T_F31); R(p_F32); R(p_TTT_F44); R(p_T_F14_x3); R(p_TT); R(p_TFTF); R(p_TSFF); R(p_TFFFTFFF_S); R(p_T_nop_F); R(p_FFFF_dep); R(p_SSSS_dep); R(p_F_salu); R(p_S_salu);
    }
    return 0;
}
