// Micro-benchmark (round 5): does a transcendental / matrix / cross-lane instruction take the double rate away from the OTHER waves of its SIMD too?
// Waves of two kinds in one launch (6 per SIMD): "clean" waves run 32 double-rate instructions per body, "dirty" waves the same with one suspect per body
// (or per 8 instructions).  Wall time per body for all-clean, all-dirty and the 50 / 50 mix: if the loss stayed inside the dirty waves, the mix would sit
// half-way between the two; if it hit the whole SIMD, the mix would be as slow as all-dirty.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OUTER 64
#define F4 "v_mul_f32 %0,%0,%1\n v_fma_f32 %4,%4,%5,%5\n v_mul_f32 %2,%2,%3\n v_fma_f32 %6,%6,%7,%7\n"
#define F8 F4 F4
#define F32 F8 F8 F8 F8
#define ASM(B) asm volatile(".rept 16\n" B "\n.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "vcc", "scc", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83")
template <int SUSPECT>
__global__ void __launch_bounds__(256) k_mix(float* sink, int dirty_mod) {       // dirty_mod: 0 = nobody, 1 = everybody, 2 = every second wave of every SIMD
    float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f, e = 1.1f, f = 0.9f, g = 1.3f, h = 0.8f;
    // (workgroups of FOUR waves, one per SIMD of a compute unit; which of them are dirty alternates with the workgroup's number on its XCD, so that every SIMD
    // gets both kinds — blockIdx & 1 alone would put all dirty waves on the odd XCDs)
    const bool dirty = dirty_mod == 1 || (dirty_mod == 2 && (((threadIdx.x >> 6) + (blockIdx.x >> 3) + (blockIdx.x >> 8)) & 1));
    if (!dirty) {
#pragma unroll 1
        for (int it = 0; it < OUTER; ++it) ASM(F32);
    } else {
#pragma unroll 1
        for (int it = 0; it < OUTER; ++it) {
            if constexpr (SUSPECT == 0) ASM("v_rcp_f32 %1,%1\n" F32);
            else if constexpr (SUSPECT == 1) ASM("v_mfma_f32_32x32x2_f32 v[68:83], %0, %1, v[68:83]\n" F32);
            else if constexpr (SUSPECT == 2) ASM("v_permlane32_swap_b32 %5, %6\n" F32);
            else if constexpr (SUSPECT == 3) ASM("v_rcp_f32 %1,%1\n" F8 "v_sqrt_f32 %3,%3\n" F8 "v_rcp_f32 %5,%5\n" F8 "v_rcp_f32 %7,%7\n" F8);      // one per 8
            else if constexpr (SUSPECT == 5) ASM("v_rcp_f32 %1,%1\n s_nop 0\n" F32);                                   // … followed by a scalar instruction, as in the kernels
            else if constexpr (SUSPECT == 6) ASM("v_sqrt_f32 %1,%1\n v_rcp_f32 %3,%3\n s_nop 0\n" F32 F8 F8 F8);            // the pair loop's ratio: 2 per 56
            else if constexpr (SUSPECT == 7) ASM("v_mfma_f32_32x32x2_f32 v[68:83], %0, %1, v[68:83]\n s_nop 7\n" F32);
            else if constexpr (SUSPECT == 8) ASM("v_mfma_f32_32x32x16_f16 v[68:83], v[60:63], v[64:67], v[68:83]\n s_nop 0\n" F32);   // the f16-input form: 8 passes
            else if constexpr (SUSPECT == 9) ASM("v_mfma_f32_32x32x16_f16 v[68:83], v[60:63], v[64:67], v[68:83]\n v_mfma_f32_32x32x16_f16 v[68:83], v[60:63], v[64:67], v[68:83]\n s_nop 0\n" F32);
            else if constexpr (SUSPECT == 10) ASM("v_mfma_f32_32x32x2_f32 v[68:83], %0, %1, v[68:83]\n v_mfma_f32_32x32x2_f32 v[68:83], %0, %1, v[68:83]\n s_nop 0\n" F32);
            else ASM("v_alignbit_b32 v60,v60,v61,31\n v_alignbit_b32 v60,v60,v62,31\n" F8 "v_alignbit_b32 v60,v60,v61,31\n v_alignbit_b32 v60,v60,v62,31\n" F8 "v_alignbit_b32 v60,v60,v61,31\n v_alignbit_b32 v60,v60,v62,31\n" F8 "v_alignbit_b32 v60,v60,v61,31\n v_alignbit_b32 v60,v60,v62,31\n" F8);
        }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <class K> double ms_of(K launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const int W = 6, blocks = 1024 * W / 4;
    float* sink; (void)hipMalloc(&sink, blocks * 256 * 4);
    auto run = [&](const char* name, auto kern) {
        double t[3];
        for (int m = 0; m < 3; ++m) t[m] = ms_of([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, sink, m); }) * 1e-3 * 2.4e9 / ((double)W * OUTER * 16);
        printf("%-28s all clean %6.1f   all dirty %6.1f   every second wave dirty %6.1f   (half-way: %6.1f) cycles per body of 32 double-rate instructions @2.4GHz\n",
               name, t[0], t[1], t[2], 0.5 * (t[0] + t[1]));
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("v_rcp_f32, 1 per 32", k_mix<0>); run("v_mfma_f32_32x32x2, 1 per 32", k_mix<1>); run("v_permlane32_swap, 1 per 32", k_mix<2>);
        run("transcendental, 1 per 8", k_mix<3>); run("v_rcp + s_nop, 1 per 32", k_mix<5>); run("sqrt, rcp, s_nop per 56", k_mix<6>); run("v_mfma + s_nop 7, 1 per 32", k_mix<7>); run("f16 mfma 32x32x16, 1 per 32", k_mix<8>); run("f16 mfma 32x32x16, 2 per 32", k_mix<9>); run("f32 mfma 32x32x2, 2 per 32", k_mix<10>); run("2 v_alignbit per 8 (control)", k_mix<4>);
    }
    return 0;
}
