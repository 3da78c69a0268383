// Micro-benchmark: does a wave's DEPENDENT chain of vector-ALU instructions issue as fast as independent instructions?
// cycles per wave64 instruction per SIMD at 1, 2, 3, 4, 5, 6, 8 waves per SIMD for a fully dependent chain, two and four
// interleaved chains (v_fma_f32, v_mul_f32, v_add_f32), and a chain through a transcendental.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/dep_chain tools/ubench/dep_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define OUTER 64
#define DEFK(NAME, BODY)                                                                                  \
    __global__ void __launch_bounds__(64) NAME(float* sink) {                                             \
        float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f, e = 0.999f, f = 0.001f;     \
        _Pragma("unroll 1") for (int it = 0; it < OUTER; ++it) {                                          \
            asm volatile(".rept 32\n" BODY "\n.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));   \
        }                                                                                                 \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;                                              \
    }
DEFK(k_fma_dep1, "v_fma_f32 %0,%0,%4,%5\n v_fma_f32 %0,%0,%4,%5\n v_fma_f32 %0,%0,%4,%5\n v_fma_f32 %0,%0,%4,%5")
DEFK(k_fma_dep2, "v_fma_f32 %0,%0,%4,%5\n v_fma_f32 %1,%1,%4,%5\n v_fma_f32 %0,%0,%4,%5\n v_fma_f32 %1,%1,%4,%5")
DEFK(k_fma_dep4, "v_fma_f32 %0,%0,%4,%5\n v_fma_f32 %1,%1,%4,%5\n v_fma_f32 %2,%2,%4,%5\n v_fma_f32 %3,%3,%4,%5")
DEFK(k_mul_dep1, "v_mul_f32 %0,%0,%4\n v_mul_f32 %0,%0,%4\n v_mul_f32 %0,%0,%4\n v_mul_f32 %0,%0,%4")
DEFK(k_mul_dep2, "v_mul_f32 %0,%0,%4\n v_mul_f32 %1,%1,%4\n v_mul_f32 %0,%0,%4\n v_mul_f32 %1,%1,%4")
DEFK(k_mul_dep4, "v_mul_f32 %0,%0,%4\n v_mul_f32 %1,%1,%4\n v_mul_f32 %2,%2,%4\n v_mul_f32 %3,%3,%4")
DEFK(k_add_dep1, "v_add_f32 %0,%0,%5\n v_add_f32 %0,%0,%5\n v_add_f32 %0,%0,%5\n v_add_f32 %0,%0,%5")
DEFK(k_add_dep4, "v_add_f32 %0,%0,%5\n v_add_f32 %1,%1,%5\n v_add_f32 %2,%2,%5\n v_add_f32 %3,%3,%5")
DEFK(k_mix_dep1, "v_mul_f32 %0,%0,%4\n v_fma_f32 %0,%0,%4,%5\n v_sub_f32 %0,%0,%5\n v_fma_f32 %0,%0,%4,%5")
DEFK(k_mix_dep4, "v_mul_f32 %0,%0,%4\n v_fma_f32 %1,%1,%4,%5\n v_sub_f32 %2,%2,%5\n v_fma_f32 %3,%3,%4,%5")
DEFK(k_rcp_dep1, "v_rcp_f32 %0,%0\n s_nop 0\n v_fma_f32 %0,%0,%4,%5\n v_rcp_f32 %0,%0\n s_nop 0\n v_fma_f32 %0,%0,%4,%5")
DEFK(k_rcp_dep2, "v_rcp_f32 %0,%0\n v_rcp_f32 %1,%1\n v_fma_f32 %0,%0,%4,%5\n v_fma_f32 %1,%1,%4,%5")
void run(const char* name, void (*kern)(float*), float* sink, int w, int per_body) {
    const int blocks = 1024 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)w * OUTER * 32 * per_body;
    printf("%-12s %d waves/SIMD: %7.3f ms  %5.2f cycles per instruction per SIMD @2.4 GHz (one wave alone: %5.2f per instruction)\n", name, w, ms,
           ms * 1e-3 * 2.4e9 / inst_per_simd, ms * 1e-3 * 2.4e9 / (OUTER * 32.0 * per_body));
}
#define R(K, N) run(#K, K, sink, w, N)
int main() {
    float* sink; hipMalloc(&sink, 1024 * 16 * 64 * 4);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        R(k_fma_dep1, 4); R(k_fma_dep2, 4); R(k_fma_dep4, 4); R(k_mul_dep1, 4); R(k_mul_dep2, 4); R(k_mul_dep4, 4); R(k_add_dep1, 4); R(k_add_dep4, 4);
        R(k_mix_dep1, 4); R(k_mix_dep4, 4); R(k_rcp_dep1, 4); R(k_rcp_dep2, 4);
    }
    return 0;
}
