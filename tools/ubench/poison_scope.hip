// Micro-benchmark (round 5): is the loss of the double rate (tools/ubench/double_rate.hip) confined to the WAVE that issues a transcendental /
// matrix / cross-lane / packed instruction, or does it hit every wave of the SIMD?  And how long does it last?
// Part 1: every sixth wave runs a body with the suspect, the other five run 32 double-rate instructions per body; the shader-clock time of the
//         clean waves is compared with a launch without suspects.
// Part 2: one suspect followed by N double-rate instructions, N = 32 … 1024, all waves alike.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OUTER 64
#define F4 "v_mul_f32 %0,%0,%1\n v_fma_f32 %4,%4,%5,%5\n v_mul_f32 %2,%2,%3\n v_fma_f32 %6,%6,%7,%7\n"
#define F32 F4 F4 F4 F4 F4 F4 F4 F4
#define BODY(PRE, REPT, B) asm volatile(PRE "\n.rept " #REPT "\n" B "\n.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "vcc", "scc", "s20", "s21")

template <int SUSPECT>
__global__ void __launch_bounds__(64) k_scope(float* sink, long long* clk, int every) {
    float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f, e = 1.1f, f = 0.9f, g = 1.3f, h = 0.8f;
    const bool dirty = every > 0 && (blockIdx.x % every) == 0;
    const long long t0 = __builtin_readcyclecounter();
    if (!dirty) {
#pragma unroll 1
        for (int it = 0; it < OUTER; ++it) BODY("", 16, F32);
    } else {
#pragma unroll 1
        for (int it = 0; it < OUTER; ++it) {
            if constexpr (SUSPECT == 0) BODY("", 16, "v_rcp_f32 %1,%1\n" F32);
            else if constexpr (SUSPECT == 1) BODY("", 16, "v_pk_mul_f32 v[60:61], v[62:63], v[64:65]\n" F32);
            else if constexpr (SUSPECT == 2) BODY("", 16, "v_mfma_f32_32x32x2_f32 v[68:83], %0, %1, v[68:83]\n" F32);
            else BODY("", 16, "v_permlane32_swap_b32 %5, %6\n" F32);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int N32>
__global__ void __launch_bounds__(64) k_persist(float* sink) {
    float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f, e = 1.1f, f = 0.9f, g = 1.3f, h = 0.8f;
#pragma unroll 1
    for (int it = 0; it < OUTER * 32 / N32; ++it) {
        asm volatile("v_rcp_f32 %1,%1\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
#pragma unroll 1
        for (int k = 0; k < N32; ++k) BODY("", 1, F32);
    }
    sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

template <class K> double time_ms(K launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    const int W = 6, blocks = 1024 * W;
    float* sink; long long* clk; (void)hipMalloc(&sink, blocks * 64 * 4); (void)hipMalloc(&clk, blocks * 8);
    std::vector<long long> h(blocks);
    auto part1 = [&](const char* name, auto kern) {
        for (int every : {0, 6, 2}) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink, clk, every);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink, clk, every);
            (void)hipDeviceSynchronize(); (void)hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
            double clean = 0, dirty = 0; int nc = 0, nd = 0;
            for (int b = 0; b < blocks; ++b) { if (every > 0 && b % every == 0) { dirty += h[b]; ++nd; } else { clean += h[b]; ++nc; } }
            printf("%-10s suspect in every %d-th wave: clean waves %8.0f clock ticks per wave (%d waves), waves with the suspect %8.0f (%d)\n", name, every,
                   nc ? clean / nc : 0.0, nc, nd ? dirty / nd : 0.0, nd);
        }
    };
    part1("v_rcp", k_scope<0>); part1("v_pk_mul", k_scope<1>); part1("v_mfma", k_scope<2>); part1("permlane", k_scope<3>);
    auto part2 = [&](const char* name, auto kern, int n32) {
        const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink); });
        const double per32 = ms * 1e-3 * 2.4e9 / ((double)W * OUTER * 32);
        printf("one v_rcp_f32 per %5d double-rate instructions: %6.2f cycles per 32 of them (%s)\n", 32 * n32, per32, name);
    };
    part2("k_persist<1>", k_persist<1>, 1); part2("k_persist<2>", k_persist<2>, 2); part2("k_persist<4>", k_persist<4>, 4);
    part2("k_persist<8>", k_persist<8>, 8); part2("k_persist<16>", k_persist<16>, 16); part2("k_persist<32>", k_persist<32>, 32);
    return 0;
}
