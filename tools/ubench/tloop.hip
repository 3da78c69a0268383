// Micro-benchmark of the phase-1 inner loop shapes: SIMD cycles per target iteration (2 candidate chunks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096
template <int V> __global__ void __launch_bounds__(64) k(unsigned long long* out, float* sink, const float4* tq) {
    const int lane = threadIdx.x;
    float c0x = lane * 0.01f, c0y = lane * 0.02f, c0z = lane * 0.03f, c0c = 0.5f;
    float c1x = lane * 0.011f, c1y = lane * 0.021f, c1z = lane * 0.031f, c1c = 0.6f;
    float m2x = lane * 0.1f, m2y = lane * 0.2f, m2z = 0.3f * lane, thr = 0.7f;
    int w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (V == 0) {            // as in the kernel: 4 readlane, 6 fma, 2 cmp, 4 writelane via m0
        asm volatile(
            "s_mov_b32 s20, 0\n"
            "1:\n"
            "s_and_b32 s21, s20, 63\n"
            "v_readlane_b32 s8, %4, s21\n v_readlane_b32 s9, %5, s21\n v_readlane_b32 s10, %6, s21\n v_readlane_b32 s11, %7, s21\n"
            "v_fma_f32 v40, s8, %8, %11\n v_fma_f32 v41, s8, %12, %15\n"
            "v_fmac_f32 v40, s9, %9\n v_fmac_f32 v41, s9, %13\n"
            "v_fmac_f32 v40, s10, %10\n v_fmac_f32 v41, s10, %14\n"
            "v_cmp_ge_f32 vcc, s11, v40\n v_cmp_ge_f32 s[12:13], s11, v41\n"
            "s_mov_b32 s22, m0\n s_mov_b32 m0, s21\n s_nop 0\n"
            "v_writelane_b32 %0, vcc_lo, m0\n v_writelane_b32 %1, vcc_hi, m0\n v_writelane_b32 %2, s12, m0\n v_writelane_b32 %3, s13, m0\n"
            "s_mov_b32 m0, s22\n"
            "s_add_i32 s20, s20, 1\n s_cmp_lt_u32 s20, %16\n s_cbranch_scc1 1b\n"
            : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)
            : "v"(m2x), "v"(m2y), "v"(m2z), "v"(thr), "v"(c0x), "v"(c0y), "v"(c0z), "v"(c0c), "v"(c1x), "v"(c1y), "v"(c1z), "v"(c1c), "s"(ITERS)
            : "s8", "s9", "s10", "s11", "s12", "s13", "s20", "s21", "s22", "vcc", "v40", "v41");
    } else if constexpr (V == 1) {     // no writelanes (cmp results or-ed on SALU)
        asm volatile(
            "s_mov_b32 s20, 0\n s_mov_b64 s[14:15], 0\n"
            "1:\n"
            "s_and_b32 s21, s20, 63\n"
            "v_readlane_b32 s8, %4, s21\n v_readlane_b32 s9, %5, s21\n v_readlane_b32 s10, %6, s21\n v_readlane_b32 s11, %7, s21\n"
            "v_fma_f32 v40, s8, %8, %11\n v_fma_f32 v41, s8, %12, %15\n"
            "v_fmac_f32 v40, s9, %9\n v_fmac_f32 v41, s9, %13\n"
            "v_fmac_f32 v40, s10, %10\n v_fmac_f32 v41, s10, %14\n"
            "v_cmp_ge_f32 vcc, s11, v40\n v_cmp_ge_f32 s[12:13], s11, v41\n"
            "s_or_b64 s[14:15], s[14:15], vcc\n s_or_b64 s[14:15], s[14:15], s[12:13]\n"
            "s_add_i32 s20, s20, 1\n s_cmp_lt_u32 s20, %16\n s_cbranch_scc1 1b\n"
            "v_mov_b32 %0, s14\n v_mov_b32 %1, s15\n"
            : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)
            : "v"(m2x), "v"(m2y), "v"(m2z), "v"(thr), "v"(c0x), "v"(c0y), "v"(c0z), "v"(c0c), "v"(c1x), "v"(c1y), "v"(c1z), "v"(c1c), "s"(ITERS)
            : "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s20", "s21", "vcc", "v40", "v41");
    } else if constexpr (V == 2) {     // target data by scalar loads (s_load_dwordx4), writelanes kept
        asm volatile(
            "s_mov_b32 s20, 0\n"
            "1:\n"
            "s_and_b32 s21, s20, 63\n s_lshl_b32 s23, s21, 4\n"
            "s_load_dwordx4 s[8:11], %17, s23\n s_waitcnt lgkmcnt(0)\n"
            "v_fma_f32 v40, s8, %8, %11\n v_fma_f32 v41, s8, %12, %15\n"
            "v_fmac_f32 v40, s9, %9\n v_fmac_f32 v41, s9, %13\n"
            "v_fmac_f32 v40, s10, %10\n v_fmac_f32 v41, s10, %14\n"
            "v_cmp_ge_f32 vcc, s11, v40\n v_cmp_ge_f32 s[12:13], s11, v41\n"
            "s_mov_b32 s22, m0\n s_mov_b32 m0, s21\n s_nop 0\n"
            "v_writelane_b32 %0, vcc_lo, m0\n v_writelane_b32 %1, vcc_hi, m0\n v_writelane_b32 %2, s12, m0\n v_writelane_b32 %3, s13, m0\n"
            "s_mov_b32 m0, s22\n"
            "s_add_i32 s20, s20, 1\n s_cmp_lt_u32 s20, %16\n s_cbranch_scc1 1b\n"
            : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)
            : "v"(m2x), "v"(m2y), "v"(m2z), "v"(thr), "v"(c0x), "v"(c0y), "v"(c0z), "v"(c0c), "v"(c1x), "v"(c1y), "v"(c1z), "v"(c1c), "s"(ITERS), "s"(tq)
            : "s8", "s9", "s10", "s11", "s12", "s13", "s20", "s21", "s22", "s23", "vcc", "v40", "v41");
    } else if constexpr (V == 3) {     // only the fma+cmp core (targets fixed in SGPRs)
        asm volatile(
            "s_mov_b32 s20, 0\n s_mov_b64 s[14:15], 0\n v_readlane_b32 s8, %4, 3\n v_readlane_b32 s9, %5, 3\n v_readlane_b32 s10, %6, 3\n v_readlane_b32 s11, %7, 3\n"
            "1:\n"
            "v_fma_f32 v40, s8, %8, %11\n v_fma_f32 v41, s8, %12, %15\n"
            "v_fmac_f32 v40, s9, %9\n v_fmac_f32 v41, s9, %13\n"
            "v_fmac_f32 v40, s10, %10\n v_fmac_f32 v41, s10, %14\n"
            "v_cmp_ge_f32 vcc, s11, v40\n v_cmp_ge_f32 s[12:13], s11, v41\n"
            "s_or_b64 s[14:15], s[14:15], vcc\n s_or_b64 s[14:15], s[14:15], s[12:13]\n"
            "s_add_i32 s20, s20, 1\n s_cmp_lt_u32 s20, %16\n s_cbranch_scc1 1b\n"
            "v_mov_b32 %0, s14\n v_mov_b32 %1, s15\n"
            : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)
            : "v"(m2x), "v"(m2y), "v"(m2z), "v"(thr), "v"(c0x), "v"(c0y), "v"(c0z), "v"(c0c), "v"(c1x), "v"(c1y), "v"(c1z), "v"(c1c), "s"(ITERS)
            : "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s20", "vcc", "v40", "v41");
    } else if constexpr (V == 4) {     // as V0 but two targets per iteration (independent chains interleaved)
        asm volatile(
            "s_mov_b32 s20, 0\n"
            "1:\n"
            "s_and_b32 s21, s20, 63\n s_add_i32 s24, s21, 1\n s_and_b32 s24, s24, 63\n"
            "v_readlane_b32 s8, %4, s21\n v_readlane_b32 s9, %5, s21\n v_readlane_b32 s10, %6, s21\n v_readlane_b32 s11, %7, s21\n"
            "v_readlane_b32 s16, %4, s24\n v_readlane_b32 s17, %5, s24\n v_readlane_b32 s18, %6, s24\n v_readlane_b32 s19, %7, s24\n"
            "v_fma_f32 v40, s8, %8, %11\n v_fma_f32 v41, s8, %12, %15\n v_fma_f32 v42, s16, %8, %11\n v_fma_f32 v43, s16, %12, %15\n"
            "v_fmac_f32 v40, s9, %9\n v_fmac_f32 v41, s9, %13\n v_fmac_f32 v42, s17, %9\n v_fmac_f32 v43, s17, %13\n"
            "v_fmac_f32 v40, s10, %10\n v_fmac_f32 v41, s10, %14\n v_fmac_f32 v42, s18, %10\n v_fmac_f32 v43, s18, %14\n"
            "v_cmp_ge_f32 vcc, s11, v40\n v_cmp_ge_f32 s[12:13], s11, v41\n v_cmp_ge_f32 s[26:27], s19, v42\n v_cmp_ge_f32 s[28:29], s19, v43\n"
            "s_mov_b32 s22, m0\n s_mov_b32 m0, s21\n s_nop 0\n"
            "v_writelane_b32 %0, vcc_lo, m0\n v_writelane_b32 %1, vcc_hi, m0\n v_writelane_b32 %2, s12, m0\n v_writelane_b32 %3, s13, m0\n"
            "s_mov_b32 m0, s24\n s_nop 0\n"
            "v_writelane_b32 %0, s26, m0\n v_writelane_b32 %1, s27, m0\n v_writelane_b32 %2, s28, m0\n v_writelane_b32 %3, s29, m0\n"
            "s_mov_b32 m0, s22\n"
            "s_add_i32 s20, s20, 2\n s_cmp_lt_u32 s20, %16\n s_cbranch_scc1 1b\n"
            : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)
            : "v"(m2x), "v"(m2y), "v"(m2z), "v"(thr), "v"(c0x), "v"(c0y), "v"(c0z), "v"(c0c), "v"(c1x), "v"(c1y), "v"(c1z), "v"(c1c), "s"(ITERS)
            : "s8", "s9", "s10", "s11", "s12", "s13", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s24", "s26", "s27", "s28", "s29", "vcc", "v40", "v41", "v42", "v43");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = w0 + w1 + w2 + w3;
}

template <int V> void run(const char* name, int blocks, const float4* tq) {
    unsigned long long* d; float* s;
    hipMalloc(&d, blocks * 8); hipMalloc(&s, blocks * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, s, tq);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, s, tq);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += v;
    double wps = blocks / 1024.0;
    // wall-clock based SIMD cycles per target-iteration assuming 2.4 GHz
    printf("%-34s waves/SIMD=%4.1f  ticks/iter/wave = %7.2f   wall: %.3f ms -> SIMD cycles/iter @2.4GHz = %.2f\n", name, wps,
           sum / blocks / ITERS, ms, ms * 1e-3 * 2.4e9 / (ITERS * (wps < 1 ? 1 : wps)));
    hipFree(d); hipFree(s);
}

int main() {
    float4* tq; hipMalloc(&tq, 64 * 16); hipMemset(tq, 0, 64 * 16);
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<0>("V0 readlane+fma+cmp+writelane", blocks, tq);
        run<1>("V1 no writelane", blocks, tq);
        run<2>("V2 s_load targets + writelane", blocks, tq);
        run<3>("V3 fma+cmp core only", blocks, tq);
        run<4>("V4 two targets / iteration", blocks, tq);
    }
    return 0;
}
