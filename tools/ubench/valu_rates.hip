// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the neighbour
// kernel leans on.  One wave per SIMD per CU pair is enough: we time with s_memtime inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 512
template <int OP> __global__ void __launch_bounds__(64) k(unsigned long long* out, float* sink, int waves_note) {
    float a = threadIdx.x * 0.5f, b = 1.0001f, c = 0.3f, d = 0.7f, e = 1.1f, f = 0.9f, g2 = 0.2f, h = 0.4f;
    int ia = threadIdx.x, ib = 3;
    unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
        if constexpr (OP == 0) {   // v_fma_f32 independent x8
            asm volatile(".rept 64\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         " v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n .endr"
                         : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g2), "+v"(h), "+v"(b) : "v"(1.0f), "v"(0.0f));
        } else if constexpr (OP == 1) {   // v_pk_fma_f32 x4 pairs
            asm volatile(".rept 64\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         " v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n .endr"
                         : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g2) : "v"(1.0), "v"(0.0));
        } else if constexpr (OP == 2) {   // v_readlane_b32 x8
            int s0, s1, s2, s3;
            asm volatile(".rept 64\n v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %5, 9\n"
                         " v_readlane_b32 %0, %4, 13\n v_readlane_b32 %1, %5, 15\n v_readlane_b32 %2, %4, 17\n v_readlane_b32 %3, %5, 19\n .endr"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(ia), "v"(ib));
            ib += s0 + s1 + s2 + s3;
        } else if constexpr (OP == 3) {   // v_writelane_b32 x8 (inline-constant lane select)
            asm volatile(".rept 64\n v_writelane_b32 %0, %2, 3\n v_writelane_b32 %1, %2, 5\n v_writelane_b32 %0, %2, 7\n v_writelane_b32 %1, %2, 9\n"
                         " v_writelane_b32 %0, %2, 13\n v_writelane_b32 %1, %2, 15\n v_writelane_b32 %0, %2, 17\n v_writelane_b32 %1, %2, 19\n .endr"
                         : "+v"(ia), "+v"(ib) : "s"(it));
        } else if constexpr (OP == 4) {   // v_cmp_ge_f32 to SGPR pair x8
            unsigned long long m0_, m1_, m2_, m3_;
            asm volatile(".rept 64\n v_cmp_ge_f32 %0, %4, %5\n v_cmp_ge_f32 %1, %5, %4\n v_cmp_ge_f32 %2, %4, %5\n v_cmp_ge_f32 %3, %5, %4\n"
                         " v_cmp_ge_f32 %0, %4, %5\n v_cmp_ge_f32 %1, %5, %4\n v_cmp_ge_f32 %2, %4, %5\n v_cmp_ge_f32 %3, %5, %4\n .endr"
                         : "=s"(m0_), "=s"(m1_), "=s"(m2_), "=s"(m3_) : "v"(a), "v"(b));
            ib += (int)(m0_ + m1_ + m2_ + m3_);
        } else if constexpr (OP == 5) {   // v_mov_b32 x8
            asm volatile(".rept 64\n v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5\n"
                         " v_mov_b32 %0, %5\n v_mov_b32 %1, %4\n v_mov_b32 %2, %5\n v_mov_b32 %3, %4\n .endr"
                         : "+v"(a), "+v"(c), "+v"(e), "+v"(g2) : "v"(b), "v"(d));
        } else if constexpr (OP == 6) {   // v_add_u32 x8
            asm volatile(".rept 64\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n"
                         " v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2\n .endr"
                         : "+v"(ia), "+v"(ib) : "v"(3));
        } else if constexpr (OP == 7) {   // v_rcp_f32 x8
            asm volatile(".rept 64\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         " v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n .endr"
                         : "+v"(a), "+v"(c), "+v"(e), "+v"(g2));
        } else if constexpr (OP == 8) {   // v_fma_f32 with SGPR operand x8
            float sa = 1.0f + it;
            asm volatile(".rept 64\n v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n"
                         " v_fma_f32 %0, %0, %4, %1\n v_fma_f32 %1, %1, %4, %2\n v_fma_f32 %2, %2, %4, %3\n v_fma_f32 %3, %3, %4, %0\n .endr"
                         : "+v"(a), "+v"(c), "+v"(e), "+v"(g2) : "s"(sa));
        } else if constexpr (OP == 9) {   // v_cndmask_b32 x8
            asm volatile(".rept 64\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n"
                         " v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n .endr"
                         : "+v"(a), "+v"(c) : "v"(b) : "vcc");
        } else if constexpr (OP == 10) {  // v_mul_f32 / v_sub_f32 mix x8
            asm volatile(".rept 64\n v_mul_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4\n"
                         " v_mul_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4\n .endr"
                         : "+v"(a), "+v"(c), "+v"(e), "+v"(g2) : "v"(b));
        } else if constexpr (OP == 11) {  // v_sqrt_f32
            asm volatile(".rept 64\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         " v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n .endr"
                         : "+v"(a), "+v"(c), "+v"(e), "+v"(g2));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g2 + h + ia + ib;
}

template <int OP> void run(const char* name, int blocks) {
    unsigned long long* d; float* s;
    hipMalloc(&d, blocks * 8); hipMalloc(&s, blocks * 64 * 4);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, s, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, s, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += v;
    // readcyclecounter = s_memtime (fixed 100 MHz?) or shader clock: report raw per-instruction ticks
    printf("%-28s blocks=%5d  ticks/inst = %.3f\n", name, blocks, sum / blocks / (64.0 * 64 * 8));
    hipFree(d); hipFree(s);
}

int main() {
    for (int blocks : {1024, 4096, 8192}) {   // 1 / 4 / 8 waves per CU → 0.25 / 1 / 2 waves per SIMD
        run<0>("v_fma_f32", blocks); run<8>("v_fma_f32 (sgpr src)", blocks); run<10>("v_mul/v_sub_f32", blocks);
        run<1>("v_pk_fma_f32", blocks); run<2>("v_readlane_b32", blocks); run<3>("v_writelane_b32", blocks);
        run<4>("v_cmp_ge_f32 -> sgpr", blocks); run<5>("v_mov_b32", blocks); run<6>("v_add_u32", blocks);
        run<9>("v_cndmask_b32", blocks); run<7>("v_rcp_f32", blocks); run<11>("v_sqrt_f32", blocks);
    }
    return 0;
}
