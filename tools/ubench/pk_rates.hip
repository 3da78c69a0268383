// Micro-benchmark (round 5): what the packed fp32 instructions of the pair loop cost next to the scalar ones they replace — plain,
// with the neg / op_sel modifiers the loop uses, and interleaved with double-rate instructions (v_mul / v_add) as in the loop.
// Cycles per wave64 instruction per SIMD at 6 waves per SIMD (the occupancy of the force kernels) and 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OUTER 64
#define DEFK(NAME, NI, BODY)                                                                              \
    __global__ void __launch_bounds__(64) NAME(float* sink) {                                             \
        float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f;                              \
        double da = a, db = b, dc = c, dd = d;                                                            \
        _Pragma("unroll 1") for (int it = 0; it < OUTER; ++it) {                                          \
            asm volatile(".rept 32\n" BODY "\n.endr"                                                      \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(da), "+v"(db), "+v"(dc), "+v"(dd) :: "vcc", "scc"); \
        }                                                                                                 \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + (float)(da + db + dc + dd);                 \
    }                                                                                                     \
    static const int NAME##_n = NI;
// %0-%3 float, %4-%7 vgpr pairs
DEFK(k_fma, 4,       "v_fma_f32 %0,%0,%1,%1\n v_fma_f32 %1,%1,%2,%2\n v_fma_f32 %2,%2,%3,%3\n v_fma_f32 %3,%3,%0,%0")
DEFK(k_mul, 4,       "v_mul_f32 %0,%0,%1\n v_mul_f32 %1,%1,%2\n v_mul_f32 %2,%2,%3\n v_mul_f32 %3,%3,%0")
DEFK(k_sub, 4,       "v_sub_f32 %0,%0,%1\n v_sub_f32 %1,%1,%2\n v_sub_f32 %2,%2,%3\n v_sub_f32 %3,%3,%0")
DEFK(k_pkfma, 4,     "v_pk_fma_f32 %4,%4,%5,%5\n v_pk_fma_f32 %5,%5,%6,%6\n v_pk_fma_f32 %6,%6,%7,%7\n v_pk_fma_f32 %7,%7,%4,%4")
DEFK(k_pkfma_sel, 4, "v_pk_fma_f32 %4,%4,%5,%5 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %5,%5,%6,%6 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %6,%6,%7,%7 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %7,%7,%4,%4 op_sel_hi:[0,1,1]")
DEFK(k_pkmul, 4,     "v_pk_mul_f32 %4,%4,%5\n v_pk_mul_f32 %5,%5,%6\n v_pk_mul_f32 %6,%6,%7\n v_pk_mul_f32 %7,%7,%4")
DEFK(k_pkadd, 4,     "v_pk_add_f32 %4,%4,%5\n v_pk_add_f32 %5,%5,%6\n v_pk_add_f32 %6,%6,%7\n v_pk_add_f32 %7,%7,%4")
DEFK(k_pkadd_neg, 4, "v_pk_add_f32 %4,%4,%5 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5,%5,%6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %6,%6,%7 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %7,%7,%4 neg_lo:[0,1] neg_hi:[0,1]")
// interleaved with double-rate instructions
DEFK(k_mul_fma, 4,   "v_mul_f32 %0,%0,%1\n v_fma_f32 %1,%1,%2,%2\n v_mul_f32 %2,%2,%3\n v_fma_f32 %3,%3,%0,%0")
DEFK(k_mul_pkfma, 4, "v_mul_f32 %0,%0,%1\n v_pk_fma_f32 %4,%4,%5,%5\n v_mul_f32 %2,%2,%3\n v_pk_fma_f32 %6,%6,%7,%7")
DEFK(k_mul_pkmul, 4, "v_mul_f32 %0,%0,%1\n v_pk_mul_f32 %4,%4,%5\n v_mul_f32 %2,%2,%3\n v_pk_mul_f32 %6,%6,%7")
DEFK(k_mul3_pk, 4,   "v_mul_f32 %0,%0,%1\n v_mul_f32 %1,%1,%2\n v_mul_f32 %2,%2,%3\n v_pk_fma_f32 %6,%6,%7,%7")
DEFK(k_mul3_fma, 4,  "v_mul_f32 %0,%0,%1\n v_mul_f32 %1,%1,%2\n v_mul_f32 %2,%2,%3\n v_fma_f32 %3,%3,%0,%0")
// the head of the pair loop, scalar (round 4: 6 sub, 2 mul, 4 fmac) and packed (round 5: 2 pk_add, 2 sub, 2 pk_mul, 2 add, 1 pk_fma)
DEFK(k_head_scalar, 12, "v_sub_f32 %0,%0,%1\n v_sub_f32 %1,%1,%2\n v_mul_f32 %2,%0,%0\n v_sub_f32 %3,%3,%0\n v_fmac_f32 %2,%1,%1\n v_fmac_f32 %2,%3,%3\n"
                        "v_sub_f32 %0,%0,%1\n v_sub_f32 %1,%1,%2\n v_mul_f32 %2,%0,%0\n v_sub_f32 %3,%3,%0\n v_fmac_f32 %2,%1,%1\n v_fmac_f32 %2,%3,%3")
DEFK(k_head_packed, 9,  "v_pk_add_f32 %4,%4,%5 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %5,%5,%6 neg_lo:[0,1] neg_hi:[0,1]\n v_sub_f32 %0,%0,%1\n v_pk_mul_f32 %6,%4,%4\n v_pk_mul_f32 %7,%4,%5\n"
                        "v_sub_f32 %1,%1,%2\n v_add_f32 %2,%2,%3\n v_add_f32 %3,%3,%0\n v_pk_fma_f32 %7,%4,%4,%6 op_sel_hi:[0,1,1]")

// the other instruction kinds of the pair loop and of phase 1 (re-measured on this hardware: profiles/HISTORY.md §4.3's table had v_fma at 4.4)
#define DEFI(NAME, NI, BODY)                                                                              \
    __global__ void __launch_bounds__(64) NAME(float* sink) {                                             \
        float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f;                              \
        unsigned ia = threadIdx.x, ib = 3, ic = 5, id = 7;                                                \
        unsigned long long s0 = 1, s1 = 2;                                                                \
        _Pragma("unroll 1") for (int it = 0; it < OUTER; ++it) {                                          \
            asm volatile(".rept 32\n" BODY "\n.endr"                                                      \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id), "+s"(s0), "+s"(s1) :: "vcc", "scc"); \
        }                                                                                                 \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + ia + ib + ic + id + (float)(s0 + s1);       \
    }                                                                                                     \
    static const int NAME##_n = NI;
// %0-%3 float, %4-%7 uint, %8,%9 sgpr pairs
DEFI(i_fmac, 4,     "v_fmac_f32 %0,%1,%2\n v_fmac_f32 %1,%2,%3\n v_fmac_f32 %2,%3,%0\n v_fmac_f32 %3,%0,%1")
DEFI(i_fma_clamp, 4,"v_fma_f32 %0,%0,%1,1.0 clamp\n v_fma_f32 %1,%1,%2,1.0 clamp\n v_fma_f32 %2,%2,%3,1.0 clamp\n v_fma_f32 %3,%3,%0,1.0 clamp")
DEFI(i_fma_abs, 4,  "v_fma_f32 %0,|%0|,%1,-%1\n v_fma_f32 %1,|%1|,%2,-%2\n v_fma_f32 %2,|%2|,%3,-%3\n v_fma_f32 %3,|%3|,%0,-%0")
DEFI(i_add_abs, 4,  "v_add_f32_e64 %0,|%0|,|%1|\n v_add_f32_e64 %1,|%1|,|%2|\n v_sub_f32_e64 %2,|%2|,|%3|\n v_sub_f32_e64 %3,|%3|,|%0|")
DEFI(i_mul_clamp, 4,"v_mul_f32_e64 %0,%0,%1 clamp\n v_mul_f32_e64 %1,%1,%2 clamp\n v_mul_f32_e64 %2,%2,%3 clamp\n v_mul_f32_e64 %3,%3,%0 clamp")
DEFI(i_minmax, 4,   "v_max_f32 %0,%0,%1\n v_min_f32 %1,%1,%2\n v_max_f32 %2,%2,%3\n v_min_f32 %3,%3,%0")
DEFI(i_mov, 4,      "v_mov_b32 %0,%1\n v_mov_b32 %1,%2\n v_mov_b32 %2,%3\n v_mov_b32 %3,%0")
DEFI(i_cnd_vcc, 4,  "v_cndmask_b32 %0,%0,%1,vcc\n v_cndmask_b32 %1,%1,%2,vcc\n v_cndmask_b32 %2,%2,%3,vcc\n v_cndmask_b32 %3,%3,%0,vcc")
DEFI(i_cmpf_vcc, 4, "v_cmp_lt_f32 vcc,%0,%1\n v_cmp_lt_f32 vcc,%1,%2\n v_cmp_lt_f32 vcc,%2,%3\n v_cmp_lt_f32 vcc,%3,%0")
DEFI(i_cmpu_vcc, 4, "v_cmp_lt_u32 vcc,%4,%5\n v_cmp_lt_u32 vcc,%5,%6\n v_cmp_lt_u32 vcc,%6,%7\n v_cmp_lt_u32 vcc,%7,%4")
DEFI(i_cmpu_sgpr, 4,"v_cmp_lt_u32 %8,%4,%5\n v_cmp_lt_u32 %9,%5,%6\n v_cmp_lt_u32 %8,%6,%7\n v_cmp_lt_u32 %9,%7,%4")
DEFI(i_addu, 4,     "v_add_u32 %4,%4,%5\n v_sub_u32 %5,%5,%6\n v_add_u32 %6,%6,%7\n v_sub_u32 %7,%7,%4")
DEFI(i_andor, 4,    "v_and_b32 %4,%4,%5\n v_or_b32 %5,%5,%6\n v_and_b32 %6,%6,%7\n v_or_b32 %7,%7,%4")
DEFI(i_addco, 4,    "v_add_co_u32 %4,%8,-1,%4\n v_add_co_u32 %5,%9,-1,%5\n v_add_co_u32 %6,%8,-1,%6\n v_add_co_u32 %7,%9,-1,%7")
DEFI(i_ffbl, 4,     "v_ffbl_b32 %4,%5\n v_ffbl_b32 %5,%6\n v_ffbl_b32 %6,%7\n v_ffbl_b32 %7,%4")
DEFI(i_lshladd, 4,  "v_lshl_add_u32 %4,%4,5,%5\n v_lshl_add_u32 %5,%5,5,%6\n v_lshl_add_u32 %6,%6,5,%7\n v_lshl_add_u32 %7,%7,5,%4")
DEFI(i_lshl, 4,     "v_lshlrev_b32 %4,3,%4\n v_lshrrev_b32 %5,3,%5\n v_lshlrev_b32 %6,3,%6\n v_lshrrev_b32 %7,3,%7")
DEFI(i_alignbit, 4, "v_alignbit_b32 %4,%4,%5,31\n v_alignbit_b32 %5,%5,%6,31\n v_alignbit_b32 %6,%6,%7,31\n v_alignbit_b32 %7,%7,%4,31")
DEFI(i_bfi, 4,      "v_bfi_b32 %4,%4,%5,%6\n v_bfi_b32 %5,%5,%6,%7\n v_bfi_b32 %6,%6,%7,%4\n v_bfi_b32 %7,%7,%4,%5")
DEFI(i_andor3, 4,   "v_and_or_b32 %4,%4,%5,%6\n v_and_or_b32 %5,%5,%6,%7\n v_lshl_or_b32 %6,%6,1,%7\n v_lshl_or_b32 %7,%7,1,%4")
DEFI(i_rcp, 4,      "v_rcp_f32 %0,%0\n v_rcp_f32 %1,%1\n v_rcp_f32 %2,%2\n v_rcp_f32 %3,%3")
DEFI(i_sqrt, 4,     "v_sqrt_f32 %0,%0\n v_sqrt_f32 %1,%1\n v_sqrt_f32 %2,%2\n v_sqrt_f32 %3,%3")
DEFI(i_rcp_mul, 4,  "v_rcp_f32 %0,%0\n v_mul_f32 %1,%1,%2\n v_mul_f32 %2,%2,%3\n v_mul_f32 %3,%3,%1")
DEFI(i_cmp_mul, 4,  "v_cmp_lt_u32 vcc,%4,%5\n v_mul_f32 %1,%1,%2\n v_mul_f32 %2,%2,%3\n v_mul_f32 %3,%3,%1")
DEFI(i_cnd_mul, 4,  "v_cndmask_b32 %0,%0,%1,vcc\n v_mul_f32 %1,%1,%2\n v_mul_f32 %2,%2,%3\n v_mul_f32 %3,%3,%1")
DEFI(i_salu_mix, 4, "v_mul_f32 %0,%0,%1\n s_and_b64 %8,%8,%9\n v_mul_f32 %2,%2,%3\n s_or_b64 %9,%9,%8")

void run(const char* name, void (*kern)(float*), int ni, float* sink, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)waves_per_simd * OUTER * 32 * ni;
    printf("%-14s %d waves/SIMD: %.3f ms  %.2f cycles per instruction, %.1f per body of %d @2.4GHz\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / inst_per_simd, ms * 1e-3 * 2.4e9 / inst_per_simd * ni, ni);
}
#define R(K) run(#K, K, K##_n, sink, w)
int main() {
    float* sink; hipMalloc(&sink, 1024 * 16 * 64 * 4);
    for (int w : {6, 1}) {
        R(k_fma); R(k_mul); R(k_sub); R(k_pkfma); R(k_pkfma_sel); R(k_pkmul); R(k_pkadd); R(k_pkadd_neg);
        R(k_mul_fma); R(k_mul_pkfma); R(k_mul_pkmul); R(k_mul3_pk); R(k_mul3_fma); R(k_head_scalar); R(k_head_packed);
        R(i_fmac); R(i_fma_clamp); R(i_fma_abs); R(i_add_abs); R(i_mul_clamp); R(i_minmax); R(i_mov); R(i_cnd_vcc); R(i_cmpf_vcc); R(i_cmpu_vcc); R(i_cmpu_sgpr);
        R(i_addu); R(i_andor); R(i_addco); R(i_ffbl); R(i_lshladd); R(i_lshl); R(i_alignbit); R(i_bfi); R(i_andor3); R(i_rcp); R(i_sqrt); R(i_rcp_mul); R(i_cmp_mul); R(i_cnd_mul); R(i_salu_mix);
    }
    return 0;
}
