// Micro-benchmark 2: wall-clock throughput (cycles per wave64 instruction per SIMD, relative to v_fma_f32 = its own
// measured figure) of the VALU / SALU instructions the neighbour kernel is made of.  8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OUTER 64
#define DEFK(NAME, BODY)                                                                                  \
    __global__ void __launch_bounds__(64) NAME(float* sink) {                                             \
        float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f;                              \
        unsigned ia = threadIdx.x, ib = 3, ic = 5, id = 7;                                                \
        double da = a, db = b, dc = c, dd = d;                                                            \
        unsigned long long s0 = 1, s1 = 2;                                                                \
        _Pragma("unroll 1") for (int it = 0; it < OUTER; ++it) {                                          \
            asm volatile(".rept 32\n" BODY "\n.endr"                                                      \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id),   \
                           "+v"(da), "+v"(db), "+v"(dc), "+v"(dd), "+s"(s0), "+s"(s1)                     \
                         :: "vcc", "scc");                                                               \
        }                                                                                                 \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + ia + ib + ic + id + (float)(da + db + dc + dd) + (float)(s0 + s1); \
    }
// %0-%3 float, %4-%7 uint, %8-%11 double (vgpr pairs), %12,%13 sgpr pairs.  Each body = 4 instructions.
DEFK(k_fma,      "v_fma_f32 %0,%0,%1,%1\n v_fma_f32 %1,%1,%2,%2\n v_fma_f32 %2,%2,%3,%3\n v_fma_f32 %3,%3,%0,%0")
DEFK(k_mul,      "v_mul_f32 %0,%0,%1\n v_mul_f32 %1,%1,%2\n v_mul_f32 %2,%2,%3\n v_mul_f32 %3,%3,%0")
DEFK(k_add,      "v_add_f32 %0,%0,%1\n v_add_f32 %1,%1,%2\n v_add_f32 %2,%2,%3\n v_add_f32 %3,%3,%0")
DEFK(k_max,      "v_max_f32 %0,%0,%1\n v_min_f32 %1,%1,%2\n v_max_f32 %2,%2,%3\n v_min_f32 %3,%3,%0")
DEFK(k_mov,      "v_mov_b32 %0,%1\n v_mov_b32 %1,%2\n v_mov_b32 %2,%3\n v_mov_b32 %3,%0")
DEFK(k_cnd_vcc,  "v_cndmask_b32 %0,%0,%1,vcc\n v_cndmask_b32 %1,%1,%2,vcc\n v_cndmask_b32 %2,%2,%3,vcc\n v_cndmask_b32 %3,%3,%0,vcc")
DEFK(k_cnd_sgpr, "v_cndmask_b32 %0,%0,%1,%12\n v_cndmask_b32 %1,%1,%2,%13\n v_cndmask_b32 %2,%2,%3,%12\n v_cndmask_b32 %3,%3,%0,%13")
DEFK(k_cmp_vcc,  "v_cmp_lt_f32 vcc,%0,%1\n v_cmp_lt_f32 vcc,%1,%2\n v_cmp_lt_f32 vcc,%2,%3\n v_cmp_lt_f32 vcc,%3,%0")
DEFK(k_cmp_sgpr, "v_cmp_lt_f32 %12,%0,%1\n v_cmp_lt_f32 %13,%1,%2\n v_cmp_lt_f32 %12,%2,%3\n v_cmp_lt_f32 %13,%3,%0")
DEFK(k_cmpu_vcc, "v_cmp_lt_u32 vcc,%4,%5\n v_cmp_lt_u32 vcc,%5,%6\n v_cmp_lt_u32 vcc,%6,%7\n v_cmp_lt_u32 vcc,%7,%4")
DEFK(k_addu,     "v_add_u32 %4,%4,%5\n v_add_u32 %5,%5,%6\n v_add_u32 %6,%6,%7\n v_add_u32 %7,%7,%4")
DEFK(k_and,      "v_and_b32 %4,%4,%5\n v_or_b32 %5,%5,%6\n v_and_b32 %6,%6,%7\n v_or_b32 %7,%7,%4")
DEFK(k_lshl,     "v_lshlrev_b32 %4,3,%4\n v_lshrrev_b32 %5,3,%5\n v_lshlrev_b32 %6,3,%6\n v_lshrrev_b32 %7,3,%7")
DEFK(k_lshl64,   "v_lshlrev_b64 %8,3,%8\n v_lshrrev_b64 %9,3,%9\n v_lshlrev_b64 %10,3,%10\n v_lshrrev_b64 %11,3,%11")
DEFK(k_addco,    "v_add_co_u32 %4,vcc,%4,%5\n v_addc_co_u32 %5,vcc,%5,%6,vcc\n v_add_co_u32 %6,vcc,%6,%7\n v_addc_co_u32 %7,vcc,%7,%4,vcc")
DEFK(k_ffbl,     "v_ffbl_b32 %4,%5\n v_ffbl_b32 %5,%6\n v_ffbl_b32 %6,%7\n v_ffbl_b32 %7,%4")
DEFK(k_alignbit, "v_alignbit_b32 %4,%4,%5,31\n v_alignbit_b32 %5,%5,%6,31\n v_alignbit_b32 %6,%6,%7,31\n v_alignbit_b32 %7,%7,%4,31")
DEFK(k_mad24,    "v_mad_u32_u24 %4,%4,%5,%6\n v_mad_u32_u24 %5,%5,%6,%7\n v_mad_u32_u24 %6,%6,%7,%4\n v_mad_u32_u24 %7,%7,%4,%5")
DEFK(k_mullo,    "v_mul_lo_u32 %4,%4,%5\n v_mul_lo_u32 %5,%5,%6\n v_mul_lo_u32 %6,%6,%7\n v_mul_lo_u32 %7,%7,%4")
DEFK(k_bfe,      "v_bfe_u32 %4,%4,3,5\n v_bfe_u32 %5,%5,3,5\n v_bfe_u32 %6,%6,3,5\n v_bfe_u32 %7,%7,3,5")
DEFK(k_add3,     "v_add3_u32 %4,%4,%5,%6\n v_lshl_add_u32 %5,%5,2,%6\n v_and_or_b32 %6,%6,%7,%4\n v_add3_u32 %7,%7,%4,%5")
DEFK(k_rcp,      "v_rcp_f32 %0,%0\n v_rcp_f32 %1,%1\n v_rcp_f32 %2,%2\n v_rcp_f32 %3,%3")
DEFK(k_sqrt,     "v_sqrt_f32 %0,%0\n v_rsq_f32 %1,%1\n v_sqrt_f32 %2,%2\n v_rsq_f32 %3,%3")
DEFK(k_pkfma,    "v_pk_fma_f32 %8,%8,%9,%9\n v_pk_fma_f32 %9,%9,%10,%10\n v_pk_fma_f32 %10,%10,%11,%11\n v_pk_fma_f32 %11,%11,%8,%8")
DEFK(k_pkmul,    "v_pk_mul_f32 %8,%8,%9\n v_pk_add_f32 %9,%9,%10\n v_pk_mul_f32 %10,%10,%11\n v_pk_add_f32 %11,%11,%8")
DEFK(k_fma64,    "v_fma_f64 %8,%8,%9,%9\n v_fma_f64 %9,%9,%10,%10\n v_fma_f64 %10,%10,%11,%11\n v_fma_f64 %11,%11,%8,%8")
DEFK(k_mul64,    "v_mul_f64 %8,%8,%9\n v_add_f64 %9,%9,%10\n v_mul_f64 %10,%10,%11\n v_add_f64 %11,%11,%8")
DEFK(k_rcp64,    "v_rcp_f64 %8,%8\n v_rcp_f64 %9,%9\n v_sqrt_f64 %10,%10\n v_rsq_f64 %11,%11")
DEFK(k_readlane, "v_readlane_b32 s20,%4,3\n v_readlane_b32 s21,%5,5\n v_readfirstlane_b32 s22,%6\n v_readfirstlane_b32 s23,%7")
DEFK(k_salu,     "s_add_u32 s20,s20,s21\n s_and_b32 s21,s21,s22\n s_lshl_b32 s22,s22,1\n s_or_b32 s23,s23,s20")
DEFK(k_salu64,   "s_and_b64 %12,%12,%13\n s_or_b64 %13,%13,%12\n s_andn2_b64 %12,%12,%13\n s_xor_b64 %13,%13,%12")
DEFK(k_mix_vs,   "v_fma_f32 %0,%0,%1,%1\n s_add_u32 s20,s20,s21\n v_fma_f32 %2,%2,%3,%3\n s_and_b32 s21,s21,s22")
DEFK(k_mbcnt,    "v_mbcnt_lo_u32_b32 %4,%5,%4\n v_mbcnt_hi_u32_b32 %5,%6,%5\n v_mbcnt_lo_u32_b32 %6,%7,%6\n v_mbcnt_hi_u32_b32 %7,%4,%7")
DEFK(k_fmac,     "v_fmac_f32 %0,%1,%2\n v_fmac_f32 %1,%2,%3\n v_fmac_f32 %2,%3,%0\n v_fmac_f32 %3,%0,%1")
DEFK(k_fma_sgpr, "v_fma_f32 %0,%0,s20,%1\n v_fma_f32 %1,%1,s21,%2\n v_fma_f32 %2,%2,s22,%3\n v_fma_f32 %3,%3,s23,%0")
DEFK(k_fma_abs,  "v_fma_f32 %0,|%0|,%1,-%1\n v_fma_f32 %1,|%1|,%2,-%2\n v_fma_f32 %2,|%2|,%3,-%3\n v_fma_f32 %3,|%3|,%0,-%0")
DEFK(k_cvt,      "v_cvt_f32_u32 %0,%4\n v_cvt_u32_f32 %5,%1\n v_cvt_f32_i32 %2,%6\n v_cvt_f32_u32 %3,%7")

static double ref = 0;
void run(const char* name, void (*kern)(float*), float* sink, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)waves_per_simd * OUTER * 32 * 4;
    const double cyc = ms * 1e-3 * 2.4e9 / inst_per_simd;
    if (ref == 0) ref = cyc;
    printf("%-12s %d waves/SIMD: %.3f ms  %.2f cyc/inst @2.4GHz  (%.2fx v_fma_f32)\n", name, waves_per_simd, ms, cyc, cyc / ref);
}
#define R(K) run(#K, K, sink, w)
int main() {
    float* sink; hipMalloc(&sink, 1024 * 16 * 64 * 4);
    for (int w : {8, 2}) {
        ref = 0;
        R(k_fma); R(k_fmac); R(k_fma_sgpr); R(k_fma_abs); R(k_mul); R(k_add); R(k_max); R(k_mov); R(k_cnd_vcc); R(k_cnd_sgpr);
        R(k_cmp_vcc); R(k_cmp_sgpr); R(k_cmpu_vcc); R(k_addu); R(k_and); R(k_lshl); R(k_lshl64); R(k_addco); R(k_ffbl); R(k_alignbit);
        R(k_mad24); R(k_mullo); R(k_bfe); R(k_add3); R(k_cvt); R(k_rcp); R(k_sqrt); R(k_pkfma); R(k_pkmul); R(k_fma64); R(k_mul64); R(k_rcp64);
        R(k_readlane); R(k_salu); R(k_salu64); R(k_mix_vs); R(k_mbcnt);
    }
    return 0;
}
