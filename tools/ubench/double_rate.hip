// Micro-benchmark (round 5): what takes the double rate (2.4-2.7 cycles per wave64 instruction per SIMD) away from v_mul / v_fma / v_add —
// the force kernel runs its ~15 k vector instructions per tile at 4.2 cycles each although two thirds of them are double-rate instructions.
// Each body = 32 double-rate instructions + the suspect; cycles per BODY per SIMD at 6 waves per SIMD, 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OUTER 64
#define DEFI(NAME, PRE, BODY, POST)                                                                       \
    __global__ void __launch_bounds__(64) NAME(float* sink, float* mem) {                                 \
        float a = threadIdx.x * 0.5f + 1.f, b = 1.0001f, c = 0.3f, d = 0.7f, e = 1.1f, f = 0.9f, g = 1.3f, h = 0.8f; \
        unsigned ia = threadIdx.x, ib = 3, ic = 5, id = 7;                                                \
        unsigned long long s0 = 1, s1 = 2;                                                                \
        float* mp = mem + threadIdx.x * 4;                                                                \
        __shared__ float lds[256]; lds[threadIdx.x] = a; unsigned la = threadIdx.x * 4;                   \
        _Pragma("unroll 1") for (int it = 0; it < OUTER; ++it) {                                          \
            asm volatile(PRE "\n.rept 16\n" BODY "\n.endr\n" POST                                         \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id), "+s"(s0), "+s"(s1), \
                           "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(mp), "+v"(la) :: "vcc", "scc", "s20", "s21", "s22", "s23", "memory"); \
        }                                                                                                 \
        sink[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h + ia + ib + ic + id + (float)(s0 + s1) + lds[(threadIdx.x + 1) & 63]; \
    }
// %0-%3, %10-%13 float; %4-%7 uint; %8, %9 sgpr pairs; %14 = 64-bit global pointer; %15 = LDS byte address
#define F4 "v_mul_f32 %0,%0,%1\n v_fma_f32 %10,%10,%11,%11\n v_mul_f32 %2,%2,%3\n v_fma_f32 %12,%12,%13,%13\n"
#define F32 F4 F4 F4 F4 F4 F4 F4 F4
DEFI(d_base,        "", F32, "")
DEFI(d_sgpr_src,    "", "v_mul_f32 %0,s20,%0\n v_fma_f32 %10,%10,s21,%11\n v_mul_f32 %2,s22,%2\n v_fma_f32 %12,%12,s23,%13\n" F4 F4 F4 F4 F4 F4 F4, "")
DEFI(d_sgpr_all,    "", ".rept 8\n v_mul_f32 %0,s20,%0\n v_fma_f32 %10,%10,s21,%11\n v_mul_f32 %2,s22,%2\n v_fma_f32 %12,%12,s23,%13\n .endr", "")
DEFI(d_literal,     "", ".rept 8\n v_mul_f32 %0,0x3f800347,%0\n v_add_f32 %10,0x3a83126f,%10\n v_mul_f32 %2,0x3f800347,%2\n v_add_f32 %12,0x3a83126f,%12\n .endr", "")
DEFI(d_inline_const,"", ".rept 8\n v_mul_f32 %0,1.0,%0\n v_add_f32 %10,0.5,%10\n v_mul_f32 %2,1.0,%2\n v_add_f32 %12,0.5,%12\n .endr", "")
DEFI(d_partial_exec,"s_mov_b64 s[20:21], exec\n s_mov_b32 exec_lo, 0x0f0f0f0f\n s_mov_b32 exec_hi, 0x0f0f0f0f", F32, "s_mov_b64 exec, s[20:21]")
DEFI(d_half_exec,   "s_mov_b64 s[20:21], exec\n s_mov_b32 exec_hi, 0", F32, "s_mov_b64 exec, s[20:21]")
DEFI(d_waitcnt,     "", F4 F4 "s_waitcnt vmcnt(0) lgkmcnt(0)\n" F4 F4 "s_waitcnt vmcnt(0)\n" F4 F4 "s_waitcnt lgkmcnt(0)\n" F4 F4, "")
DEFI(d_saveexec,    "", F4 F4 "v_cmp_lt_u32 vcc,%4,%5\n s_and_saveexec_b64 s[20:21], vcc\n" F4 F4 "s_or_b64 exec, exec, s[20:21]\n" F4 F4 F4 F4, "")
DEFI(d_exec_write,  "", F4 F4 "s_mov_b64 s[20:21], exec\n s_mov_b64 exec, s[20:21]\n" F4 F4 F4 F4 F4 F4, "")
DEFI(d_branch,      "", F4 F4 "s_cbranch_scc1 1f\n" F4 "1:\n" F4 F4 F4 F4 F4, "s_cmp_eq_u32 s20, s20")
DEFI(d_branch_vccz, "", F4 F4 "v_cmp_lt_u32 vcc,%4,%5\n s_cbranch_vccz 1f\n" F4 "1:\n" F4 F4 F4 F4 F4, "")
DEFI(d_ds_read,     "", F4 F4 "ds_read_b64 v[60:61], %15\n" F4 F4 F4 F4 F4 F4, "s_waitcnt lgkmcnt(0)")
DEFI(d_ds_write,    "", F4 F4 "ds_write_b64 %15, v[60:61]\n" F4 F4 F4 F4 F4 F4, "s_waitcnt lgkmcnt(0)")
DEFI(d_gload,       "", F4 F4 "global_load_dwordx4 v[60:63], %14, off\n" F4 F4 F4 F4 F4 F4, "s_waitcnt vmcnt(0)")
DEFI(d_gload2,      "", F4 F4 "global_load_dwordx4 v[60:63], %14, off\n global_load_dwordx4 v[64:67], %14, off offset:16\n" F4 F4 F4 F4 F4 F4, "s_waitcnt vmcnt(0)")
DEFI(d_trans_1in32, "", "v_rcp_f32 %1,%1\n" F32, "")
DEFI(d_trans_once,  "v_rcp_f32 %1,%1", F32, "")
DEFI(d_slow_8in32,  "", ".rept 8\n v_mul_f32 %0,%0,%1\n v_fma_f32 %10,%10,%11,%11\n v_cmp_lt_u32 %8,%4,%5\n v_mul_f32 %2,%2,%3\n v_fma_f32 %12,%12,%13,%13\n .endr", "")
DEFI(d_ffbl_8in32,  "", ".rept 8\n v_mul_f32 %0,%0,%1\n v_fma_f32 %10,%10,%11,%11\n v_ffbl_b32 %6,%7\n v_mul_f32 %2,%2,%3\n v_fma_f32 %12,%12,%13,%13\n .endr", "")
DEFI(d_cndmask,     "", ".rept 8\n v_mul_f32 %0,%0,%1\n v_fma_f32 %10,%10,%11,%11\n v_cndmask_b32 %6,%7,%6,vcc\n v_mul_f32 %2,%2,%3\n v_fma_f32 %12,%12,%13,%13\n .endr", "")
DEFI(d_readlane,    "", F4 F4 "v_readlane_b32 s20, %4, 3\n" F4 F4 F4 F4 F4 F4, "")
DEFI(d_ballot,      "", F4 F4 "v_cmp_ne_u32 %8,0,%4\n s_cmp_lg_u64 %8, 0\n" F4 F4 F4 F4 F4 F4, "")
DEFI(d_mfma,        "", F4 F4 "v_mfma_f32_32x32x2_f32 v[68:83], %0, %1, v[68:83]\n" F4 F4 F4 F4 F4 F4, "")
DEFI(d_permlane,    "", F4 F4 "v_permlane32_swap_b32 %5, %6\n" F4 F4 F4 F4 F4 F4, "")
DEFI(d_pk,          "", F4 F4 "v_pk_mul_f32 v[60:61], v[62:63], v[64:65]\n" F4 F4 F4 F4 F4 F4, "")
DEFI(d_f64,         "", F4 F4 "v_mul_f64 v[60:61], v[62:63], v[64:65]\n" F4 F4 F4 F4 F4 F4, "")
DEFI(d_salu8,       "", ".rept 8\n v_mul_f32 %0,%0,%1\n s_add_u32 s20,s20,s21\n v_fma_f32 %10,%10,%11,%11\n s_and_b64 %8,%8,%9\n v_mul_f32 %2,%2,%3\n v_fma_f32 %12,%12,%13,%13\n .endr", "")
// what ends the slow phase behind a transcendental?  (one v_rcp_f32, the candidate, 32 double-rate instructions)
#define TR "v_rcp_f32 %1,%1\n"
DEFI(r_none,      "", TR F32, "")
DEFI(r_nop7,      "", TR "s_nop 7\n" F32, "")
DEFI(r_nop7x4,    "", TR "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n" F32, "")
DEFI(r_waitcnt,   "", TR "s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)\n" F32, "")
DEFI(r_branch,    "", TR "s_branch 1f\n1:\n" F32, "")
DEFI(r_cbranch,   "", TR "s_cbranch_scc1 1f\n1:\n" F32, "")
DEFI(r_sleep,     "", TR "s_sleep 0\n" F32, "")
DEFI(r_setprio,   "", TR "s_setprio 0\n" F32, "")
DEFI(r_vnop,      "", TR "v_nop\n" F32, "")
DEFI(r_salu4,     "", TR "s_add_u32 s20,s20,s21\n s_add_u32 s21,s21,s20\n s_add_u32 s20,s20,s21\n s_add_u32 s21,s21,s20\n" F32, "")
DEFI(r_indep,     "", "v_rcp_f32 v70,v71\n" F32, "")
DEFI(r_indep_late,"", "v_rcp_f32 v70,v71\n" F32 "v_mul_f32 %1,%1,v70\n", "")
DEFI(r_slow8,     "", TR "v_cmp_lt_u32 %8,%4,%5\n v_cmp_lt_u32 %9,%5,%6\n v_cmp_lt_u32 %8,%6,%7\n v_cmp_lt_u32 %9,%7,%4\n v_cmp_lt_u32 %8,%4,%5\n v_cmp_lt_u32 %9,%5,%6\n v_cmp_lt_u32 %8,%6,%7\n v_cmp_lt_u32 %9,%7,%4\n" F32, "")
DEFI(r_vmem2_sep, "", F4 F4 "global_load_dwordx4 v[60:63], %14, off\n" F4 "global_load_dwordx4 v[64:67], %14, off offset:16\n" F4 F4 F4 F4 F4, "s_waitcnt vmcnt(0)")
DEFI(r_vmem2_sep1,"", F4 F4 "global_load_dwordx4 v[60:63], %14, off\n v_mul_f32 %0,%0,%1\n global_load_dwordx4 v[64:67], %14, off offset:16\n" F4 F4 F4 F4 F4 F4, "s_waitcnt vmcnt(0)")
DEFI(r_vmem2_wait,"", F4 F4 "global_load_dwordx4 v[60:63], %14, off\n global_load_dwordx4 v[64:67], %14, off offset:16\n s_waitcnt vmcnt(0)\n" F4 F4 F4 F4 F4 F4, "")
DEFI(r_vmem2_br,  "", F4 F4 "global_load_dwordx4 v[60:63], %14, off\n global_load_dwordx4 v[64:67], %14, off offset:16\n s_branch 1f\n1:\n" F4 F4 F4 F4 F4 F4, "s_waitcnt vmcnt(0)")
DEFI(r_buf2,      "", F4 F4 "buffer_load_dwordx4 v[60:63], %4, s[24:27], 0 offen\n buffer_load_dwordx4 v[64:67], %4, s[24:27], 0 offen offset:16\n" F4 F4 F4 F4 F4 F4, "s_waitcnt vmcnt(0)")
DEFI(d_nop,         "", F4 F4 "s_nop 0\n" F4 F4 "s_nop 0\n" F4 F4 F4 F4, "")

void run(const char* name, void (*kern)(float*, float*), float* sink, float* mem, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink, mem);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, sink, mem);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bodies_per_simd = (double)waves_per_simd * OUTER * 16;
    printf("%-16s %d waves/SIMD: %7.2f cycles per body (32 double-rate instructions + the suspect) @2.4GHz\n", name, waves_per_simd,
           ms * 1e-3 * 2.4e9 / bodies_per_simd);
}
#define R(K) run(#K, K, sink, mem, w)
int main() {
    float *sink, *mem; (void)hipMalloc(&sink, 1024 * 16 * 64 * 4); (void)hipMalloc(&mem, 1 << 20); (void)hipMemset(mem, 0, 1 << 20);
    for (int w : {6, 2}) {
        R(d_base); R(d_sgpr_src); R(d_sgpr_all); R(d_literal); R(d_inline_const); R(d_partial_exec); R(d_half_exec); R(d_waitcnt); R(d_saveexec); R(d_exec_write);
        R(d_branch); R(d_branch_vccz); R(d_ds_read); R(d_ds_write); R(d_gload); R(d_gload2); R(d_trans_1in32); R(d_trans_once); R(d_slow_8in32); R(d_ffbl_8in32);
        R(d_cndmask); R(d_readlane); R(d_ballot); R(d_mfma); R(d_permlane); R(d_pk); R(d_f64); R(d_salu8); R(d_nop);
        R(r_none); R(r_nop7); R(r_nop7x4); R(r_waitcnt); R(r_branch); R(r_cbranch); R(r_sleep); R(r_setprio); R(r_vnop); R(r_salu4); R(r_indep); R(r_indep_late); R(r_slow8);
        R(r_vmem2_sep); R(r_vmem2_sep1); R(r_vmem2_wait); R(r_vmem2_br);
    }
    return 0;
}
