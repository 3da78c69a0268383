// Feasibility probe for the phase-1 design of profiles/HISTORY.md §8: the sign bits of a 32×32 accumulator block gathered ON THE MATRIX PIPE.
//  (1) `v_cvt_pknorm_u16_f32 dst, -a, -b` on values pre-scaled so that every non-zero magnitude is ≥ 1: each half is exactly
//      0x0000 or 0xffff (0xffff iff the input is negative; ±0, +inf and NaN give 0).  (`v_cvt_pkrtz_f16_f32 … clamp` assembles
//      but the clamp is IGNORED by the hardware: the halves come out as ±65504 — first version of this probe.)
//  (2) eight registers of a block converted that way ARE a B operand of v_mfma_i32_32x32x32_i8 (a lane's 16 bytes = two per value,
//      each −1 or 0).  A constant A operand with the weights (2^j, 0) for value j < 7 and (64, 64) for value 7, in rows 0 / 4 for
//      the slots of lane half 0 and in rows 1 / 5 for those of half 1, leaves −(byte of half 0) in accumulator 0 and −(byte of
//      half 1) in accumulator 1 of BOTH lanes of a column: exact integers, no lane exchange.
// Compared against the v_alignbit construction the kernel uses today.  Prints the number of mismatches (0 = feasible).
// (What became of it: profiles/r03_pair_loop_experiments.md §10 — exact here, but slower in the kernel and not exact on near-ties of the real chain.)
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mask_pack tools/ubench/mask_pack.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk01(float a, float b) {
    unsigned r;
    asm volatile("v_cvt_pknorm_u16_f32 %0, -%1, -%2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// in: [block][lane][16] accumulator-like values (already scaled); out_ref: [block][lane][2] today's bytes (registers 0–7, 8–15);
// out_mfma: [block][lane][2] the 16-bit masks of the column from the product; out_pk: [block][lane][8] the packed conversions
__global__ void __launch_bounds__(64) k_probe(const float* in, unsigned* out_ref, unsigned* out_mfma, unsigned* out_pk) {
    const int lane = threadIdx.x, h = lane >> 5, m = lane & 31;
    const float* d = in + ((size_t)blockIdx.x * 64 + lane) * 16;
    // the constant operand: row m of A' against the 16 byte slots (8 values × 2) of lane half h
    const bool mine = h == 0 ? (m == 0 || m == 4) : (m == 1 || m == 5);
    i32x4 w;
    w[0] = mine ? (1 | (2 << 16)) : 0;              // value 0: bytes (1, 0), value 1: (2, 0)
    w[1] = mine ? (4 | (8 << 16)) : 0;
    w[2] = mine ? (16 | (32 << 16)) : 0;
    w[3] = mine ? (64 | (0x4040 << 16)) : 0;        // value 6: (64, 0), value 7: (64, 64)
    for (int s = 0; s < 2; ++s) {
        i32x4 pk;
        for (int q = 0; q < 4; ++q) {
            pk[q] = (int)pk01(d[8 * s + 2 * q], d[8 * s + 2 * q + 1]);
            out_pk[((size_t)blockIdx.x * 64 + lane) * 8 + 4 * s + q] = (unsigned)pk[q];
        }
        i32x16 acc = {0};
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, pk, acc, 0, 0, 0);
        out_mfma[((size_t)blockIdx.x * 64 + lane) * 2 + s] = (unsigned)(-(acc[0] + 256 * acc[1]));
        unsigned mk = 0;                              // today's construction: bit j = sign of register 8s + j
        for (int j = 7; j >= 0; --j) mk = __builtin_amdgcn_alignbit(mk, __float_as_uint(d[8 * s + j]), 31);
        out_ref[((size_t)blockIdx.x * 64 + lane) * 2 + s] = mk & 0xffu;
    }
}

int main() {
    const int nb = 4096;
    std::vector<float> h((size_t)nb * 64 * 16);
    srand(7);
    const float H2 = 9e-4f;                                          // H ≈ 0.03
    const float scale = std::ldexp(1.0f, 26 - std::ilogb(H2));      // every non-zero |d| ≥ ulp(H²) lands at ≥ 4
    for (size_t i = 0; i < h.size(); ++i) {
        const int c = rand() % 16;
        float v = ((rand() % 2000001) - 1000000) * 1e-9f;            // |c − t|² − H²: ±1e-3
        if (c == 0) v = 0.0f; else if (c == 1) v = -0.0f; else if (c == 2) v = std::ldexp(H2, -23); else if (c == 3) v = -std::ldexp(H2, -23);   // ties, one-ulp results
        else if (c == 4) v = 1e30f;                                  // the padding value of an invalid candidate (→ +inf after scaling)
        else if (c == 5) v = NAN;                                    // a row nobody has written
        h[i] = v * scale;
    }
    float* din; unsigned *dref, *dm, *dpk;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dref, (size_t)nb * 64 * 2 * 4); hipMalloc(&dm, (size_t)nb * 64 * 2 * 4); hipMalloc(&dpk, (size_t)nb * 64 * 8 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(nb), dim3(64), 0, 0, din, dref, dm, dpk);
    std::vector<unsigned> ref((size_t)nb * 64 * 2), m(ref.size()), pk((size_t)nb * 64 * 8);
    hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, m.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(pk.data(), dpk, pk.size() * 4, hipMemcpyDeviceToHost);
    size_t bad_cvt = 0, special = 0;
    for (size_t b = 0; b < (size_t)nb * 64; ++b)
        for (int i = 0; i < 16; ++i) {
            const float v = h[b * 16 + i];
            const unsigned got = (pk[b * 8 + i / 2] >> (16 * (i & 1))) & 0xffffu;
            const bool want1 = v < 0.0f;                             // −0, NaN, +inf: 0
            bad_cvt += got != (want1 ? 0xffffu : 0u);
            special += (v != v) || (v == 0.0f && std::signbit(v));
        }
    // the product against "x < 0" of the sixteen values of the column's two lanes (NaN and −0 have their sign bits set or not
    // by accident: today's sign test takes a −0 and a negative NaN, x < 0 takes neither — rows of that kind are counted apart)
    size_t bad_mask = 0, differs_from_alignbit = 0;
    for (int b = 0; b < nb; ++b)
        for (int lane = 0; lane < 64; ++lane) for (int s = 0; s < 2; ++s) {
            const int n = lane & 31;
            unsigned want = 0;
            for (int hh = 0; hh < 2; ++hh) for (int j = 0; j < 8; ++j) want |= (unsigned)(h[((size_t)b * 64 + n + 32 * hh) * 16 + 8 * s + j] < 0.0f) << (8 * hh + j);
            bad_mask += m[((size_t)b * 64 + lane) * 2 + s] != want;
            differs_from_alignbit += want != (ref[((size_t)b * 64 + n) * 2 + s] | (ref[((size_t)b * 64 + n + 32) * 2 + s] << 8));
        }
    printf("conversions: %zu wrong of %zu (%zu of the inputs are NaN or −0)\n", bad_cvt, (size_t)nb * 64 * 16, special);
    printf("packing product: %zu wrong 16-bit masks of %zu (%zu of them differ from the sign-bit masks: the −0 / NaN inputs)\n", bad_mask, (size_t)nb * 64 * 2, differs_from_alignbit);
    return (bad_cvt || bad_mask) ? 1 : 0;
}
