// Micro-benchmark 4 (round 5): what decides the cost of a wave-level b128 gather — the number of lanes, or the number of distinct segments?
// Address patterns inside a window of 32-byte records (6 waves per SIMD, cycles per CU and instruction):
//   0 every lane its own random record (packet 0)                 1 coalesced: lane l reads packet l of a random run of 64
//   2 lane PAIRS (2l, 2l+1) read packets 0 / 1 of ONE random record   3 QUADS read 4 consecutive packets (two records)
//   4 as 0 with the upper 32 lanes switched off                    5 lanes l and l+32 read packets 0 / 1 of one random record
//   6 every lane its own record, records of adjacent lanes NEAR each other
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 1024
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(const float4* src, float* sink, int window_recs, unsigned seed, unsigned bytes, int shared_window) {
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned off = (shared_window ? 0u : (wid & 255)) * window_recs * 32u;
    unsigned acc = 0;
    const unsigned g = MODE == 2 ? lane >> 1 : MODE == 3 ? lane >> 2 : MODE == 5 ? (lane & 31) : MODE == 1 ? 0 : lane;
    unsigned s = seed + (wid * 64 + g) * 2654435761u;
    for (int it = 0; it < ITERS; ++it) {
        s = s * 1664525u + 1013904223u;
        unsigned rec = (s >> 8) & (window_recs - 1);
        unsigned a = 0;
        if (MODE == 0 || MODE == 4) a = off + rec * 32u;
        if (MODE == 1) a = off + (rec & (window_recs - 1) & ~31u) * 32u + lane * 16u;
        if (MODE == 2) a = off + rec * 32u + (lane & 1) * 16u;
        if (MODE == 3) a = off + (rec & ~1u) * 32u + (lane & 3) * 16u;
        if (MODE == 5) a = off + rec * 32u + (lane >> 5) * 16u;
        if (MODE == 6) a = off + (((rec & ~63u) + lane + (rec & 3)) & (window_recs - 1)) * 32u;
        if (MODE == 4) { if (lane < 32) { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, a, 0, 0); acc += v.x + v.w; } }
        else { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, a, 0, 0); acc += v.x + v.w; }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name, const float4* src, float* sink, int window, int blocks, unsigned bytes, int sh = 1) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, src, sink, window, 1u, bytes, sh);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, src, sink, window, 7u, bytes, sh);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double it_per_cu = (double)blocks * 4 * ITERS / 256.0;
    printf("(one window for all waves: L1 hits) %-44s window %7d B: %.3f ms -> %6.1f CU-cycles per instruction @2.4GHz\n", name, window * 32, ms, ms * 1e-3 * 2.4e9 / it_per_cu);
}
int main() {
    float4* src; float* sink; const size_t n = 1 << 24;
    (void)hipMalloc(&src, n * 16); (void)hipMemset(src, 0, n * 16);
    const int blocks = 256 * 6;
    (void)hipMalloc(&sink, blocks * 256 * 4);
    for (int w : {128, 512}) {
        run<0>("64 lanes, 64 random records", src, sink, w, blocks, (unsigned)(n * 16));
        run<1>("coalesced run of 64 packets", src, sink, w, blocks, (unsigned)(n * 16));
        run<2>("lane pairs: one record each (32 records)", src, sink, w, blocks, (unsigned)(n * 16));
        run<3>("quads: two records each (32 records)", src, sink, w, blocks, (unsigned)(n * 16));
        run<4>("32 lanes, 32 random records", src, sink, w, blocks, (unsigned)(n * 16));
        run<5>("lanes l, l+32: one record each (32 records)", src, sink, w, blocks, (unsigned)(n * 16));
        run<6>("64 lanes, records of adjacent lanes adjacent", src, sink, w, blocks, (unsigned)(n * 16));
    }
    return 0;
}
