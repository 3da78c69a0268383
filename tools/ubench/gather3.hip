// Micro-benchmark 3 (round 5): what a wave-level gather costs the texture path by its WIDTH — 32-byte records, scattered inside a window, through the
// buffer loads the kernels use: b128 + b128 (the pair loop today), b128 + b96, b128 + b64, and each width alone.  6 waves per SIMD, cycles per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 1024
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x3_t __attribute__((ext_vector_type(3)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(const float4* src, float* sink, int window_recs, unsigned seed, unsigned bytes, int shared_window) {
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    const unsigned off = (shared_window ? 0u : (unsigned)((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255)) * window_recs * 32u;
    unsigned acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned a = off + ((s >> 8) & (window_recs - 1)) * 32u;
        if (MODE == 0) { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, a, 0, 0), w = __builtin_amdgcn_raw_buffer_load_b128(r, a + 16, 0, 0); acc += v.x + v.w + w.y + w.w; }
        if (MODE == 1) { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, a, 0, 0); u32x3_t w = __builtin_amdgcn_raw_buffer_load_b96(r, a + 16, 0, 0); acc += v.x + v.w + w.y + w.z; }
        if (MODE == 2) { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, a, 0, 0); u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(r, a + 16, 0, 0); acc += v.x + v.w + w.y + w.x; }
        if (MODE == 3) { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, a, 0, 0); acc += v.x + v.w; }
        if (MODE == 4) { u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(r, a, 0, 0); acc += v.x + v.z; }
        if (MODE == 5) { u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, a, 0, 0); acc += v.x + v.y; }
        if (MODE == 6) { unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, a, 0, 0); acc += v; }
        if (MODE == 7) { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, a, 0, 0); u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(r, a + 16, 0, 0); unsigned x = __builtin_amdgcn_raw_buffer_load_b32(r, a + 24, 0, 0); acc += v.x + v.w + w.y + w.x + x; }
        if (MODE == 8) { u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(r, a, 0, 0); u32x3_t w = __builtin_amdgcn_raw_buffer_load_b96(r, a + 12, 0, 0); acc += v.x + v.z + w.y + w.z; }
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
    printf("(one window for all waves: L1 hits) %-34s window %7d B: %.3f ms -> %6.1f CU-cycles per wave iteration @2.4GHz\n", name, window * 32, ms, ms * 1e-3 * 2.4e9 / it_per_cu);
}
int main() {
    float4* src; float* sink; const size_t n = 1 << 24;
    (void)hipMalloc(&src, n * 16); (void)hipMemset(src, 0, n * 16);
    const int blocks = 256 * 6;           // 6 waves per SIMD, one round
    (void)hipMalloc(&sink, blocks * 256 * 4);
    for (int w : {8, 128, 512, 4096}) {
        run<0>("b128 + b128 (one record)", src, sink, w, blocks, (unsigned)(n * 16));
        run<1>("b128 + b96", src, sink, w, blocks, (unsigned)(n * 16));
        run<2>("b128 + b64", src, sink, w, blocks, (unsigned)(n * 16));
        run<7>("b128 + b64 + b32", src, sink, w, blocks, (unsigned)(n * 16));
        run<8>("b96 + b96", src, sink, w, blocks, (unsigned)(n * 16));
        run<3>("b128", src, sink, w, blocks, (unsigned)(n * 16));
        run<4>("b96", src, sink, w, blocks, (unsigned)(n * 16));
        run<5>("b64", src, sink, w, blocks, (unsigned)(n * 16));
        run<6>("b32", src, sink, w, blocks, (unsigned)(n * 16));
    }
    return 0;
}
