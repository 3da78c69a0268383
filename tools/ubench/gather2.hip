// Micro-benchmark 2: per-lane gathers — locality window, AoS 32-B records vs two 16-B arrays, ds_bpermute.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
// MODE 0: one b128 | 1: two b128 from two arrays (same idx) | 2: two b128 from one 32-B record | 3: 8x ds_bpermute | 4: 2x LDS b128 (AoS 32B) | 5: LDS 2 arrays
template <int MODE>
__global__ void __launch_bounds__(64) k(const float4* src, const float4* src2, float* sink, int window_elems, unsigned seed) {
    __shared__ float4 sh[1024];
    const int lane = threadIdx.x;
    if (MODE >= 4) for (int i = lane; i < 1024; i += 64) sh[i] = src[i];
    __syncthreads();
    unsigned s = seed + lane * 2654435761u + blockIdx.x * 40503u;
    const size_t off = (size_t)(blockIdx.x % 64) * window_elems * 2;
    float acc = 0; float r0 = lane, r1 = lane * 2, r2 = lane * 3, r3 = lane * 5, r4 = 1, r5 = 2, r6 = 3, r7 = 4;
    for (int it = 0; it < ITERS; ++it) {
        s = s * 1664525u + 1013904223u;
        const int idx = (s >> 8) % window_elems;
        if (MODE == 0) { float4 v = src[off + idx]; acc += v.x + v.w; }
        if (MODE == 1) { float4 v = src[off + idx]; float4 w = src2[off + idx]; acc += v.x + v.w + w.y; }
        if (MODE == 2) { float4 v = src[off + 2 * idx]; float4 w = src[off + 2 * idx + 1]; acc += v.x + v.w + w.y; }
        if (MODE == 3) {
            const int a = (idx & 63) << 2;
            acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, r0)));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, r1)));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, r2)));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, r3)));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, r4)));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, r5)));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, r6)));
            acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a, __builtin_bit_cast(int, r7)));
            r0 += 1; r4 += acc;
        }
        if (MODE == 4) { float4 v = sh[(2 * idx) & 1023]; float4 w = sh[((2 * idx) & 1023) + 1]; acc += v.x + v.w + w.y; }
        if (MODE == 5) { float4 v = sh[idx & 511]; float4 w = sh[512 + (idx & 511)]; acc += v.x + v.w + w.y; }
    }
    sink[blockIdx.x * 64 + lane] = acc;
}
template <int MODE> void run(const char* name, const float4* src, const float4* src2, float* sink, int window, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, src, src2, sink, window, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, src, src2, sink, window, 7u);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double it_per_cu = (double)blocks * ITERS / 256.0;
    printf("%-28s window %7d B blocks %5d: %.3f ms -> %.1f CU-cycles per iteration @2.4GHz\n", name, window * 16, blocks, ms,
           ms * 1e-3 * 2.4e9 / it_per_cu);
}
int main() {
    float4 *src, *src2; float* sink; size_t n = 1 << 23;
    hipMalloc(&src, n * 16); hipMemset(src, 0, n * 16); hipMalloc(&src2, n * 16); hipMemset(src2, 0, n * 16);
    hipMalloc(&sink, 8192 * 64 * 4);
    const int blocks = 8192;
    for (int w : {4, 16, 64, 192, 512, 4096, 65536}) {
        run<0>("global 1x16B", src, src2, sink, w, blocks);
        run<1>("global 2x16B two arrays", src, src2, sink, w, blocks);
        run<2>("global 2x16B one 32B record", src, src2, sink, w, blocks);
    }
    run<3>("8x ds_bpermute", src, src2, sink, 64, blocks);
    run<4>("LDS 2x16B one 32B record", src, src2, sink, 512, blocks);
    run<5>("LDS 2x16B two arrays", src, src2, sink, 512, blocks);
    return 0;
}
