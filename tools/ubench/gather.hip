// Micro-benchmark: cost of per-lane gathers through the vector memory path (TA/TCP) vs LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 2048
template <int BYTES, int LDS>
__global__ void __launch_bounds__(64) k(const float4* src, float* sink, int window_elems, unsigned seed) {
    __shared__ float4 sh[512];
    const int lane = threadIdx.x;
    if (LDS) for (int i = lane; i < 512; i += 64) sh[i] = src[i];
    __syncthreads();
    unsigned s = seed + lane * 2654435761u + blockIdx.x * 40503u;
    const float4* base = src + (size_t)(blockIdx.x % 64) * window_elems;
    float acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        s = s * 1664525u + 1013904223u;
        const int idx = (s >> 8) % window_elems;
        if (LDS) {
            if (BYTES == 16) { float4 v = sh[idx & 511]; acc += v.x + v.w; }
            else if (BYTES == 8) { float2 v = ((const float2*)sh)[(idx & 511) * 2]; acc += v.x + v.y; }
            else { acc += ((const float*)sh)[(idx & 511) * 4]; }
        } else {
            if (BYTES == 16) { float4 v = base[idx]; acc += v.x + v.w; }
            else if (BYTES == 8) { float2 v = ((const float2*)base)[idx * 2]; acc += v.x + v.y; }
            else { acc += ((const float*)base)[idx * 4]; }
        }
    }
    sink[blockIdx.x * 64 + lane] = acc;
}
template <int BYTES, int LDS> void run(const char* name, const float4* src, float* sink, int window, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BYTES, LDS>), dim3(blocks), dim3(64), 0, 0, src, sink, window, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BYTES, LDS>), dim3(blocks), dim3(64), 0, 0, src, sink, window, 7u);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_cu = (double)blocks * ITERS / 256.0;
    printf("%-22s window %6d B  blocks %5d: %.3f ms  -> %.1f CU-cycles per wave-gather @2.4GHz, %.1f GB/s/CU\n", name,
           window * 16, blocks, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_cu, 64.0 * BYTES * wave_instr_per_cu / (ms * 1e-3) / 1e9);
}
int main() {
    float4* src; float* sink; size_t n = 1 << 22;
    hipMalloc(&src, n * 16); hipMemset(src, 0, n * 16); hipMalloc(&sink, 8192 * 64 * 4);
    for (int blocks : {2048, 8192}) {
        for (int w : {64, 256, 4096, 65536}) {
            run<16, 0>("global 16B", src, sink, w, blocks);
            run<8, 0>("global 8B", src, sink, w, blocks);
            run<4, 0>("global 4B", src, sink, w, blocks);
        }
        run<16, 1>("LDS 16B", src, sink, 512, blocks);
        run<8, 1>("LDS 8B", src, sink, 512, blocks);
        run<4, 1>("LDS 4B", src, sink, 512, blocks);
    }
    return 0;
}
