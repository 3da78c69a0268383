// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  Each kernel fills a quarter of the chip
// for ~1 ms; serialised: ~2 ms per pair, overlapped: ~1 ms.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(float* out, int iters) {
    float x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) out[0] = x;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 400000;
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, s);
            for (int k = 0; k < 4; ++k)
                hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, d, iters);
            hipEventRecord(b, s);
            hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("flags=%d  4 kernels: %.3f ms\n", flags, ms);
        }
    }
    return 0;
}
