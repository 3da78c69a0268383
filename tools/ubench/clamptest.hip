// Do the clamp output modifiers used by the pair loop do what sphmi_kernels.h assumes?  (gfx950, IEEE mode on)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* out, float b, float big) {
    const int i = threadIdx.x;
    float r; asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(r) : "v"(in[i]), "s"(b));
    float s; asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(s) : "v"(in[i]), "s"(big));
    out[2 * i] = r; out[2 * i + 1] = s;
}
int main() {
    float h[8] = {0.0f, 0.5f, 1.0f, 3.0f, -1.0f, 1000.0f, -1000.0f, 1e-3f}, *d, *o, r[16];
    hipMalloc(&d, 32); hipMalloc(&o, 64); hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, -0.5f, 1099511627776.0f);
    hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("x = %g: clamp(1 - x/2) = %g (want %g)   step(x) = %g (want %g)\n", h[i], r[2 * i],
                                       fminf(fmaxf(1 - h[i] / 2, 0), 1), r[2 * i + 1], h[i] > 0 ? 1.0f : 0.0f);
    return 0;
}
