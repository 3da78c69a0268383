// Probe: does hipStreamWaitValue32 park a stream until the HOST (or another stream) writes the signal word — on a created stream and on the null stream?
// (tests/mock_rccl's asynchronous mode rests on it.)   hipcc --offload-arch=gfx950 wait_value_probe.cpp -o wait_value_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    int can = -1; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    void* sig = nullptr; CK(hipExtMallocWithFlags(&sig, 8, hipMallocSignalMemory));
    *(volatile uint64_t*)sig = 0;
    hipStream_t s, w; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&w, hipStreamNonBlocking));
    for (int which = 0; which < 2; ++which) {
        hipStream_t q = which == 0 ? s : nullptr;
        const uint32_t v = 5 + which;
        CK(hipStreamWaitValue32(q, sig, v, hipStreamWaitValueGte, 0xffffffffu));
        std::thread t([&] { std::this_thread::sleep_for(std::chrono::milliseconds(300)); (void)hipStreamWriteValue32(w, sig, v, 0); (void)hipStreamSynchronize(w); });
        auto t0 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(q));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        t.join();
        printf("%s stream: hipStreamSynchronize returned after %.1f ms (the signal was written after 300 ms)\n", which == 0 ? "created" : "null", ms);
    }
    return 0;
}
