// Development probe: does a captured HIP graph run independent branches concurrently on this ROCm?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long cycles, int *out) {
    long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (out) out[0] = 1;
}
int main() {
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t f, j; hipEventCreateWithFlags(&f, hipEventDisableTiming); hipEventCreateWithFlags(&j, hipEventDisableTiming);
    int *d; hipMalloc(&d, 64);
    const long long cyc = 210000; // ~100 us
    for (int mode = 0; mode < 3; mode++) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(a, hipStreamCaptureModeRelaxed);
        if (mode == 0) { // serial chain of 20 kernels
            for (int i = 0; i < 20; i++) spin<<<1, 64, 0, a>>>(cyc, d);
        } else if (mode == 1) { // two parallel branches of 10
            hipEventRecord(f, a); hipStreamWaitEvent(b, f, 0);
            for (int i = 0; i < 10; i++) { spin<<<1, 64, 0, a>>>(cyc, d); spin<<<1, 64, 0, b>>>(cyc, d + 8); }
            hipEventRecord(j, b); hipStreamWaitEvent(a, j, 0);
        } else { // pipelined cross dependencies: b[i] after a[i]; a[i+2] after b[i]
            hipEvent_t ea[20], eb[20];
            for (int i = 0; i < 20; i++) { hipEventCreateWithFlags(&ea[i], hipEventDisableTiming); hipEventCreateWithFlags(&eb[i], hipEventDisableTiming); }
            hipEventRecord(f, a); hipStreamWaitEvent(b, f, 0);
            for (int i = 0; i < 10; i++) {
                if (i >= 2) hipStreamWaitEvent(a, eb[i - 2], 0);
                spin<<<1, 64, 0, a>>>(cyc, d); hipEventRecord(ea[i], a);
                hipStreamWaitEvent(b, ea[i], 0);
                spin<<<1, 64, 0, b>>>(cyc, d + 8); hipEventRecord(eb[i], b);
            }
            hipStreamWaitEvent(a, eb[9], 0);
        }
        hipStreamEndCapture(a, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, a); hipStreamSynchronize(a);
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; r++) hipGraphLaunch(ge, a);
        hipStreamSynchronize(a);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 5;
        printf("mode %d: %.1f us per graph (20 kernels of ~100 us)\n", mode, us);
    }
    // direct multi-stream, no graph
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 10; i++) { spin<<<1, 64, 0, a>>>(cyc, d); spin<<<1, 64, 0, b>>>(cyc, d + 8); }
    hipDeviceSynchronize();
    printf("direct 2 streams: %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    // launch overhead of tiny kernels, direct
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 1000; i++) spin<<<1, 64, 0, a>>>(0, d);
    hipDeviceSynchronize();
    printf("1000 tiny launches: %.1f us each\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1000);
    return 0;
}
