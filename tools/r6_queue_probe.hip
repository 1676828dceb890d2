// Round 6 probe: how many branches of a captured HIP graph are co-resident? N one-wave kernels on N streams each raise a counter and wait (<= 20 ms)
// until all N have arrived. With more branches than hardware queues the late ones sit behind an early one in the same in-order queue and never arrive
// while it waits.   hipcc --offload-arch=gfx950 tools/r6_queue_probe.hip -o tools/r6_queue_probe; [GPU_MAX_HW_QUEUES=8] tools/r6_queue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void meet(unsigned *cnt, unsigned n, unsigned *seen)
{
    if (threadIdx.x == 0) atomicAdd(cnt, 1u);
    const unsigned long long t0 = wall_clock64();
    unsigned v;
    while ((v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < n && wall_clock64() - t0 < 2000000ull) __builtin_amdgcn_s_sleep(8);
    atomicMax(seen, v);
}
int main()
{
    unsigned *d;
    hipMalloc(&d, 64);
    for (int use_graph = 0; use_graph < 2; use_graph++)
        for (int n = 2; n <= 10; n++) {
            hipStream_t s[16];
            for (int i = 0; i < n; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
            hipEvent_t f, j[16];
            hipEventCreateWithFlags(&f, hipEventDisableTiming);
            for (int i = 0; i < n; i++) hipEventCreateWithFlags(&j[i], hipEventDisableTiming);
            hipMemset(d, 0, 64);
            hipGraph_t g; hipGraphExec_t ge;
            if (use_graph) hipStreamBeginCapture(s[0], hipStreamCaptureModeRelaxed);
            hipEventRecord(f, s[0]);
            for (int i = 1; i < n; i++) hipStreamWaitEvent(s[i], f, 0);
            for (int i = 0; i < n; i++) meet<<<1, 64, 0, s[i]>>>(d, (unsigned)n, d + 8);
            for (int i = 1; i < n; i++) { hipEventRecord(j[i], s[i]); hipStreamWaitEvent(s[0], j[i], 0); }
            if (use_graph) { hipStreamEndCapture(s[0], &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0); hipGraphLaunch(ge, s[0]); }
            hipDeviceSynchronize();
            unsigned h[16];
            hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("%s, %d branches: the most any branch saw arrive while it waited: %u%s\n", use_graph ? "graph" : "streams", n, h[8], h[8] == (unsigned)n ? "" : "  <-- not all co-resident");
            for (int i = 0; i < n; i++) hipStreamDestroy(s[i]);
        }
    return 0;
}
