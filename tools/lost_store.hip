// lost_store.hip — standalone reproducer attempt for the "lost store" of DESIGN.md (known limits): one persistent WRITER workgroup publishes
// words write-through (the chain workgroup's st_sc1: relaxed agent-scope atomic store) at the chain's pace, POLLER workgroups on the other
// XCDs wait for each word with agent-scope loads (the update rows' / fold workgroups' ld_sc1), and a stream of short memory-bound kernels
// runs beside them (the mat-vec launches). Every poll is bounded; what is recorded is how long a word stayed invisible to a poller
// after the writer's own clock says it was stored. In the sampler, once in ~3 000 dense sweeps one such store stays invisible to every other
// XCD for > 3 s, always next to a ~0.9 ms pause of the launch stream. Does the pattern alone — without the sampler — show it?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/lost_store tools/lost_store.hip && tools/lost_store [seconds] [noise 0/1] [alloc 0/1/2]
//   alloc: 0 hipMalloc, 1 hipExtMallocWithFlags(uncached), 2 fine-grained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int RING = 1 << 16;          // words of the hand-off ring (one per cache line of 128 bytes: RING * 16 u64)
constexpr int STRIDE = 16;             // u64 per line
constexpr unsigned long long SENT = ~0ull;

__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); } // 100 MHz

struct stats {
    unsigned long long words, over_10us, over_100us, over_1ms, over_10ms, lost, max_ticks, xcc_mask;
    unsigned long long lost_idx[8], lost_seen_after[8];
};

// the writer: word i of epoch e gets the value e * RING + i + 1 (never the sentinel); paced by s_sleep like a chain sub-block (~2.5 us)
__global__ void k_writer(unsigned long long *ring, unsigned long long *wclock, volatile unsigned *stop, unsigned long long *progress, int pace)
{
    if (threadIdx.x != 0) return;
    unsigned long long n = 0;
    while (!*stop) {
        const unsigned long long i = n % RING;
        __hip_atomic_store(&ring[i * STRIDE], n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // st_sc1
        wclock[i] = wall();
        n++;
        if ((n & 1023) == 0) __hip_atomic_store(progress, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < pace; k++) __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_store(progress, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a poller block: lane 0 waits for word after word (every 64th word of the sequence, offset by its block index, so that the blocks together
// cover the sequence and no block has to keep up with the writer's full rate)
__global__ void k_poller(const unsigned long long *ring, volatile unsigned *stop, stats *st, int nblocks)
{
    if (threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    stats loc{};
    loc.xcc_mask = 1ull << (xcc & 15u);
    unsigned long long n = blockIdx.x; // sequence number waited for
    while (!*stop) {
        const unsigned long long i = n % RING, want = n + 1;
        const unsigned long long t0 = wall();
        unsigned long long v, waited = 0;
        for (;;) {
            v = __hip_atomic_load(&ring[i * STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ld_sc1
            if (v != SENT && v >= want) break; // (>= : the writer may have lapped this poller)
            waited = wall() - t0;
            if (waited > 30000000ull || *stop) break; // 300 ms: given up
            __builtin_amdgcn_s_sleep(2);
        }
        if (*stop) break;
        loc.words++;
        if (v == SENT || v < want) {
            if (loc.lost < 8) { loc.lost_idx[loc.lost] = n; loc.lost_seen_after[loc.lost] = v; }
            loc.lost++;
        }
        // (the wait of a poller that arrived BEFORE the writer is mostly the writer's pace; what matters is the tail)
        if (waited > 1000ull) loc.over_10us++;
        if (waited > 10000ull) loc.over_100us++;
        if (waited > 100000ull) loc.over_1ms++;
        if (waited > 1000000ull) loc.over_10ms++;
        if (waited > loc.max_ticks) loc.max_ticks = waited;
        n += nblocks;
    }
    st[blockIdx.x] = loc;
}

// the launch stream: a short streaming kernel, back to back
__global__ void k_noise(const float4 *__restrict__ a, float *__restrict__ out, size_t n4)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.f) out[0] = s;
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
    const int noise = argc > 2 ? atoi(argv[2]) : 1, alloc = argc > 3 ? atoi(argv[3]) : 0;
    unsigned long long *ring, *wclock, *progress;
    unsigned *stop;
    stats *st;
    const int NPOLL = 56; // blocks: dealt round-robin over the 8 XCDs
    const size_t rbytes = sizeof(unsigned long long) * RING * STRIDE;
    if (alloc == 0) CHECK(hipMalloc(reinterpret_cast<void **>(&ring), rbytes));
    else CHECK(hipExtMallocWithFlags(reinterpret_cast<void **>(&ring), rbytes, alloc == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    CHECK(hipMemset(ring, 0xff, rbytes));
    CHECK(hipMalloc(reinterpret_cast<void **>(&wclock), sizeof(unsigned long long) * RING));
    CHECK(hipMalloc(reinterpret_cast<void **>(&progress), 8));
    CHECK(hipMemset(progress, 0, 8));
    CHECK(hipHostMalloc(reinterpret_cast<void **>(&stop), 4, hipHostMallocMapped));
    *stop = 0;
    CHECK(hipMalloc(reinterpret_cast<void **>(&st), sizeof(stats) * NPOLL));
    CHECK(hipMemset(st, 0, sizeof(stats) * NPOLL));
    float4 *big;
    float *out;
    const size_t n4 = (size_t)64 << 20; // 1 GiB of float4 = what ~6 mat-vec launches read
    CHECK(hipMalloc(reinterpret_cast<void **>(&big), n4 * sizeof(float4)));
    CHECK(hipMemset(big, 0, n4 * sizeof(float4)));
    CHECK(hipMalloc(reinterpret_cast<void **>(&out), 4));
    hipStream_t sw, sp, sn;
    CHECK(hipStreamCreateWithFlags(&sw, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_poller, dim3(NPOLL), dim3(64), 0, sp, ring, stop, st, NPOLL);
    hipLaunchKernelGGL(k_writer, dim3(1), dim3(64), 0, sw, ring, wclock, stop, progress, 1);
    CHECK(hipGetLastError());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, sn));
    size_t launches = 0;
    float ms = 0.f;
    // 44.8 MB per launch = one 2-bit mat-vec launch's bytes; ~1 800 blocks
    const size_t per = (size_t)44800000 / sizeof(float4);
    while (ms < seconds * 1e3f) {
        if (noise)
            for (int k = 0; k < 64; k++, launches++) hipLaunchKernelGGL(k_noise, dim3(1792), dim3(256), 0, sn, big + (launches % 20) * per, out, per);
        CHECK(hipEventRecord(e1, sn));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (!noise) { struct timespec ts = {0, 20000000}; nanosleep(&ts, nullptr); CHECK(hipEventRecord(e1, sn)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); }
    }
    *stop = 1;
    CHECK(hipDeviceSynchronize());
    std::vector<stats> h(NPOLL);
    unsigned long long prog = 0;
    CHECK(hipMemcpy(h.data(), st, sizeof(stats) * NPOLL, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&prog, progress, 8, hipMemcpyDeviceToHost));
    stats tot{};
    for (auto &s : h) {
        tot.words += s.words; tot.over_10us += s.over_10us; tot.over_100us += s.over_100us; tot.over_1ms += s.over_1ms; tot.over_10ms += s.over_10ms;
        tot.lost += s.lost; tot.xcc_mask |= s.xcc_mask;
        if (s.max_ticks > tot.max_ticks) tot.max_ticks = s.max_ticks;
    }
    printf("lost_store: %.1f s, alloc kind %d, launch stream %s (%zu launches, %.1f us each): writer stored %llu words (%.2f us apart); pollers on XCD mask 0x%llx waited for %llu\n",
           ms / 1e3, alloc, noise ? "on" : "off", launches, launches ? ms * 1e3 / launches : 0.0, prog, prog ? ms * 1e3 / prog : 0.0, tot.xcc_mask, tot.words);
    printf("  waits > 10 us: %llu, > 100 us: %llu, > 1 ms: %llu, > 10 ms: %llu, given up after 300 ms (LOST): %llu; longest wait %.3f ms\n", tot.over_10us, tot.over_100us,
           tot.over_1ms, tot.over_10ms, tot.lost, tot.max_ticks / 1e5);
    for (int b = 0; b < NPOLL; b++)
        for (unsigned long long k = 0; k < h[b].lost && k < 8; k++) {
            unsigned long long now = 0;
            CHECK(hipMemcpy(&now, ring + (h[b].lost_idx[k] % RING) * STRIDE, 8, hipMemcpyDeviceToHost));
            printf("  block %d gave up on word %llu (last seen %llx); memory holds %llx after the kernels ended\n", b, h[b].lost_idx[k], h[b].lost_seen_after[k], now);
        }
    return 0;
}
