// tools/row_trip.hip — what one "trip" of k_chain_group's fold costs a lone workgroup: 3 movers x 22 Gram rows of 512 int32, one
// word per lane and row (66 loads per lane in flight), rows of a mover 23 MB apart (one per band block) in a 16 GB buffer.
//   hipcc -O3 --offload-arch=gfx950 tools/row_trip.hip -o /tmp/row_trip && /tmp/row_trip [stream]
// Modes: cold (rows never touched), warm-L2 (the same rows requested by 16-lane line touches and waited for, then the trip),
// again (the trip repeated at once), near (the 66 rows contiguous), and the same with a streaming kernel on the other compute units.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int P = 512, NR = 22, CH = 3;
constexpr size_t PP = (size_t)P * P, PSTEP = 22 * PP;

__device__ inline const int32_t *rowptr(const int32_t *G, uint32_t h, int f, int i, int mode, size_t npanels)
{
    // mover f of trip h: panel and row from a hash; row i of the mover one band block further
    const uint32_t x = (h * 3u + (uint32_t)f) * 2654435761u;
    const size_t p = (x >> 8) % npanels, r = x & (P - 1);
    if (mode == 3) return G + p * PP + ((size_t)(f * NR + i)) * P; // contiguous rows
    return G + p * PP + (size_t)i * PSTEP + r * P;
}

template <int MODE>
__global__ __launch_bounds__(512) void k_trip(const int32_t *__restrict__ G, size_t npanels, int trips, long long *out, int *sink, int seed)
{
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    long long total = 0, total_pf = 0;
    int acc = 0;
    for (int h = 0; h < trips; h++) {
        const uint32_t hh = (uint32_t)(seed * 7919 + h);
        if (MODE == 1) { // touch every line of the 66 rows: 16 lanes per row, 4 rows per wave instruction
            const long long c0 = clock64();
            int pf[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int q = min((k * 8 + wave) * 4 + (lane >> 4), CH * NR - 1);
                pf[k] = rowptr(G, hh, q / NR, q % NR, 0, npanels)[(lane & 15) * 32];
            }
            acc += pf[0] + pf[1] + pf[2];
            __syncthreads();
            total_pf += clock64() - c0;
        }
        if (MODE >= 4) { // the same through LDS-DMA (no register destination), one lane per 128 / 64 / 32 bytes of every row
            __shared__ int junk[64];
            const int stride = MODE == 4 ? 32 : MODE == 5 ? 16 : 8, lpr = P / stride; // ints between lanes, lanes per row
            const int rpi = 64 / lpr, ninstr = (CH * NR + rpi - 1) / rpi;               // rows per wave instruction
            const long long c0 = clock64();
            for (int k = wave; k < ninstr; k += 8) {
                const int q = min(k * rpi + lane / lpr, CH * NR - 1);
                const int32_t *src = rowptr(G, hh, q / NR, q % NR, 0, npanels) + (lane % lpr) * stride;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src), "s"(__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)junk))
                             : "memory");
            }
            if (MODE < 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            total_pf += clock64() - c0;
        }
        for (int rep = 0; rep < (MODE == 2 ? 2 : 1); rep++) {
            __syncthreads();
            const long long c0 = clock64();
            int g[CH][NR];
#pragma unroll
            for (int f = 0; f < CH; f++)
#pragma unroll
                for (int i = 0; i < NR; i++) g[f][i] = rowptr(G, hh, f, i, MODE, npanels)[t];
            int s = 0;
#pragma unroll
            for (int f = 0; f < CH; f++)
#pragma unroll
                for (int i = 0; i < NR; i++) s += g[f][i];
            acc += s;
            __syncthreads();
            if (MODE != 2 || rep == 1) total += clock64() - c0;
        }
    }
    sink[t] = acc;
    if (t == 0) { out[0] = total; out[1] = total_pf; }
}

__global__ __launch_bounds__(256) void k_stream(const int4 *__restrict__ src, size_t n16, int *sink, const volatile int *stop)
{
    int a = 0;
    for (int round = 0; round < 100000 && !*stop; round++)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(src) + i);
            a += v.x + v.y + v.z + v.w;
        }
    if (a == 0x7fffffff) sink[0] = a;
}

static long long *out;  // (pinned host memory: read without a device-wide synchronisation while the streaming kernel runs)
static int *sink;
static hipStream_t st1;

template <int MODE>
int run(const char *name, const int32_t *G, size_t npanels, int seed)
{
    const int trips = 200;
    hipLaunchKernelGGL(k_trip<MODE>, dim3(1), dim3(512), 0, st1, G, npanels, trips, out, sink, seed);
    CHECK(hipStreamSynchronize(st1));
    const long long h[2] = {out[0], out[1]};
    printf("  %-34s %8.0f cycles per trip (135 KB: %.1f cycles per 128-byte line)", name, (double)h[0] / trips, (double)h[0] / trips / 1056.0);
    if (MODE == 1 || MODE >= 4) printf("   [line touches before it: %.0f cycles]", (double)h[1] / trips);
    printf("\n");
    return 0;
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)16 << 30;
    int32_t *G;
    CHECK(hipMalloc(&G, bytes + NR * PSTEP * 4));
    CHECK(hipMemset(G, 1, bytes + NR * PSTEP * 4));
    const size_t npanels = bytes / 4 / PP; // bases within the first 16 GB, rows reach up to 22 band blocks further
    const bool with_stream = argc > 1;
    int4 *S = nullptr;
    int *stop = nullptr, *ssink = nullptr;
    hipStream_t st2;
    CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&st1, hipStreamNonBlocking));
    CHECK(hipHostMalloc(&out, 16));
    CHECK(hipMalloc(&sink, 512 * 4));
    {
        const size_t sb = (size_t)8 << 30;
        CHECK(hipMalloc(&S, sb));
        CHECK(hipMemset(S, 0, sb));
        CHECK(hipHostMalloc(&stop, 4));
        *stop = 0;
        CHECK(hipMalloc(&ssink, 4));
        CHECK(hipDeviceSynchronize());
    }
    for (int pass = 0; pass < (with_stream ? 2 : 1); pass++) {
        if (pass == 1) {
            const size_t sb = (size_t)8 << 30;
            hipLaunchKernelGGL(k_stream, dim3(255 * 4), dim3(256), 0, st2, S, sb / 16, ssink, stop);
            printf("with a streaming kernel (255 x 4 workgroups reading 8 GB over and over) beside it:\n");
        } else
            printf("alone on the device:\n");
        if (run<0>("cold rows, 23 MB apart", G, npanels, 1 + 10 * pass)) return 1;
        if (run<1>("the same after line touches", G, npanels, 2 + 10 * pass)) return 1;
        if (run<2>("the trip repeated at once", G, npanels, 3 + 10 * pass)) return 1;
        if (run<3>("cold rows, contiguous", G, npanels, 4 + 10 * pass)) return 1;
        if (run<4>("after LDS-DMA touches, 128 B apart", G, npanels, 5 + 10 * pass)) return 1;
        if (run<5>("after LDS-DMA touches, 64 B apart", G, npanels, 6 + 10 * pass)) return 1;
        if (run<6>("after LDS-DMA touches, 32 B apart", G, npanels, 7 + 10 * pass)) return 1;
    }
    *stop = 1;
    CHECK(hipDeviceSynchronize());
    return 0;
}
