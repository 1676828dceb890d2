// tools/dotq_bench.hip — development harness for the exact fixed-point mat-vec (k_dotq), standalone.
//   hipcc -O3 --offload-arch=gfx950 tools/dotq_bench.hip -o /tmp/dotq_bench && /tmp/dotq_bench [n m cols_per_launch NS]
// Streams a 25 GB int8 genotype matrix through the kernel in launches of `cols_per_launch` columns (as the sweep does),
// checks a few hundred columns against a plain int64 reference and prints us per launch and TB/s.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int ND = 7;          // int8 digits of the fixed-point residual
constexpr int NDMA = 16;       // LDS-DMA instructions per stage
constexpr int SLOT = 1040;     // bytes of LDS per DMA instruction (1024 + 16: rotates the banks by 4 per instruction)
constexpr int STAGE = NDMA * SLOT;
constexpr int RS = 256;        // rows per stage

__device__ __forceinline__ void dma16(unsigned voff, const int8_t *sbase, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}

__device__ __forceinline__ void dma16nt(unsigned voff, const int8_t *sbase, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

__device__ __forceinline__ void sload(v4i &d, const int8_t *base, unsigned off)
{
    asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(d) : "s"(base), "s"(off) : "memory");
}

#define WAIT7(d) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d[0]), "+s"(d[1]), "+s"(d[2]), "+s"(d[3]), "+s"(d[4]), "+s"(d[5]), "+s"(d[6])::"memory")

__device__ __forceinline__ void mac(int (&acc)[ND], const v4i &x, const v4i (&d)[ND])
{
#pragma unroll
    for (int k = 0; k < ND; k++) {
        acc[k] = __builtin_amdgcn_sdot4(x.x, d[k].x, acc[k], false);
        acc[k] = __builtin_amdgcn_sdot4(x.y, d[k].y, acc[k], false);
        acc[k] = __builtin_amdgcn_sdot4(x.z, d[k].z, acc[k], false);
        acc[k] = __builtin_amdgcn_sdot4(x.w, d[k].w, acc[k], false);
    }
}

// one wave = 64 columns x NS stages of 256 rows; lane = column
template <int NBUF>
__global__ __launch_bounds__(64) void k_dotq(const int8_t *__restrict__ X, long ld, const int8_t *__restrict__ dig, int nstages,
                                             int NS, long long *__restrict__ acc64, long accstride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int cg = blockIdx.x;
    const int st0 = blockIdx.y * NS, st1 = min(nstages, st0 + NS);
    if (st0 >= st1) return;
    const int8_t *xg = X + (long)cg * 64 * ld;
    const unsigned voff = (unsigned)((lane >> 4) * ld + (lane & 15) * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const int8_t *dk[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) dk[k] = dig + (long)k * ld;

    auto issue = [&](int st, int b) {
        const int8_t *base = xg + (long)st * RS;
        const unsigned dst = lds0 + (unsigned)b * STAGE;
#pragma unroll
        for (int i = 0; i < NDMA; i++) dma16(voff, base + (long)(4 * i) * ld, dst + i * SLOT);
    };
    int acc[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) acc[k] = 0;

#pragma unroll
    for (int q = 0; q < NBUF - 1; q++)
        if (st0 + q < st1) issue(st0 + q, q);
    int b = 0;
    for (int st = st0; st < st1; ++st) {
        const int ahead = st1 - 1 - st; // stages after this one
        if (ahead >= NBUF - 1) issue(st + NBUF - 1, (b + NBUF - 1) % NBUF);
        // stage st has landed when at most min(ahead, NBUF-1) later stages are outstanding
        const int later = min(ahead, NBUF - 1);
        if (later == 0) wait_vm<0>();
        else if (later == 1) wait_vm<NDMA>();
        else wait_vm<2 * NDMA>();
        const v4i *px = reinterpret_cast<const v4i *>(smem + b * STAGE + (lane >> 2) * SLOT + (lane & 3) * 256);
        const unsigned roff = (unsigned)st * RS;
        v4i dA[ND], dB[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) sload(dA[k], dk[k], roff);
        v4i xa = px[0], xb;
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            WAIT7(dA);
#pragma unroll
            for (int k = 0; k < ND; k++) sload(dB[k], dk[k], roff + (s + 1) * 16);
            xb = px[s + 1];
            mac(acc, xa, dA);
            WAIT7(dB);
            if (s + 2 < 16) {
#pragma unroll
                for (int k = 0; k < ND; k++) sload(dA[k], dk[k], roff + (s + 2) * 16);
                xa = px[s + 2];
            }
            mac(acc, xb, dB);
        }
        b = (b + 1 == NBUF) ? 0 : b + 1;
    }
#pragma unroll
    for (int k = 0; k < ND; k++)
        __hip_atomic_fetch_add(acc64 + (long)k * accstride + cg * 64 + lane, (long long)acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- variant C: X tile AND the stage's digit planes arrive by LDS-DMA; digits are read back with wave-uniform
// (broadcast) ds_read_b128, the X column of each lane with a per-lane ds_read_b128. Nothing but DMA in the vmcnt queue. ----
template <int RSX, int NBUF, int CPL, int PAD>
__global__ __launch_bounds__(64) void k_dotq_l(const int8_t *__restrict__ X, long ld, const int8_t *__restrict__ dig, int nstages,
                                               int NS, long long *__restrict__ acc64, long accstride)
{
    constexpr int LPC = RSX / 16;          // lanes per column in one DMA instruction
    constexpr int CPI = 64 / LPC;          // columns (or digit planes) per DMA instruction
    constexpr int NX = 64 * CPL / CPI;     // DMA instructions for the X tile (64 * CPL columns)
    constexpr int NDG = (ND + CPI - 1) / CPI; // DMA instructions for the digit planes
    constexpr int SL = 1024 + PAD;
    constexpr int XB = NX * SL, BUF = XB + NDG * 1024, PER = NX + NDG, STEPS = RSX / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int cg = blockIdx.x;
    const int st0 = blockIdx.y * NS, st1 = min(nstages, st0 + NS);
    if (st0 >= st1) return;
    const int8_t *xg = X + (long)cg * 64 * CPL * ld;
    const unsigned voff = (unsigned)((lane / LPC) * ld + (lane % LPC) * 16);
    unsigned doff[NDG];
#pragma unroll
    for (int i = 0; i < NDG; i++) doff[i] = (unsigned)(min(i * CPI + lane / LPC, ND - 1) * ld + (lane % LPC) * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    int acc[CPL][ND];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int k = 0; k < ND; k++) acc[c][k] = 0;

    auto issue = [&](int st, int b) {
        const int8_t *base = xg + (long)st * RSX;
        const unsigned dst = lds0 + (unsigned)b * BUF;
#pragma unroll
        for (int i = 0; i < NX; i++) dma16nt(voff, base + (long)(CPI * i) * ld, dst + i * SL);
#pragma unroll
        for (int i = 0; i < NDG; i++) dma16(doff[i], dig + (long)st * RSX, dst + XB + i * 1024);
    };
#pragma unroll
    for (int q = 0; q < NBUF - 1; q++)
        if (st0 + q < st1) issue(st0 + q, q);
    int b = 0;
    for (int st = st0; st < st1; ++st) {
        const int ahead = st1 - 1 - st;
        if (ahead >= NBUF - 1) issue(st + NBUF - 1, (b + NBUF - 1) % NBUF);
        const int later = min(ahead, NBUF - 1);
        if (later == 0) wait_vm<0>();
        else if (later == 1) wait_vm<PER>();
        else if (later == 2) wait_vm<2 * PER>();
        else wait_vm<3 * PER>();
        const char *bp = smem + b * BUF;
        const char *pd = bp + XB;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            v4i x[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) { // lane's c-th column = 64 c + lane
                const int col = 64 * c + lane;
                x[c] = *reinterpret_cast<const v4i *>(bp + (col / CPI) * SL + (col % CPI) * RSX + s * 16);
            }
#pragma unroll
            for (int k = 0; k < ND; k++) {
                const v4i d = *reinterpret_cast<const v4i *>(pd + (k / CPI) * 1024 + (k % CPI) * RSX + s * 16);
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    acc[c][k] = __builtin_amdgcn_sdot4(x[c].x, d.x, acc[c][k], false);
                    acc[c][k] = __builtin_amdgcn_sdot4(x[c].y, d.y, acc[c][k], false);
                    acc[c][k] = __builtin_amdgcn_sdot4(x[c].z, d.z, acc[c][k], false);
                    acc[c][k] = __builtin_amdgcn_sdot4(x[c].w, d.w, acc[c][k], false);
                }
            }
        }
        b = (b + 1 == NBUF) ? 0 : b + 1;
    }
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int k = 0; k < ND; k++)
            __hip_atomic_fetch_add(acc64 + (long)k * accstride + cg * 64 * CPL + 64 * c + lane, (long long)acc[c][k], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_fill(int8_t *X, size_t nbytes, unsigned seed, int lim)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < nbytes / 4; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        unsigned out = 0;
        for (int b = 0; b < 4; b++) {
            int v = (int)((h >> (8 * b)) & 0xff);
            v = lim > 0 ? v % lim : (v - 128 == -128 ? -127 : v - 128);
            out |= ((unsigned)(v & 0xff)) << (8 * b);
        }
        reinterpret_cast<unsigned *>(X)[i] = out;
    }
}

__global__ void k_ref(const int8_t *X, long ld, const int8_t *dig, int n, int ncols, long long *out, long stride)
{
    const int c = blockIdx.x, k = blockIdx.y;
    __shared__ long long red[256];
    long long s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += (long long)X[(long)c * ld + i] * (long long)dig[(long)k * ld + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(long)k * stride + c] = red[0];
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 50000;
    const int m = argc > 2 ? atoi(argv[2]) : 500000;
    const int cpl = argc > 3 ? atoi(argv[3]) : 3072;
    const int NS = argc > 4 ? atoi(argv[4]) : 4;
    const int nbuf = argc > 5 ? atoi(argv[5]) : 2;
    const long ld = ((long)n + 255) / 256 * 256;
    const int m_pad = (m + cpl - 1) / cpl * cpl;
    const int rsx = argc > 6 ? atoi(argv[6]) : 256;
    const int nstages = (int)(ld / rsx);
    int8_t *X, *dig;
    long long *acc, *ref;
    CHECK(hipMalloc(&X, (size_t)ld * m_pad));
    CHECK(hipMalloc(&dig, (size_t)ld * ND));
    CHECK(hipMalloc(&acc, sizeof(long long) * ND * (size_t)m_pad));
    CHECK(hipMalloc(&ref, sizeof(long long) * ND * 512));
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, X, (size_t)ld * m_pad, 12345u, 3);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, dig, (size_t)ld * ND, 777u, 0);
    CHECK(hipMemset(acc, 0, sizeof(long long) * ND * (size_t)m_pad));
    CHECK(hipDeviceSynchronize());
    const int nsplit = (nstages + NS - 1) / NS;
    typedef void (*kfn)(const int8_t *, long, const int8_t *, int, int, long long *, long);
    const int cplane = argc > 7 ? atoi(argv[7]) : 1;
    const int pad = argc > 8 ? atoi(argv[8]) : 16;
    kfn kern = nullptr;
    size_t lds = 0;
#define SEL(R, B, C, P) if (rsx == R && nbuf == B && cplane == C && pad == P) { kern = k_dotq_l<R, B, C, P>; \
        lds = (size_t)B * ((64 * C / (1024 / R)) * (1024 + P) + ((ND + 1024 / R - 1) / (1024 / R)) * 1024); }
    SEL(128, 2, 1, 16) SEL(128, 3, 1, 16) SEL(128, 2, 2, 16) SEL(128, 2, 1, 0) SEL(128, 2, 1, 32) SEL(128, 2, 1, 64)
    SEL(256, 2, 1, 16) SEL(128, 2, 2, 32) SEL(128, 3, 2, 16) SEL(64, 2, 2, 16) SEL(64, 3, 2, 16) SEL(64, 4, 1, 16) SEL(64, 2, 1, 16)
    if (!kern) { printf("no such variant\n"); return 2; }
    const int colsper = 64 * cplane;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nlaunch = m_pad / cpl;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipMemset(acc, 0, sizeof(long long) * ND * (size_t)m_pad));
        CHECK(hipEventRecord(e0, 0));
        for (int g = 0; g < nlaunch; g++)
            hipLaunchKernelGGL(kern, dim3(cpl / colsper, nsplit), dim3(64), lds, 0, X + (size_t)g * cpl * ld, ld, dig, nstages, NS,
                               acc + (size_t)g * cpl, (long)m_pad);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %d launches of %d cols (grid %d x %d, NS=%d, nbuf=%d, RS=%d, cols/lane=%d, pad=%d): %.2f us/launch, %.3f TB/s (n*cols bytes)\n", rep, nlaunch,
               cpl, cpl / colsper, nsplit, NS, nbuf, rsx, cplane, pad, ms * 1e3 / nlaunch, (double)n * m_pad / (ms * 1e-3) / 1e12);
    }
    // check: first 256 and last 256 columns
    std::vector<long long> hacc((size_t)ND * m_pad), href((size_t)ND * 512);
    CHECK(hipMemcpy(hacc.data(), acc, sizeof(long long) * hacc.size(), hipMemcpyDeviceToHost));
    long bad = 0;
    for (int part = 0; part < 2; part++) {
        const int c0 = part == 0 ? 0 : m_pad - 256;
        hipLaunchKernelGGL(k_ref, dim3(256, ND), dim3(256), 0, 0, X + (size_t)c0 * ld, ld, dig, (int)ld, 256, ref + part * 256, 512L);
    }
    CHECK(hipMemcpy(href.data(), ref, sizeof(long long) * href.size(), hipMemcpyDeviceToHost));
    for (int part = 0; part < 2; part++)
        for (int k = 0; k < ND; k++)
            for (int c = 0; c < 256; c++) {
                const int col = (part == 0 ? 0 : m_pad - 256) + c;
                if (hacc[(size_t)k * m_pad + col] != href[(size_t)k * 512 + part * 256 + c]) {
                    if (bad < 5) printf("MISMATCH col %d k %d: %lld vs %lld\n", col, k, hacc[(size_t)k * m_pad + col], href[(size_t)k * 512 + part * 256 + c]);
                    bad++;
                }
            }
    printf("check: %ld mismatches of %d\n", bad, 2 * ND * 256);
    return bad != 0;
}
