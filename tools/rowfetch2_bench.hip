// rowfetch2_bench.hip — how should ONE workgroup fetch N scattered Gram rows of 512 entries (the fold of k_chain_group)?
// Round 5: halving the bytes of a row (int16 residuals instead of int32) did not shorten the fold when the rows were fetched the same way,
// one entry per lane and row; bringing them as whole rows by LDS-DMA made it longer. This measures the candidates side by side, cycles per
// batch of N rows, all requests of a batch in flight, alone on the device and beside a streaming kernel:
//   A  int32 rows, one dword per lane and row (the int32 band as the chain reads it: 8 wave-loads of 256 B per row)
//   B  int16 rows, one short per lane and row (8 wave-loads of 128 B per row)
//   C  int16 rows, a whole row per wave-load (global_load_dwordx4: 64 lanes x 16 B = 1 KiB), rows dealt over the 8 waves, staged through LDS
//      (ds_write_b128, barrier) and read back one short per thread and row
//   D  int16 rows, a whole row per LDS-DMA piece (global_load_lds_dwordx4), barrier, read back
//   E  int32 rows, two dwordx4 wave-loads per row, staged through LDS like C
//   hipcc --offload-arch=gfx950 -O3 tools/rowfetch2_bench.hip -o tools/rowfetch2_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE, int N>
__global__ __launch_bounds__(512) void k_fetch(const char *__restrict__ buf, size_t nrows, int iters, long long *out, int *sink, unsigned seed)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const size_t rowbytes = (MODE == 0 || MODE == 4) ? 2048 : 1024;
    long long tot = 0;
    int acc = 0;
    unsigned s = seed;
    for (int it = 0; it < iters; it++) {
        size_t rows[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            s = s * 1664525u + 1013904223u;
            rows[i] = (size_t)(((unsigned long long)s * nrows) >> 32);
        }
        __syncthreads();
        const long long t0 = clock64();
        if (MODE == 0) {
            int v[N];
#pragma unroll
            for (int i = 0; i < N; i++) v[i] = reinterpret_cast<const int *>(buf + rows[i] * rowbytes)[t];
#pragma unroll
            for (int i = 0; i < N; i++) acc += v[i];
        } else if (MODE == 1) {
            int v[N];
#pragma unroll
            for (int i = 0; i < N; i++) v[i] = reinterpret_cast<const short *>(buf + rows[i] * rowbytes)[t];
#pragma unroll
            for (int i = 0; i < N; i++) acc += v[i];
        } else if (MODE == 2) { // whole int16 row per wave-load, rows w, w + 8, ... by wave w
            v4i v[(N + 7) / 8];
#pragma unroll
            for (int k = 0; k < (N + 7) / 8; k++) {
                const int i = min(wave + 8 * k, N - 1);
                size_t r = rows[0];
#pragma unroll
                for (int x = 1; x < N; x++) r = (i == x) ? rows[x] : r;
                v[k] = reinterpret_cast<const v4i *>(buf + r * rowbytes)[lane];
            }
#pragma unroll
            for (int k = 0; k < (N + 7) / 8; k++) {
                const int i = min(wave + 8 * k, N - 1);
                *reinterpret_cast<v4i *>(smem + (size_t)i * 1024 + lane * 16) = v[k];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < N; i++) acc += reinterpret_cast<const short *>(smem + (size_t)i * 1024)[t];
        } else if (MODE == 3) { // LDS-DMA
#pragma unroll
            for (int k = 0; k < (N + 7) / 8; k++) {
                const int i = min(wave + 8 * k, N - 1);
                size_t r = rows[0];
#pragma unroll
                for (int x = 1; x < N; x++) r = (i == x) ? rows[x] : r;
                const char *rp = buf + r * rowbytes;
                const unsigned long long u = (unsigned long long)(uintptr_t)rp;
                const char *up = reinterpret_cast<const char *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                                                            (unsigned)__builtin_amdgcn_readfirstlane((int)u)));
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)smem + (unsigned)i * 1024u));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"((unsigned)lane * 16u), "s"(up), "s"(dst)
                             : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int i = 0; i < N; i++) acc += reinterpret_cast<const short *>(smem + (size_t)i * 1024)[t];
        } else { // MODE 4: int32 rows, two dwordx4 wave-loads per row
            v4i v[2 * ((N + 7) / 8)];
#pragma unroll
            for (int k = 0; k < (N + 7) / 8; k++) {
                const int i = min(wave + 8 * k, N - 1);
                size_t r = rows[0];
#pragma unroll
                for (int x = 1; x < N; x++) r = (i == x) ? rows[x] : r;
                v[2 * k] = reinterpret_cast<const v4i *>(buf + r * rowbytes)[lane];
                v[2 * k + 1] = reinterpret_cast<const v4i *>(buf + r * rowbytes + 1024)[lane];
            }
#pragma unroll
            for (int k = 0; k < (N + 7) / 8; k++) {
                const int i = min(wave + 8 * k, N - 1);
                *reinterpret_cast<v4i *>(smem + (size_t)i * 2048 + lane * 16) = v[2 * k];
                *reinterpret_cast<v4i *>(smem + (size_t)i * 2048 + 1024 + lane * 16) = v[2 * k + 1];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < N; i++) acc += reinterpret_cast<const int *>(smem + (size_t)i * 2048)[t];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        tot += clock64() - t0;
    }
    if (t == 0) out[0] = tot / iters;
    if (acc == 0x12345678) sink[0] = acc;
}

__global__ void k_stream(const int4 *__restrict__ a, size_t n, int reps, int *sink)
{
    int acc = 0;
    for (int r = 0; r < reps; r++)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const int4 v = a[i];
            acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 0x12345678) sink[0] = acc;
}

template <int MODE, int N>
static void run(const char *buf, size_t bytes, long long *dout, int *sink, hipStream_t st, const char *tag)
{
    const size_t rowbytes = (MODE == 0 || MODE == 4) ? 2048 : 1024;
    const char *names[] = {"A int32, dword per lane", "B int16, short per lane", "C int16, whole row per wave-load + LDS", "D int16, whole row by LDS-DMA", "E int32, 2 x dwordx4 per row + LDS"};
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fetch<MODE, N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((k_fetch<MODE, N>), dim3(1), dim3(512), 140 * 1024, st, buf, bytes / rowbytes, 200, dout, sink, 12345u + N);
    CK(hipStreamSynchronize(st));
    long long h;
    CK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
    printf("  %-7s %-40s N=%2d rows: %7lld cycles per batch (%6.0f per row)\n", tag, names[MODE], N, h, (double)h / N);
}

template <int N>
static void all_modes(const char *buf, size_t bytes, long long *dout, int *sink, hipStream_t st, const char *tag)
{
    run<0, N>(buf, bytes, dout, sink, st, tag);
    run<1, N>(buf, bytes, dout, sink, st, tag);
    run<2, N>(buf, bytes, dout, sink, st, tag);
    run<3, N>(buf, bytes, dout, sink, st, tag);
    run<4, N>(buf, bytes, dout, sink, st, tag);
}

int main()
{
    const size_t bytes = (size_t)3 << 30;
    char *buf;
    int *sink;
    long long *dout;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&dout, 8));
    int4 *sbuf;
    const size_t sbytes = (size_t)4 << 30;
    CK(hipMalloc(&sbuf, sbytes));
    CK(hipMemset(sbuf, 0, sbytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int loaded = 0; loaded < 2; loaded++) {
        if (loaded) hipLaunchKernelGGL(k_stream, dim3(255 * 8), dim3(256), 0, s2, sbuf, sbytes / 16, 200, sink);
        const char *tag = loaded ? "loaded" : "alone";
        all_modes<16>(buf, bytes, dout, sink, s1, tag);
        all_modes<32>(buf, bytes, dout, sink, s1, tag);
        all_modes<64>(buf, bytes, dout, sink, s1, tag);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
