// What one workgroup pays for fetching N scattered 2-KiB rows (one dword per thread per row, all N loads in flight), the
// access pattern of the chain's forward fold / row-cache misses: cycles per batch, for N = 1..64, over a multi-GB buffer,
// alone on the device and with a streaming kernel saturating HBM beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/rowfetch_bench.hip -o tools/rowfetch_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int N>
__global__ __launch_bounds__(512) void k_fetch(const int *__restrict__ buf, size_t nrows, int iters, long long *out, int *sink, unsigned seed)
{
    const int t = threadIdx.x;
    long long tot = 0;
    int acc = 0;
    unsigned s = seed;
    for (int it = 0; it < iters; it++) {
        int v[N];
        size_t rows[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            s = s * 1664525u + 1013904223u;
            rows[i] = (size_t)(((unsigned long long)s * nrows) >> 32);
        }
        __syncthreads();
        const long long t0 = clock64();
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = buf[rows[i] * 512 + t];
#pragma unroll
        for (int i = 0; i < N; i++) acc += v[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// the same batch fetched twice: the second pass finds the rows in this XCD's L2 (the 32-KiB L1 holds an eighth of them)
template <int N>
__global__ __launch_bounds__(512) void k_refetch(const int *__restrict__ buf, size_t nrows, int iters, long long *out, int *sink, unsigned seed)
{
    const int t = threadIdx.x;
    long long tot = 0;
    int acc = 0;
    unsigned s = seed;
    for (int it = 0; it < iters; it++) {
        int v[N];
        size_t rows[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            s = s * 1664525u + 1013904223u;
            rows[i] = (size_t)(((unsigned long long)s * nrows) >> 32);
        }
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = buf[rows[i] * 512 + t];
#pragma unroll
        for (int i = 0; i < N; i++) acc += v[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const long long t0 = clock64();
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = __builtin_nontemporal_load(buf + rows[i] * 512 + t);
#pragma unroll
        for (int i = 0; i < N; i++) acc += v[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tot += clock64() - t0;
    }
    if (t == 0) out[0] = tot / iters;
    if (acc == 0x12345678) sink[0] = acc;
}

template <int N>
static void run2(const int *buf, size_t nrows, long long *dout, int *sink, hipStream_t st)
{
    hipLaunchKernelGGL(k_refetch<N>, dim3(1), dim3(512), 0, st, buf, nrows, 200, dout, sink, 777u + N);
    CK(hipStreamSynchronize(st));
    long long h;
    CK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
    printf("  L2-warm  N=%2d rows in flight: %7lld cycles per batch (%6.0f per row)\n", N, h, (double)h / N);
}

template <int N>
static void run(const int *buf, size_t nrows, long long *dout, int *sink, hipStream_t st, const char *tag)
{
    hipLaunchKernelGGL(k_fetch<N>, dim3(1), dim3(512), 0, st, buf, nrows, 200, dout, sink, 12345u + N);
    CK(hipStreamSynchronize(st));
    long long h;
    CK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
    printf("  %-8s N=%2d rows in flight: %7lld cycles per batch (%6.0f per row)\n", tag, N, h, (double)h / N);
}

int main()
{
    const size_t bytes = (size_t)3 << 30, nrows = bytes / 2048;
    int *buf, *sink;
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
    run2<1>(buf, nrows, dout, sink, s1);
    run2<8>(buf, nrows, dout, sink, s1);
    run2<32>(buf, nrows, dout, sink, s1);
    run2<64>(buf, nrows, dout, sink, s1);
    for (int loaded = 0; loaded < 2; loaded++) {
        if (loaded) hipLaunchKernelGGL(k_stream, dim3(255 * 8), dim3(256), 0, s2, sbuf, sbytes / 16, 60, sink);
        const char *tag = loaded ? "loaded" : "alone";
        run<1>(buf, nrows, dout, sink, s1, tag);
        run<2>(buf, nrows, dout, sink, s1, tag);
        run<4>(buf, nrows, dout, sink, s1, tag);
        run<8>(buf, nrows, dout, sink, s1, tag);
        run<16>(buf, nrows, dout, sink, s1, tag);
        run<32>(buf, nrows, dout, sink, s1, tag);
        run<64>(buf, nrows, dout, sink, s1, tag);
    }
    CK(hipDeviceSynchronize());
    // and what a plain streaming read of 4 GiB reaches on this device (the ceiling any mat-vec is measured against)
    for (int blocks : {256 * 4, 256 * 8, 256 * 16}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, s2, sbuf, sbytes / 16, 1, sink);
        CK(hipEventRecord(e0, s2));
        hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, s2, sbuf, sbytes / 16, 4, sink);
        CK(hipEventRecord(e1, s2));
        CK(hipStreamSynchronize(s2));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  streaming read, %5d blocks x 256: %.2f TB/s\n", blocks, 4.0 * sbytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
