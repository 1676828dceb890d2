// tools/dot4_rate.hip — issue rate of v_dot4_i32_i8 on gfx950 (what bounds a VALU-only int8 mat-vec).
//   hipcc -O3 --offload-arch=gfx950 tools/dot4_rate.hip -o /tmp/dot4_rate && /tmp/dot4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(int *out, int iters, int a0, int b0, long long *clk)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    int acc[16];
    int a = a0 + threadIdx.x, b = b0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_sdot4(a, b, acc[i], false);
            else if (MODE == 1) acc[i] = acc[i] * a + b;                       // v_mad / v_mul_lo + add
            else acc[i] = __builtin_amdgcn_sdot4((a >> (2 * (i & 3))) & 0x03030303, b, acc[i], false); // with the 2-bit expansion
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int MODE>
int run(const char *name, int blocks)
{
    int *out;
    CHECK(hipMalloc(&out, sizeof(int) * 64 * blocks));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    long long *clk;
    CHECK(hipMalloc(&clk, 16));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, 3, 5, (long long *)nullptr);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 3, 5, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * iters * 16; // wave-instructions of the measured kind
    printf("%-28s blocks %5d (%.1f waves/SIMD): %.3f ms, %.2f G wave-instr/s, %.2f cycles per wave-instr per SIMD at 2.4 GHz\n", name, blocks,
           blocks / 1024.0, ms, ops / ms * 1e-6, 2.4e9 / (ops / (ms * 1e-3) / 1024.0));
    long long h[2];
    CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    printf("        shader clock during the kernel: %.0f MHz (s_memtime ticks / 100 MHz wall ticks); cycles per wave-instr per SIMD at that clock: %.2f\n",
           100.0 * h[0] / h[1], (100e6 * h[0] / h[1]) / (ops / (ms * 1e-3) / 1024.0));
    CHECK(hipFree(clk));
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    for (int blocks : {1024, 2048, 4096, 8192}) {
        if (run<0>("v_dot4_i32_i8", blocks)) return 1;
        if (run<2>("shift+and+v_dot4_i32_i8", blocks)) return 1;
        if (run<1>("int mad", blocks)) return 1;
    }
    return 0;
}
