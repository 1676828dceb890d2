// hb_stats.hpp — marker statistics (reference src/Bayes.cpp:310-317), integer-exact.
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

// ---------------------------------------------------------------------------------------------
// marker statistics, reference src/Bayes.cpp:310-317 — integer-exact
// one workgroup per column
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stats(const int8_t *__restrict__ X, int64_t ld, int n, int m,
                                               double *__restrict__ xpx, double *__restrict__ vx,
                                               int *__restrict__ xinfo, double *__restrict__ s1out)
{
    __shared__ long long red[4];
    const int j = blockIdx.x;
    const int8_t *col = X + (int64_t)j * ld;
    long long s1 = 0, s2 = 0;
    int mn = 127, mx = -128;
    for (int64_t r0 = (int64_t)threadIdx.x * 16; r0 < ld; r0 += 256 * 16) {
        const int4 v = *reinterpret_cast<const int4 *>(col + r0);
        const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = (int)(int8_t)(w[q] >> (8 * b));
                if (r0 + q * 4 + b < n) {
                    s1 += x;
                    s2 += x * x;
                    mn = min(mn, x);
                    mx = max(mx, x);
                }
            }
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (j < m) {
        atomicMin(&xinfo[0], mn);
        atomicMax(&xinfo[1], mx);
    }
    if (threadIdx.x == 0) {
        if (s1out) s1out[j] = j < m ? (double)s1 : 0.0; // (row-sharded cross-check mode: the shards' integer sums are added up by the host)
        if (j < m) {
            xpx[j] = (double)s2;
            const long long num = (long long)n * s2 - s1 * s1; // n*S2 - S1^2, exact
            vx[j] = (num == 0 || n < 2) ? 0.0 : (double)num / ((double)n * (double)(n - 1));
        } else {
            xpx[j] = 0.0;
            vx[j] = 0.0;
        }
    }
}

