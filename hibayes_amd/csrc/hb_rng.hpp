// hb_rng.hpp — counter-based draws shared by the device kernels and the host MCMC loop.
//
// Generator: rocRAND's Philox4x32-10 engine (rocrand_philox4x32_10.h, host+device).  Draws are
// ADDRESSED, not streamed: block `blk` of stream `sub` under `seed` is
//     rocrand_init(seed, /*subsequence*/ sub, /*offset*/ 4*blk)  ->  rocrand4()
// i.e. Philox counter {blk, sub}, key seed.  Stream layout:
//     sub = (purpose << 56) | iter
//     purpose 1, marker stream: blk = global_marker * 64 + b
//         b=0 inclusion uniform, b=1 effect normal, b=2/3 BayesL inverse-Gaussian normal/uniform,
//         b=4+2a / 5+2a normal/uniform of gamma attempt a (per-marker chi^2 of BayesA/B)
//     purpose 2, host stream:   blk = running counter within the iteration, consumed in the
//         reference's draw order (src/Bayes.cpp:480 intercept, :490 covariates, :505/:511 random
//         effects, :713/:716 varg and Pi, :823 vare)
//     purpose 3, synthetic genotypes
// The reference draws the same quantities from R's global Mersenne-Twister (src/stats.cpp:3-24).
#pragma once
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <math.h>
#include <stdint.h>

#define HB_PURPOSE_MARKER 1ull
#define HB_PURPOSE_HOST 2ull
#define HB_PURPOSE_DATA 3ull
#define HB_BLK_PER_MARKER 64ull

__host__ __device__ inline uint4 hb_block(uint64_t seed, uint64_t sub, uint64_t blk)
{
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, sub, 4ull * blk, &st);
    return rocrand4(&st);
}

// 53-bit uniform strictly inside (0,1)
__host__ __device__ inline double hb_u53(uint32_t whi, uint32_t wlo)
{
    return ((double)(whi >> 5) * 67108864.0 + (double)(wlo >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

__host__ __device__ inline double hb_uniform_blk(uint64_t seed, uint64_t sub, uint64_t blk)
{
    uint4 w = hb_block(seed, sub, blk);
    return hb_u53(w.x, w.y);
}

// Box-Muller on one block (cosine branch)
__host__ __device__ inline double hb_normal_blk(uint64_t seed, uint64_t sub, uint64_t blk)
{
    uint4 w = hb_block(seed, sub, blk);
    double u1 = hb_u53(w.x, w.y);
    double u2 = hb_u53(w.z, w.w);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

// Sequential view of one stream: each draw consumes whole blocks.
struct hb_stream {
    uint64_t seed, sub, blk;
    __host__ __device__ hb_stream(uint64_t s, uint64_t su, uint64_t b) : seed(s), sub(su), blk(b) {}
    __host__ __device__ double unif() { return hb_uniform_blk(seed, sub, blk++); }
    __host__ __device__ double norm() { return hb_normal_blk(seed, sub, blk++); }
    // Marsaglia & Tsang (2000); stands where the reference calls R::rgamma (src/stats.cpp:13-15).
    // One attempt = one normal then one uniform; shape < 1 boosted with one more uniform.
    __host__ __device__ double gamma(double shape, double scale)
    {
        double a = shape < 1.0 ? shape + 1.0 : shape;
        double d = a - 1.0 / 3.0;
        double c = 1.0 / sqrt(9.0 * d);
        double x, v, u;
        for (;;) {
            x = norm();
            u = unif();
            v = 1.0 + c * x;
            if (v <= 0.0) continue;
            v = v * v * v;
            if (u < 1.0 - 0.0331 * (x * x) * (x * x)) break;
            if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) break;
        }
        double out = d * v;
        if (shape < 1.0) {
            u = unif();
            out *= pow(u, 1.0 / shape);
        }
        return out * scale;
    }
    // src/stats.cpp:22-24
    __host__ __device__ double chisq(double df) { return gamma(0.5 * df, 2.0); }
    // src/stats.cpp:55-67
    __host__ __device__ double invgauss(double mu, double lambda)
    {
        double z = norm();
        double y = z * z;
        double x = mu + 0.5 * mu * mu * y / lambda -
                   0.5 * (mu / lambda) * sqrt(4.0 * mu * lambda * y + mu * mu * y * y);
        double u = unif();
        if (u <= mu / (mu + x)) return x;
        return mu * mu / x;
    }
};

__host__ __device__ inline uint64_t hb_sub(uint64_t purpose, uint64_t iter)
{
    return (purpose << 56) | (iter & 0x00ffffffffffffffull);
}
