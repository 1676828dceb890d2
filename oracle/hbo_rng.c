/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see hbo_rng.h for scope and pinning status).
 */
#include "hbo_rng.h"
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * Mersenne-Twister, seeded the way R's set.seed() does it.
 * Third-party algorithm (R is not under /root/reference): R src/main/RNG.c
 *   RNG_Init():  50 warm-up steps of seed = 69069*seed+1, then 625 words i_seed[j] = seed
 *                after one more LCG step each; FixupSeeds() forces i_seed[0] (= mti) to 624.
 *   MT_genrand(): standard MT19937 tempering, scaled by 2.3283064365386963e-10.
 *   fixup():      keeps the result strictly inside (0,1).
 * ------------------------------------------------------------------------------------ */
#define MT_N 624
#define MT_M 397

void hbo_mt_set_seed(hbo_mt_t *s, uint32_t seed)
{
    for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;
    /* i_seed[0] is the position word (overwritten with 624), i_seed[1..624] the state */
    seed = 69069u * seed + 1u; /* would be dummy[0] */
    for (int j = 0; j < MT_N; j++) {
        seed = 69069u * seed + 1u;
        s->mt[j] = seed;
    }
    s->mti = MT_N;
}

static uint32_t mt_next(hbo_mt_t *s)
{
    static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
    uint32_t y;
    if (s->mti >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (s->mt[kk] & 0x80000000u) | (s->mt[kk + 1] & 0x7fffffffu);
            s->mt[kk] = s->mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 0x1u];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (s->mt[kk] & 0x80000000u) | (s->mt[kk + 1] & 0x7fffffffu);
            s->mt[kk] = s->mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 0x1u];
        }
        y = (s->mt[MT_N - 1] & 0x80000000u) | (s->mt[0] & 0x7fffffffu);
        s->mt[MT_N - 1] = s->mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 0x1u];
        s->mti = 0;
    }
    y = s->mt[s->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

double hbo_mt_unif_rand(hbo_mt_t *s)
{
    const double i2_32m1 = 2.328306437080797e-10;
    double x = (double)mt_next(s) * 2.3283064365386963e-10;
    if (x <= 0.0) return 0.5 * i2_32m1;
    if ((1.0 - x) <= 0.0) return 1.0 - 0.5 * i2_32m1;
    return x;
}

/* Wichura (1988) Algorithm AS241, PPND16 — what R's qnorm5() evaluates. */
double hbo_qnorm(double p)
{
    double q = p - 0.5, r, val;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r +
                        67265.770927008700853) * r + 45921.953931549871457) * r +
                      13731.693765509461125) * r + 1971.5909503065514427) * r +
                    133.14166789178437745) * r + 3.387132872796366608) /
              (((((((r * 5226.495278852854561 + 28729.085735721942674) * r +
                    39307.89580009271061) * r + 21213.794301586595867) * r +
                  5394.1960214247511077) * r + 687.1870074920579083) * r +
                42.313330701600911252) * r + 1.0);
        return val;
    }
    r = (q < 0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    if (r <= 5.0) {
        r -= 1.6;
        val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r +
                    0.24178072517745061177) * r + 1.27045825245236838258) * r +
                  3.64784832476320460504) * r + 5.7694972214606914055) * r +
                4.6303378461565452959) * r + 1.42343711074968357734) /
              (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r +
                    0.0151986665636164571966) * r + 0.14810397642748007459) * r +
                  0.68976733498510000455) * r + 1.6763848301838038494) * r +
                2.05319162663775882187) * r + 1.0);
    } else {
        r -= 5.0;
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r +
                    0.0012426609473880784386) * r + 0.026532189526576123093) * r +
                  0.29656057182850489123) * r + 1.7848265399172913358) * r +
                5.4637849111641143699) * r + 6.6579046435011037772) /
              (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r +
                    1.8463183175100546818e-5) * r + 7.868691311456132591e-4) * r +
                  0.0148753612908506148525) * r + 0.13692988092273580531) * r +
                0.59983220655588793769) * r + 1.0);
    }
    if (q < 0.0) val = -val;
    return val;
}

double hbo_mt_norm_rand(hbo_mt_t *s)
{
    const double BIG = 134217728.0; /* 2^27 */
    double u = hbo_mt_unif_rand(s);
    u = (double)(int)(BIG * u) + hbo_mt_unif_rand(s);
    return hbo_qnorm(u / BIG);
}

/* ------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
 * 1, 2, 3", SC'11).  Third-party algorithm: rocRAND 3.x (ROCm 7.2,
 * /opt/rocm/include/rocrand/rocrand_philox4x32_10.h) implements the same rounds; its
 * rocrand_init(seed, subsequence, offset) maps to counter = {offset/4, subsequence},
 * key = seed, which hbo_philox_block() reproduces.
 * ------------------------------------------------------------------------------------ */
void hbo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void hbo_philox_block(uint64_t seed, uint64_t sub, uint64_t blk, uint32_t out[4])
{
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)sub, (uint32_t)(sub >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    hbo_philox4x32_10(ctr, key, out);
}

double hbo_u53(uint32_t whi, uint32_t wlo)
{
    return ((double)(whi >> 5) * 67108864.0 + (double)(wlo >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

double hbo_philox_uniform(uint64_t seed, uint64_t sub, uint64_t blk)
{
    uint32_t w[4];
    hbo_philox_block(seed, sub, blk, w);
    return hbo_u53(w[0], w[1]);
}

double hbo_philox_normal(uint64_t seed, uint64_t sub, uint64_t blk)
{
    uint32_t w[4];
    hbo_philox_block(seed, sub, blk, w);
    double u1 = hbo_u53(w[0], w[1]);
    double u2 = hbo_u53(w[2], w[3]);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

/* ------------------------------------------------------------------------------------ */
void hbo_stream_init_r(hbo_stream_t *s, uint32_t seed)
{
    memset(s, 0, sizeof(*s));
    s->kind = HBO_RNG_R;
    hbo_mt_set_seed(&s->mt, seed);
}

void hbo_stream_init_philox(hbo_stream_t *s, uint64_t seed, uint64_t sub, uint64_t blk0)
{
    memset(s, 0, sizeof(*s));
    s->kind = HBO_RNG_PHILOX;
    s->seed = seed;
    s->sub = sub;
    s->blk = blk0;
}

double hbo_unif(hbo_stream_t *s)
{
    if (s->kind == HBO_RNG_R) return hbo_mt_unif_rand(&s->mt);
    return hbo_philox_uniform(s->seed, s->sub, s->blk++);
}

double hbo_norm(hbo_stream_t *s)
{
    if (s->kind == HBO_RNG_R) return hbo_mt_norm_rand(&s->mt);
    return hbo_philox_normal(s->seed, s->sub, s->blk++);
}

/* ------------------------------------------------------------------------------------
 * R's exp_rand() and rgamma().  Third-party algorithms (R is not under /root/reference; DESCRIPTION:35 pins
 * "R (>= 3.3.0)"; the files below have not changed their arithmetic since R 2.x):
 *   R src/nmath/sexp.c    Ahrens & Dieter (1972), "Computer methods for sampling from the exponential and
 *                         normal distributions", CACM 15: algorithm SA, q[k] = sum_{i<=k} ln(2)^i / i!
 *   R src/nmath/rgamma.c  a >= 1: Ahrens & Dieter (1982), "Generating gamma variates by a modified rejection
 *                         technique", CACM 25: algorithm GD (with R's expm1 in step 11);
 *                         a <  1: Ahrens & Dieter (1974), "Computer methods for sampling from gamma, beta, Poisson
 *                         and binomial distributions", Computing 12: algorithm GS.
 * Call sites in the reference: R::rgamma at src/stats.cpp:13-15 (gamma_sample), R::rchisq at :22-24
 * (= rgamma(df/2, 2.0), R src/nmath/rchisq.c).  Draw order matters for stream parity: GD consumes
 * norm_rand, [unif_rand, [exp_rand, unif_rand]*]; GS consumes [unif_rand, exp_rand]*.
 * Pinned by: set.seed(1); rexp(3) = 0.7551818 1.1816428 0.1457067 (R documentation) for exp_rand, the
 * immediate-acceptance identity rgamma = (sqrt(a - 1/2) + norm_rand/2)^2 on a seed whose first normal is
 * positive, and Kolmogorov-Smirnov tests against scipy's gamma CDF (tests/test_oracle_rng.py).
 * ------------------------------------------------------------------------------------ */
double hbo_mt_exp_rand(hbo_mt_t *s)
{
    static const double q[] = {0.6931471805599453, 0.9333736875190459, 0.9888777961838675, 0.9984589039328340,
                               0.9998292811061389, 0.9999833164100727, 0.9999985691438767, 0.9999998906925558,
                               0.9999999924734159, 0.9999999995283275, 0.9999999999728814, 0.9999999999985598,
                               0.9999999999999289, 0.9999999999999968, 0.9999999999999999, 1.0000000000000000};
    double a = 0.0;
    double u = hbo_mt_unif_rand(s);
    while (u <= 0.0 || u >= 1.0) u = hbo_mt_unif_rand(s);
    for (;;) {
        u += u;
        if (u > 1.0) break;
        a += q[0];
    }
    u -= 1.0;
    if (u <= q[0]) return a + u;
    int i = 0;
    double ustar = hbo_mt_unif_rand(s), umin = ustar;
    do {
        ustar = hbo_mt_unif_rand(s);
        if (umin > ustar) umin = ustar;
        i++;
    } while (u > q[i]);
    return a + umin * q[0];
}

double hbo_mt_rgamma(hbo_mt_t *st, double a, double scale)
{
    const double sqrt32 = 5.656854, exp_m1 = 0.36787944117144233;
    const double q1 = 0.04166669, q2 = 0.02083148, q3 = 0.00801191, q4 = 0.00144121, q5 = -7.388e-5,
                 q6 = 2.4511e-4, q7 = 2.424e-4;
    const double a1 = 0.3333333, a2 = -0.250003, a3 = 0.2000062, a4 = -0.1662921, a5 = 0.1423657,
                 a6 = -0.1367177, a7 = 0.1233795;
    double e, p, q, r, t, u, v, w, x, ret_val;
    if (!(a > 0.0) || !(scale > 0.0)) return (a == 0.0 || scale == 0.0) ? 0.0 : NAN;
    if (a < 1.0) { /* GS */
        e = 1.0 + exp_m1 * a;
        for (;;) {
            p = e * hbo_mt_unif_rand(st);
            if (p >= 1.0) {
                x = -log((e - p) / a);
                if (hbo_mt_exp_rand(st) >= (1.0 - a) * log(x)) break;
            } else {
                x = exp(log(p) / a);
                if (hbo_mt_exp_rand(st) >= x) break;
            }
        }
        return scale * x;
    }
    /* GD. (R caches s2, s, d, q0, b, si, c between calls with the same a; recomputing gives the same numbers.) */
    const double s2 = a - 0.5, s = sqrt(s2), d = sqrt32 - s * 12.0;
    t = hbo_mt_norm_rand(st);               /* step 2: immediate acceptance */
    x = s + 0.5 * t;
    ret_val = x * x;
    if (t >= 0.0) return scale * ret_val;
    u = hbo_mt_unif_rand(st);               /* step 3: squeeze acceptance */
    if (d * u <= t * t * t) return scale * ret_val;
    r = 1.0 / a;                            /* step 4 */
    const double q0 = ((((((q7 * r + q6) * r + q5) * r + q4) * r + q3) * r + q2) * r + q1) * r;
    double b, si, c;
    if (a <= 3.686) {
        b = 0.463 + s + 0.178 * s2;
        si = 1.235;
        c = 0.195 / s - 0.079 + 0.16 * s;
    } else if (a <= 13.022) {
        b = 1.654 + 0.0076 * s2;
        si = 1.68 / s + 0.275;
        c = 0.062 / s + 0.024;
    } else {
        b = 1.77;
        si = 0.75;
        c = 0.1515 / s;
    }
    if (x > 0.0) {                          /* steps 5-7: quotient acceptance */
        v = t / (s + s);
        if (fabs(v) <= 0.25)
            q = q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
        else
            q = q0 - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
        if (log(1.0 - u) <= q) return scale * ret_val;
    }
    for (;;) {                              /* steps 8-11: double-exponential hat */
        e = hbo_mt_exp_rand(st);
        u = hbo_mt_unif_rand(st);
        u = u + u - 1.0;
        t = (u < 0.0) ? b - si * e : b + si * e;
        if (t >= -0.71874483771719) {
            v = t / (s + s);
            if (fabs(v) <= 0.25)
                q = q0 + 0.5 * t * t * ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
            else
                q = q0 - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
            if (q > 0.0) {
                w = expm1(q);
                if (c * fabs(u) <= w * exp(e - 0.5 * t * t)) break;
            }
        }
    }
    x = s + 0.5 * t;
    return scale * x * x;
}

/* Gamma deviates of the unified stream.
 * R kind:      R's own rgamma (above) on the Mersenne-Twister stream — what R::rgamma at reference
 *              src/stats.cpp:13-15 consumes, draw for draw.
 * Philox kind: Marsaglia & Tsang (2000), "A simple method for generating gamma variables" — the contract shared with
 *              the device path (hb_rng.hpp): one attempt consumes one normal then one uniform; shape < 1 uses the
 *              boost g(a+1) * U^(1/a) with the extra uniform drawn after the accepted attempt. */
double hbo_gamma(hbo_stream_t *s, double shape, double scale)
{
    if (s->kind == HBO_RNG_R) return hbo_mt_rgamma(&s->mt, shape, scale);
    double a = shape < 1.0 ? shape + 1.0 : shape;
    double d = a - 1.0 / 3.0;
    double c = 1.0 / sqrt(9.0 * d);
    double x, v, u, out;
    for (;;) {
        x = hbo_norm(s);
        u = hbo_unif(s);
        v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (u < 1.0 - 0.0331 * (x * x) * (x * x)) break;
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) break;
    }
    out = d * v;
    if (shape < 1.0) {
        u = hbo_unif(s);
        out *= pow(u, 1.0 / shape);
    }
    return out * scale;
}

/* reference src/stats.cpp:22-24: R::rchisq(df) == rgamma(df/2, scale 2) */
double hbo_chisq(hbo_stream_t *s, double df)
{
    return hbo_gamma(s, 0.5 * df, 2.0);
}

/* reference src/stats.cpp:55-67 (Michael, Schucany & Haas): one normal, then one uniform */
double hbo_invgauss(hbo_stream_t *s, double mu, double lambda)
{
    double z = hbo_norm(s);
    double y = z * z;
    double x = mu + 0.5 * mu * mu * y / lambda -
               0.5 * (mu / lambda) * sqrt(4.0 * mu * lambda * y + mu * mu * y * y);
    double u = hbo_unif(s);
    if (u <= mu / (mu + x)) return x;
    return mu * mu / x;
}
