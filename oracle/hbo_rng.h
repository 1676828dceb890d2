/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker.
 *
 * Random-number layer of the CPU restatement of hibayes' individual-level sampler.
 *
 * The reference draws everything from R's global RNG through nmath
 * (reference src/stats.cpp:3-24, 55-76: unif_rand, norm_rand, R::rgamma, R::rchisq,
 * R::rnorm, R::runif).  R itself is a third-party dependency that is NOT under
 * /root/reference and is not installed in this image (DESCRIPTION:35 pins only
 * "R (>= 3.3.0)").  Two back-ends are restated here:
 *
 *  HBO_RNG_R      R's default generator after set.seed(): Mersenne-Twister (Matsumoto &
 *                 Nishimura 1998) with R's seed scrambling (R src/main/RNG.c: RNG_Init,
 *                 MT_genrand, fixup) and normal deviates by INVERSION (R src/nmath/snorm.c:
 *                 u = unif_rand(); u = (int)(2^27 u) + unif_rand(); qnorm5(u / 2^27)),
 *                 with qnorm5 = Wichura's AS241 PPND16.  Pinned by the published R outputs
 *                 set.seed(1); runif(3) and set.seed(123); rnorm(5) (tests/test_oracle_rng.py).
 *                 Gamma deviates are R's own rgamma (Ahrens-Dieter GD for a >= 1, GS for a < 1, with
 *                 R's exp_rand, R src/nmath/{rgamma,sexp,rchisq}.c), so this mode consumes the stream
 *                 exactly as R::rgamma / R::rchisq at reference src/stats.cpp:13-24 do; pinned by
 *                 set.seed(1); rexp(3) and distribution tests.
 *
 *  HBO_RNG_PHILOX counter-based Philox4x32-10 (Salmon et al., SC'11), the generator the
 *                 device path uses through rocRAND (rocrand_philox4x32_10.h: key = seed,
 *                 counter = {offset/4, subsequence}).  Draws are addressed, not streamed:
 *                 see hbo_philox_block() and the stream layout below.  Pinned by the
 *                 Random123 known-answer vectors (tests/test_oracle_rng.py).
 *
 * Stream layout for HBO_RNG_PHILOX (shared contract with the device path):
 *   key      = 64-bit user seed
 *   counter  = { blk_lo, blk_hi, sub_lo, sub_hi },  sub = (purpose << 56) | iter
 *   purpose 1 (marker stream):  blk = marker_global_index * 64 + b
 *        b = 0  inclusion uniform U        (w0,w1)
 *        b = 1  effect normal z            Box-Muller on (w0,w1),(w2,w3)
 *        b = 2  BayesL inverse-Gaussian normal,  b = 3  its uniform
 *        b = 4+2a, 5+2a   normal / uniform of gamma attempt a (BayesA/B per-marker chi^2)
 *   purpose 2 (host stream):    blk = running block counter inside the iteration,
 *        consumed in the reference's draw order (src/Bayes.cpp:480 ... :823)
 *   u53(w_hi, w_lo) = ((w_hi >> 5) * 2^26 + (w_lo >> 6) + 0.5) * 2^-53   in (0,1)
 *   normal(block)   = sqrt(-2 log u53(w0,w1)) * cos(2 pi u53(w2,w3))
 */
#ifndef HBO_RNG_H
#define HBO_RNG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { HBO_RNG_R = 0, HBO_RNG_PHILOX = 1 };

enum { HBO_PURPOSE_MARKER = 1, HBO_PURPOSE_HOST = 2, HBO_PURPOSE_DATA = 3 };
enum { HBO_BLK_PER_MARKER = 64 };

/* ---- Mersenne-Twister with R's seeding ---- */
typedef struct {
    uint32_t mt[624];
    int mti;
} hbo_mt_t;

void   hbo_mt_set_seed(hbo_mt_t *s, uint32_t seed);   /* == R's set.seed(seed), default kinds */
double hbo_mt_unif_rand(hbo_mt_t *s);                  /* == R's unif_rand()                   */
double hbo_mt_norm_rand(hbo_mt_t *s);                  /* == R's norm_rand(), INVERSION        */
double hbo_qnorm(double p);                            /* Wichura AS241 PPND16                 */
double hbo_mt_exp_rand(hbo_mt_t *s);                   /* == R's exp_rand()  (sexp.c)          */
double hbo_mt_rgamma(hbo_mt_t *s, double a, double scale); /* == R's rgamma() (Ahrens-Dieter)  */

/* ---- Philox4x32-10 ---- */
void   hbo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void   hbo_philox_block(uint64_t seed, uint64_t sub, uint64_t blk, uint32_t out[4]);
double hbo_u53(uint32_t whi, uint32_t wlo);
double hbo_philox_uniform(uint64_t seed, uint64_t sub, uint64_t blk);
double hbo_philox_normal(uint64_t seed, uint64_t sub, uint64_t blk);

/* ---- unified sequential stream used by the sampler for host-side draws ---- */
typedef struct {
    int      kind;
    hbo_mt_t mt;         /* HBO_RNG_R      */
    uint64_t seed;       /* HBO_RNG_PHILOX */
    uint64_t sub;
    uint64_t blk;
} hbo_stream_t;

void   hbo_stream_init_r(hbo_stream_t *s, uint32_t seed);
void   hbo_stream_init_philox(hbo_stream_t *s, uint64_t seed, uint64_t sub, uint64_t blk0);
double hbo_unif(hbo_stream_t *s);
double hbo_norm(hbo_stream_t *s);
double hbo_gamma(hbo_stream_t *s, double shape, double scale);  /* R kind: R's rgamma; Philox kind: Marsaglia-Tsang */
double hbo_chisq(hbo_stream_t *s, double df);                    /* gamma(df/2, 2)  */
double hbo_invgauss(hbo_stream_t *s, double mu, double lambda);  /* stats.cpp:55-67 */

#ifdef __cplusplus
}
#endif
#endif
