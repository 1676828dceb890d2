/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of hibayes' individual-level Gibbs sampler Bayes()
 * (reference src/Bayes.cpp:60-1094, v3.1.0) and of the PLINK .bed decode
 * (reference src/read_bed.cpp:98-167), written from the algorithm, in plain C.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the reported CPU baseline.  The shipped product
 * (hibayes_amd/, libhibayes_gpu.so) never links, imports or calls it.
 *
 * PARITY PIN STATUS: PINNED against a real hibayes run.  The reference has no test-suite and R/Rcpp/Armadillo
 * are not installed, so src/Bayes.cpp cannot be built here (it includes RcppArmadillo.h, R.h, Rmath.h and links
 * BLAS ddot_/daxpy_ — none present), but the reference's README.md:130-172 prints summary() of
 * ibrm(T1 ~ season + bwt + (1|loc) + (1|dam), BayesCpi, Pi = c(0.98, 0.02), 20000/16000/5, seed = 666666) on
 * inst/extdata/demo.*.  This restatement, drawing from R's stream (set.seed() Mersenne-Twister, inversion
 * normals, Ahrens-Dieter rgamma/exp_rand: hbo_rng.c), reproduces EVERY printed digit of that summary after
 * 20 000 iterations — Vg 52.10097 (SD 13.084), h2 0.35748, pi 0.92683, Ve 30.77, Vr 8.10 / 54.29, the four
 * fixed effects and their SDs, the intercept, the residual and marker-effect quantiles
 * (tests/test_oracle_sampler.py::test_oracle_reproduces_the_fit_printed_in_the_reference_readme).  A chain of
 * 20 000 x ~1 400 sequential draws only does that if every draw is consumed in the reference's order and every
 * conditional is restated exactly.  That run exercises BayesCpi, the intercept / covariate / random-effect
 * blocks, the variance and pi draws and the posterior assembly; the other five sweeps (RR, A, B, L, R) share
 * the scalar samplers and loop skeleton it pins and are restated line by line from src/Bayes.cpp:587-815.
 * Further pins: the .bed decode against README.md:81-86; the initial-state facts derived from
 * src/Bayes.cpp:310-363; R's published set.seed()/runif()/rnorm()/rexp() outputs and the
 * Random123 Philox vectors for the RNG layer.
 */
#ifndef HB_ORACLE_H
#define HB_ORACLE_H
#include <stdint.h>
#include "hbo_rng.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HBO_MAX_FOLD 16

/* model_index as at src/Bayes.cpp:97 */
enum { HBO_RR = 1, HBO_A = 2, HBO_B = 3, HBO_C = 4, HBO_L = 5, HBO_R = 6 };

/* warm hyper-parameter state (mirrors hb_warm_state of include/hibayes_gpu.h field for field; not in the reference): the scalars
 * the loop at src/Bayes.cpp:477 carries between iterations, for a chain CONTINUED from (g_init, warm) */
typedef struct {
    double mu, vare, varg, lambda2;
    double pi[8];
    const double *vargL;      /* BayesL: m, or NULL */
} hbo_warm;

typedef struct {
    /* --- data: mirrors the Bayes() argument list, src/Bayes.cpp:60-88 --- */
    int32_t n, m;
    const double *y;          /* n */
    const double *X;          /* n x m column-major double (the reference's layout), or NULL */
    const int8_t *X8;         /* n x m column-major int8 alternative (ld = n), used when X == NULL */
    const char *model;        /* "BayesRR","BayesA","BayesB","BayesBpi","BayesC","BayesCpi","BayesL","BayesR" */
    const double *Pi;         /* n_pi */
    int32_t n_pi;
    const double *fold;       /* n_fold or NULL */
    int32_t n_fold;
    const double *C;          /* n x nc column-major or NULL */
    int32_t nc;
    const char *const *R;     /* n x nr column-major C strings or NULL */
    int32_t nr;
    int32_t niter, nburn, thin;
    double dfvr, s2vr, vg, dfvg, s2vg, ve, dfve, s2ve;   /* NaN == R_NilValue */
    const uint32_t *windindx; /* m, 1-based window ids, or NULL */
    int32_t threads;
    /* --- RNG --- */
    int32_t rng_kind;         /* HBO_RNG_R | HBO_RNG_PHILOX */
    uint64_t seed;
    int64_t marker_offset;    /* global index of local marker 0 (Philox marker stream) */
    /* --- optional trace of one iteration's marker sweep (each m long, or NULL) --- */
    int32_t trace_iter;
    double *trace_rhs;
    int32_t *trace_cls;
    double *trace_g;
    /* warm start (mirrors hb_bayes_args.g_init of the GPU library; not in the reference): m effects the chain
     * starts from; entries of monomorphic markers are taken as 0 */
    const double *g_init;
    const hbo_warm *warm;     /* NULL = the reference's start (:319-374, :469) */
} hbo_args;

typedef struct {
    /* scalars */
    double Vg, Ve, h2, mu;
    int32_t n_records, nzct, nw, n_levels;
    /* caller-allocated (NULL allowed = not wanted) */
    double *beta;        /* nc */
    double *alpha;       /* m */
    double *pi;          /* n_pi */
    double *Vr;          /* nr */
    double *r_est;       /* n_levels (sum over terms), order = sorted levels per term */
    double *g;           /* n   (final-iteration u, src/Bayes.cpp:1023) */
    double *e;           /* n */
    double *pip;         /* m */
    double *gwas;        /* nw */
    /* MCMC samples, each n_records long per row */
    double *s_Vg, *s_Ve, *s_h2, *s_mu;
    double *s_beta;      /* nc x n_records col-major */
    double *s_alpha;     /* m x n_records col-major (may be NULL) */
    double *s_pi;        /* n_pi x n_records */
    double *s_Vr;        /* nr x n_records */
    /* initial-state facts (src/Bayes.cpp:310-363) for pinning */
    double vary, sumvx, varg0, s2varg, vara0, s2vara, vare0, lambda2_0, rate0;
    int32_t nvar0;
    double *xpx;         /* m */
    double *vx;          /* m */
    /* timing of the MCMC loop only (seconds) and sweeps done */
    double loop_seconds;
    int32_t iters_done;
    char error[256];
    /* the chain after its last iteration (mirrors hb_bayes_out.last / g_last / vargL_last): what g_init + warm of a following run take */
    hbo_warm last;
    double *g_last;      /* m or NULL */
    double *vargL_last;  /* m or NULL (BayesL) */
} hbo_out;

/* Full sampler. Returns 0 on success; non-zero with out->error set to the
 * reference's exception text otherwise. */
int hbo_bayes(const hbo_args *a, hbo_out *o);

/* PLINK .bed decode, SNP-major, reference src/read_bed.cpp:116-167:
 * code 00 -> 2, 10 -> 1, 11 -> 0, 01 -> missing (returned as -128); then the
 * major-genotype imputation of :182-230 when impute != 0.  out is nind x nsnp
 * column-major int8. bed points at the whole file including the 3 magic bytes. */
int hbo_decode_bed(const uint8_t *bed, int64_t nbytes, int32_t nind, int32_t nsnp,
                   int impute, int8_t *out);

/* n-long dot / axpy kernels exactly as the sampler uses them (exposed for tests) */
double hbo_ddot(int n, const double *x, const double *y);

#ifdef __cplusplus
}
#endif
#endif
