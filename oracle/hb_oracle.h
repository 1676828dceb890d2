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
 * PARITY PIN STATUS.  The reference has no test-suite and R/Rcpp/Armadillo are not
 * installed, so src/Bayes.cpp cannot be built here (it includes RcppArmadillo.h, R.h,
 * Rmath.h and links BLAS ddot_/daxpy_ — none present).  What pins this
 * restatement: (1) the .bed decode against the genotype corner printed at reference
 * README.md:81-86; (2) the derived initial-state facts for inst/extdata/demo.*
 * (n=300, var(y), sumvx, nvar0, xpx[0:5], varg, s2varg_, vare_, lambda2) recorded in
 * SURVEY.md §4 from the formulas at src/Bayes.cpp:310-363; (3) R's published outputs for
 * set.seed()/runif()/rnorm() and the Random123 Philox vectors for the RNG layer; (4) the
 * README.md:159-167 posterior band as a soft sanity check.  Chain-level output of real
 * hibayes is NOT available => the sampler itself is "parity unpinned" at bit level.
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
