/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 * CPU restatement of hibayes' summary-level sampler SBayesD() (reference src/SBayesD.cpp:5-609), see hb_sbayes_oracle.c.
 */
#ifndef HB_SBAYES_ORACLE_H
#define HB_SBAYES_ORACLE_H
#include <stdint.h>
#include "hb_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t m;
    const double *sumstat;    /* m x 4 column-major: the columns sbrm() keeps (R/sbayes.r:207): MAF, BETA, SE, NMISS; NaN = NA */
    const double *ldm;        /* m x m column-major dense LD (variance-covariance) matrix, SBayesD.cpp:7 */
    const char *model;
    const double *Pi;
    int32_t n_pi;
    const double *fold;       /* n_fold or NULL */
    int32_t n_fold;
    int32_t niter, nburn, thin;
    double vg, dfvg, s2vg, ve, dfve, s2ve; /* NaN == R_NilValue */
    const uint32_t *windindx; /* m, 1-based, or NULL */
    int32_t rng_kind;         /* HBO_RNG_R | HBO_RNG_PHILOX */
    uint64_t seed;
} hbo_sb_args;

typedef struct {
    double Vg, Ve, h2;
    int32_t n_records, nzct, nw, n, count_y;
    double vary;
    double *alpha;   /* m */
    double *pi;      /* n_pi */
    double *pip;     /* m */
    double *gwas;    /* nw */
    double *s_Vg, *s_Ve, *s_h2; /* n_records */
    double *s_alpha; /* m x n_records or NULL */
    double *s_pi;    /* n_pi x n_records */
    double *r_hat;   /* m: the Gram-space right-hand side after the last sweep (for the invariant r_hat = xy - n ldm g) */
    double *g_last;  /* m: effects after the last sweep */
    double loop_seconds;
    int32_t iters_done;
    char error[256];
} hbo_sb_out;

int hbo_sbayes(const hbo_sb_args *a, hbo_sb_out *o);

#ifdef __cplusplus
}
#endif
#endif
