/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product.
 *
 * CPU restatement of hibayes' summary-level sampler SBayesD() on a dense LD matrix (reference src/SBayesD.cpp:5-609,
 * v3.1.0; entry _hibayes_SBayesD, src/RcppExports.cpp:53; caller sbrm(), R/sbayes.r:101-239), written from the algorithm,
 * in plain C: the same six per-marker conditionals as Bayes() with the right-hand side kept in Gram space —
 * r_hat += n (g_old - g_new) ldm[:, i] after every move (:262-266) — and the two end-of-sweep variance draws of :466-474.
 *
 * PARITY PIN STATUS: see tests/test_oracle_sbayes.py — the README.md:293-311 summary of sbrm(BayesCpi) on inst/extdata/demo.ma
 * is tried there with R's stream; what it pins (or why it cannot) is stated in that test and in DESIGN.md. The scalar
 * samplers and both RNG back-ends are the pinned ones of hbo_rng.c.
 */
#include "hb_sbayes_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static int fail(hbo_sb_out *o, const char *msg)
{
    snprintf(o->error, sizeof(o->error), "%s", msg);
    return 1;
}

static int is_null(double v) { return isnan(v); }

/* Armadillo's accumulate (two interleaved accumulators): the exact compare sum(Pi) != 1 (:38) depends on it */
static double arma_sum(const double *v, int n)
{
    double a1 = 0.0, a2 = 0.0;
    int j;
    for (j = 1; j < n; j += 2) {
        a1 += v[j - 1];
        a2 += v[j];
    }
    if ((j - 1) < n) a1 += v[j - 1];
    return a1 + a2;
}

static double ddot(int n, const double *x, const double *y)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += x[i] * y[i];
    return s;
}

static void daxpy(int n, double a, const double *x, double *y)
{
    for (int i = 0; i < n; i++) y[i] += a * x[i];
}

/* marker-level draws: sequential from the one global stream (R kind) or addressed by (iteration, marker) (Philox kind) —
 * the same addressing as hb_oracle.c and the device (hbo_rng.h) */
typedef struct {
    int kind;
    hbo_stream_t *glob;
    uint64_t seed, sub;
} mdraw_t;

static double md_unif(mdraw_t *d, int j)
{
    if (d->kind == HBO_RNG_R) return hbo_unif(d->glob);
    return hbo_philox_uniform(d->seed, d->sub, (uint64_t)j * HBO_BLK_PER_MARKER + 0);
}
static double md_norm(mdraw_t *d, int j)
{
    if (d->kind == HBO_RNG_R) return hbo_norm(d->glob);
    return hbo_philox_normal(d->seed, d->sub, (uint64_t)j * HBO_BLK_PER_MARKER + 1);
}
static double md_chisq(mdraw_t *d, int j, double df)
{
    if (d->kind == HBO_RNG_R) return hbo_chisq(d->glob, df);
    hbo_stream_t t;
    hbo_stream_init_philox(&t, d->seed, d->sub, (uint64_t)j * HBO_BLK_PER_MARKER + 4);
    return hbo_chisq(&t, df);
}
static double md_invgauss(mdraw_t *d, int j, double mu, double lambda)
{
    if (d->kind == HBO_RNG_R) return hbo_invgauss(d->glob, mu, lambda);
    hbo_stream_t t;
    hbo_stream_init_philox(&t, d->seed, d->sub, (uint64_t)j * HBO_BLK_PER_MARKER + 2);
    return hbo_invgauss(&t, mu, lambda);
}

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int hbo_sbayes(const hbo_sb_args *a, hbo_sb_out *o)
{
    o->error[0] = 0;
    const int m = a->m;
    const char *model = a->model;
    /* :28 */
    const int model_index = !strcmp(model, "BayesRR") ? 1 : !strcmp(model, "BayesA") ? 2
                          : (!strcmp(model, "BayesB") || !strcmp(model, "BayesBpi")) ? 3
                          : (!strcmp(model, "BayesC") || !strcmp(model, "BayesCpi")) ? 4 : !strcmp(model, "BayesL") ? 5 : 6;
    const double *ss = a->sumstat; /* m x 4 column-major: [freq, b, se, N] (R/sbayes.r:207) */
    const double *ldm = a->ldm;
    /* :33-34  n = mean of the finite N, truncated to int */
    int n;
    {
        double s = 0;
        int c = 0;
        for (int k = 0; k < m; k++)
            if (isfinite(ss[3 * (size_t)m + k])) { s += ss[3 * (size_t)m + k]; c++; }
        n = (int)(s / (c ? c : 1));
    }
    int fixpi = (!strcmp(model, "BayesB") || !strcmp(model, "BayesC"));
    if (a->n_pi < 2) return fail(o, "Pi should be a vector.");
    double Pi[HBO_MAX_FOLD];
    const int n_fold = a->n_pi;
    if (n_fold > HBO_MAX_FOLD) return fail(o, "too many classes");
    memcpy(Pi, a->Pi, sizeof(double) * n_fold);
    if (arma_sum(Pi, n_fold) != 1) return fail(o, "sum of Pi should be 1.");
    if (Pi[0] == 1) return fail(o, "all markers have no effect size.");
    for (int i = 0; i < n_fold; i++)
        if (Pi[i] < 0 || Pi[i] > 1) return fail(o, "elements of Pi should be at the range of [0, 1]");
    double fold_[HBO_MAX_FOLD] = {0};
    if (a->fold) {
        if (a->n_fold != n_fold) return fail(o, "length of Pi and fold not equals.");
        memcpy(fold_, a->fold, sizeof(double) * n_fold);
    } else {
        if (!strcmp(model, "BayesR")) return fail(o, "'fold' should be provided for BayesR model.");
        if (n_fold != 2) return fail(o, "length of Pi and fold not equals.");
    }
    const int niter = a->niter, nburn = a->nburn, thin = a->thin;
    const int n_records = (niter - nburn) / thin;
    int count = 0, nzct = 0, NnzSnp = 0, indistflag;
    double xx, gi, gi_, rhs, lhs, logdetV, acceptProb, uhat, v, vargi;
    double *snptracker = NULL, *nzrate = NULL;
    const int always_in = (model_index == 1 || model_index == 2 || model_index == 5);
    if (always_in) { /* :69-72 */
        NnzSnp = m;
        Pi[0] = 0;
        Pi[1] = 1;
        fixpi = 1;
    } else {
        if (strcmp(model, "BayesR") && n_fold != 2)
            return fail(o, "length of Pi should be 2, the first value is the proportion of non-effect markers.");
        nzrate = (double *)calloc(m, sizeof(double));
        snptracker = (double *)calloc(m, sizeof(double));
    }
    double *xy = (double *)calloc(m, sizeof(double)), *r_hat = (double *)calloc(m, sizeof(double));
    double *tmp = (double *)calloc(m, sizeof(double)), *yyi = (double *)calloc(m, sizeof(double));
    double *g = (double *)calloc(m, sizeof(double)), *xpx = (double *)calloc(m, sizeof(double)), *vx = (double *)calloc(m, sizeof(double));
    for (int i = 0; i < m; i++) { /* :95-98 */
        vx[i] = ldm[(size_t)i * m + i];
        xpx[i] = vx[i] * n;
    }
    int count_y = 0, nvar0 = 0;
    unsigned char *ifest = (unsigned char *)malloc(m);
    for (int k = 0; k < m; k++) { /* :102-112 */
        const double b = ss[1 * (size_t)m + k], se = ss[2 * (size_t)m + k], N = ss[3 * (size_t)m + k];
        ifest[k] = 1;
        if (isnan(b) || isnan(se) || isnan(N)) {
            ifest[k] = 0;
            nvar0++;
        } else {
            xy[k] = xpx[k] * b;
            r_hat[k] = xy[k];
            yyi[k] = xpx[k] * (b * b + (N - 2) * se * se);
            count_y++;
        }
    }
    if (count_y == 0) return fail(o, "Lack of SE.");
    const double yy = arma_sum(yyi, m) / count_y;
    const double vary = yy / (n - 1);
    const double h2 = 0.5;
    const double dfvara_ = is_null(a->dfvg) ? 4 : a->dfvg;
    if (dfvara_ <= 2) return fail(o, "dfvg should not be less than 2.");
    double vara_ = is_null(a->vg) ? ((dfvara_ - 2) / dfvara_) * vary * h2 : a->vg;
    double vare_ = is_null(a->ve) ? vary * (1 - h2) : a->ve;
    const double dfvare_ = is_null(a->dfve) ? -2 : a->dfve;
    const double s2vara_ = is_null(a->s2vg) ? vara_ * (dfvara_ - 2) / dfvara_ : a->s2vg;
    const double sumvx = arma_sum(vx, m);
    double varg = vara_ / ((1 - Pi[0]) * sumvx);
    const double s2varg_ = s2vara_ / ((1 - Pi[0]) * sumvx);
    const double s2vare_ = is_null(a->s2ve) ? 0 : a->s2ve;
    if (niter < nburn) return fail(o, "Number of total iteration ('niter') shold be larger than burn-in ('nburn').");
    const double R2 = (dfvara_ - 2) / dfvara_;
    double lambda2 = 2 * (1 - R2) / (R2)*sumvx, lambda = sqrt(lambda2);
    const double shape0 = 1.1, rate0 = (shape0 - 1) / lambda2;
    double *vargL = NULL;
    if (model_index == 5) {
        vargL = (double *)malloc(sizeof(double) * m);
        for (int i = 0; i < m; i++) vargL[i] = varg;
    }
    double stemp[HBO_MAX_FOLD], fold_snp_num[HBO_MAX_FOLD] = {0}, logpi[HBO_MAX_FOLD], s[HBO_MAX_FOLD] = {0};
    double vara_fold[HBO_MAX_FOLD], vare_vara_fold[HBO_MAX_FOLD] = {0};
    for (int j = 0; j < n_fold; j++) vara_fold[j] = (vara_ / ((1 - Pi[0]) * sumvx)) * fold_[j];
    int nw = 0;
    double *wppai = NULL;
    if (a->windindx) {
        for (int i = 0; i < m; i++) nw = (int)a->windindx[i] > nw ? (int)a->windindx[i] : nw;
        wppai = (double *)calloc(nw, sizeof(double));
    }
    unsigned char *wflag = nw ? (unsigned char *)calloc(nw, 1) : NULL;
    o->n = n;
    o->count_y = count_y;
    o->vary = vary;
    o->nw = nw;
    o->n_records = n_records;
    double pi_sum[HBO_MAX_FOLD] = {0};
    double vara_sum = 0, vare_sum = 0, hsq_sum = 0;
    double *g_sum = (double *)calloc(m, sizeof(double));

    hbo_stream_t glob;
    if (a->rng_kind == HBO_RNG_R) hbo_stream_init_r(&glob, (uint32_t)a->seed);
    mdraw_t md = {a->rng_kind, &glob, a->seed, 0};
    const double t_start = now_sec();
    int iter;
    for (iter = 0; iter < niter; iter++) {
        if (a->rng_kind == HBO_RNG_PHILOX) {
            hbo_stream_init_philox(&glob, a->seed, ((uint64_t)HBO_PURPOSE_HOST << 56) | (uint64_t)iter, 0);
            md.sub = ((uint64_t)HBO_PURPOSE_MARKER << 56) | (uint64_t)iter;
        }
        switch (model_index) {
        case 1: /* :252-270 */
            for (int i = 0; i < m; i++) {
                if (!ifest[i]) continue;
                xx = xpx[i];
                gi = g[i];
                rhs = r_hat[i];
                if (gi) rhs += xx * gi;
                v = xx + vare_ / varg;
                gi = rhs / v + sqrt(vare_ / v) * md_norm(&md, i);
                gi_ = (g[i] - gi) * n;
                daxpy(m, gi_, ldm + (size_t)i * m, r_hat);
                g[i] = gi;
            }
            varg = (ddot(m, g, g) + s2varg_ * dfvara_) / hbo_chisq(&glob, dfvara_ + count_y);
            break;
        case 2: /* :271-288 */
            for (int i = 0; i < m; i++) {
                if (!ifest[i]) continue;
                xx = xpx[i];
                gi = g[i];
                varg = (gi * gi + s2varg_ * dfvara_) / md_chisq(&md, i, dfvara_ + 1);
                rhs = r_hat[i];
                if (gi) rhs += xx * gi;
                v = xx + vare_ / varg;
                gi = rhs / v + sqrt(vare_ / v) * md_norm(&md, i);
                gi_ = (g[i] - gi) * n;
                daxpy(m, gi_, ldm + (size_t)i * m, r_hat);
                g[i] = gi;
            }
            break;
        case 3: /* :289-325 */
            for (int j = 0; j < n_fold; j++) logpi[j] = log(Pi[j]);
            s[0] = logpi[0];
            for (int i = 0; i < m; i++) {
                if (!ifest[i]) continue;
                xx = xpx[i];
                gi = g[i];
                varg = (gi * gi + s2varg_ * dfvara_) / md_chisq(&md, i, dfvara_ + 1);
                rhs = r_hat[i];
                if (gi) rhs += xx * gi;
                lhs = xx / vare_;
                logdetV = log(varg * lhs + 1);
                uhat = rhs / (xx + vare_ / varg);
                s[1] = -0.5 * (logdetV - (rhs * uhat / vare_)) + logpi[1];
                acceptProb = 1 / (exp(s[0] - s[0]) + exp(s[1] - s[0]));
                indistflag = md_unif(&md, i) < acceptProb ? 0 : 1;
                snptracker[i] = indistflag;
                if (indistflag == 0) gi = 0;
                else {
                    v = xx + vare_ / varg;
                    gi = rhs / v + sqrt(vare_ / v) * md_norm(&md, i);
                }
                if (gi != g[i]) {
                    gi_ = (g[i] - gi) * n;
                    daxpy(m, gi_, ldm + (size_t)i * m, r_hat);
                    g[i] = gi;
                }
            }
            fold_snp_num[1] = arma_sum(snptracker, m);
            fold_snp_num[0] = m - nvar0 - fold_snp_num[1];
            NnzSnp = (int)fold_snp_num[1];
            if (!fixpi) {
                double xn[HBO_MAX_FOLD], sx;
                for (int j = 0; j < n_fold; j++) xn[j] = hbo_gamma(&glob, fold_snp_num[j] + 1, 1.0);
                sx = arma_sum(xn, n_fold);
                for (int j = 0; j < n_fold; j++) Pi[j] = xn[j] / sx;
            }
            break;
        case 4: /* :326-366 */
            for (int j = 0; j < n_fold; j++) logpi[j] = log(Pi[j]);
            s[0] = logpi[0];
            vargi = 0;
            for (int i = 0; i < m; i++) {
                if (!ifest[i]) continue;
                xx = xpx[i];
                gi = g[i];
                rhs = r_hat[i];
                if (gi) rhs += xx * gi;
                lhs = xx / vare_;
                logdetV = log(varg * lhs + 1);
                uhat = rhs / (xx + vare_ / varg);
                s[1] = -0.5 * (logdetV - (rhs * uhat / vare_)) + logpi[1];
                acceptProb = 1 / (exp(s[0] - s[0]) + exp(s[1] - s[0]));
                indistflag = md_unif(&md, i) < acceptProb ? 0 : 1;
                snptracker[i] = indistflag;
                if (indistflag == 0) gi = 0;
                else {
                    v = xx + vare_ / varg;
                    gi = rhs / v + sqrt(vare_ / v) * md_norm(&md, i);
                    vargi += gi * gi;
                }
                if (gi != g[i]) {
                    gi_ = (g[i] - gi) * n;
                    daxpy(m, gi_, ldm + (size_t)i * m, r_hat);
                    g[i] = gi;
                }
            }
            fold_snp_num[1] = arma_sum(snptracker, m);
            fold_snp_num[0] = m - nvar0 - fold_snp_num[1];
            NnzSnp = (int)fold_snp_num[1];
            varg = (vargi + s2varg_ * dfvara_) / hbo_chisq(&glob, dfvara_ + NnzSnp);
            if (!fixpi) {
                double xn[HBO_MAX_FOLD], sx;
                for (int j = 0; j < n_fold; j++) xn[j] = hbo_gamma(&glob, fold_snp_num[j] + 1, 1.0);
                sx = arma_sum(xn, n_fold);
                for (int j = 0; j < n_fold; j++) Pi[j] = xn[j] / sx;
            }
            break;
        case 5: /* :367-391 */
            for (int i = 0; i < m; i++) {
                if (!ifest[i]) continue;
                xx = xpx[i];
                gi = g[i];
                rhs = r_hat[i];
                if (gi) rhs += xx * gi;
                v = xx + 1 / vargL[i];
                gi = rhs / v + sqrt(vare_ / v) * md_norm(&md, i);
                if (fabs(gi) < 1e-6) gi = 1e-6;
                vargi = 1 / md_invgauss(&md, i, sqrt(vare_) * lambda / fabs(gi), lambda2);
                if (vargi > 0) vargL[i] = vargi;
                if (gi != g[i]) {
                    gi_ = (g[i] - gi) * n;
                    daxpy(m, gi_, ldm + (size_t)i * m, r_hat);
                    g[i] = gi;
                }
            }
            {
                const double shape = shape0 + count_y, rate = rate0 + arma_sum(vargL, m) / 2;
                lambda2 = hbo_gamma(&glob, shape, 1 / rate);
                lambda = sqrt(lambda2);
            }
            break;
        case 6: /* :392-461 */
            for (int j = 0; j < n_fold; j++) logpi[j] = log(Pi[j]);
            s[0] = logpi[0];
            varg = 0;
            for (int j = 1; j < n_fold; j++) vare_vara_fold[j] = vare_ / vara_fold[j];
            for (int i = 0; i < m; i++) {
                if (!ifest[i]) continue;
                xx = xpx[i];
                gi = g[i];
                rhs = r_hat[i];
                if (gi) rhs += xx * gi;
                lhs = xx / vare_;
                for (int j = 1; j < n_fold; j++) {
                    logdetV = log(vara_fold[j] * lhs + 1);
                    uhat = rhs / (xx + vare_vara_fold[j]);
                    s[j] = -0.5 * (logdetV - (rhs * uhat / vare_)) + logpi[j];
                }
                for (int j = 0; j < n_fold; j++) {
                    double temp = 0.0;
                    for (int k = 0; k < n_fold; k++) temp += exp(s[k] - s[j]);
                    stemp[j] = 1 / temp;
                }
                acceptProb = 0;
                indistflag = 0;
                const double rval = md_unif(&md, i);
                for (int j = 0; j < n_fold; j++) {
                    acceptProb += stemp[j];
                    if (rval < acceptProb) {
                        indistflag = j;
                        break;
                    }
                }
                snptracker[i] = indistflag;
                if (indistflag == 0) gi = 0;
                else {
                    v = xx + vare_vara_fold[indistflag];
                    gi = rhs / v + sqrt(vare_ / v) * md_norm(&md, i);
                    varg += (gi * gi / fold_[indistflag]);
                }
                if (gi != g[i]) {
                    gi_ = (g[i] - gi) * n;
                    daxpy(m, gi_, ldm + (size_t)i * m, r_hat);
                    g[i] = gi;
                }
            }
            for (int j = 0; j < n_fold; j++) {
                double c = 0;
                for (int i = 0; i < m; i++) c += snptracker[i] == j;
                fold_snp_num[j] = c;
            }
            NnzSnp = (int)(m - fold_snp_num[0]);
            varg = (varg + s2varg_ * dfvara_) / hbo_chisq(&glob, dfvara_ + NnzSnp);
            for (int j = 0; j < n_fold; j++) vara_fold[j] = varg * fold_[j];
            fold_snp_num[0] -= nvar0;
            if (!fixpi) {
                double xn[HBO_MAX_FOLD], sx;
                for (int j = 0; j < n_fold; j++) xn[j] = hbo_gamma(&glob, fold_snp_num[j] + 1, 1.0);
                sx = arma_sum(xn, n_fold);
                for (int j = 0; j < n_fold; j++) Pi[j] = xn[j] / sx;
            }
            break;
        }
        /* :466-474 */
        for (int i = 0; i < m; i++) tmp[i] = xy[i] - r_hat[i];
        vara_ = (ddot(m, g, tmp) + s2vara_ * dfvara_) / hbo_chisq(&glob, n + dfvara_);
        for (int i = 0; i < m; i++) tmp[i] = xy[i] + r_hat[i];
        vare_ = (yy - ddot(m, g, tmp) + s2vare_ * dfvare_) / hbo_chisq(&glob, n + dfvare_);
        if (vare_ < 0) vare_ = vara_ * 0.5;

        if (iter >= nburn) { /* :476-497 */
            if (snptracker) {
                for (int i = 0; i < m; i++)
                    if (snptracker[i]) nzrate[i] += 1;
            }
            if (nw) {
                memset(wflag, 0, nw);
                for (int i = 0; i < m; i++)
                    if (snptracker[i]) wflag[a->windindx[i] - 1] = 1;
                for (int w = 0; w < nw; w++) wppai[w] += wflag[w];
            }
            nzct++;
        }
        if (iter >= nburn && (iter + 1 - nburn) % thin == 0) { /* :499-512 */
            if (!fixpi)
                for (int j = 0; j < n_fold; j++) {
                    if (o->s_pi) o->s_pi[(size_t)count * n_fold + j] = Pi[j];
                    pi_sum[j] += Pi[j];
                }
            if (o->s_Vg) o->s_Vg[count] = vara_;
            if (o->s_Ve) o->s_Ve[count] = vare_;
            if (o->s_h2) o->s_h2[count] = vara_ / (vara_ + vare_);
            vara_sum += vara_;
            vare_sum += vare_;
            hsq_sum += vara_ / (vara_ + vare_);
            if (o->s_alpha) memcpy(o->s_alpha + (size_t)count * m, g, sizeof(double) * m);
            for (int i = 0; i < m; i++) g_sum[i] += g[i];
            count++;
        }
        if (count == n_records) {
            iter++;
            break;
        }
    }
    o->loop_seconds = now_sec() - t_start;
    o->iters_done = iter;
    /* :541-580 */
    const double Rn = (double)n_records;
    o->Vg = vara_sum / Rn;
    o->Ve = vare_sum / Rn;
    o->h2 = hsq_sum / Rn;
    if (o->alpha)
        for (int i = 0; i < m; i++) o->alpha[i] = g_sum[i] / Rn;
    if (!fixpi) {
        for (int j = 0; j < n_fold; j++) Pi[j] = pi_sum[j] / Rn;
    } else if (o->s_pi) {
        for (int r = 0; r < n_records; r++) {
            o->s_pi[(size_t)r * n_fold + 0] = Pi[0];
            o->s_pi[(size_t)r * n_fold + 1] = Pi[1];
        }
    }
    if (o->pi) memcpy(o->pi, Pi, sizeof(double) * n_fold);
    if (o->pip) {
        for (int i = 0; i < m; i++) {
            if (!nzrate) o->pip[i] = 1.0;
            else {
                double p = nzrate[i] / nzct;
                if (p == 1) p = (nzct - 1) / (double)nzct;
                o->pip[i] = p;
            }
        }
    }
    if (nw && o->gwas)
        for (int w = 0; w < nw; w++) {
            double p = wppai[w] / nzct;
            if (p == 1) p = (nzct - 1) / (double)nzct;
            o->gwas[w] = p;
        }
    o->nzct = nzct;
    if (o->r_hat) memcpy(o->r_hat, r_hat, sizeof(double) * m);
    if (o->g_last) memcpy(o->g_last, g, sizeof(double) * m);
    free(xy); free(r_hat); free(tmp); free(yyi); free(g); free(xpx); free(vx); free(ifest); free(g_sum);
    free(snptracker); free(nzrate); free(vargL); free(wppai); free(wflag);
    return 0;
}
