/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (scope and pin status: hb_oracle.h).
 *
 * Restatement of reference src/Bayes.cpp (hibayes 3.1.0) for the path the GPU engine
 * replaces: validation (:92-117, :293, :324-326, :356-358), sizes (:119-124), covariate
 * and random-effect set-up (:126-201, makeZ :29-57), marker statistics (:310-317), prior
 * defaults (:319-374), the MCMC loop (:477-917) with the intercept / covariate /
 * random-effect blocks (:479-516) and the six marker sweeps (:586-816), the variance
 * draws (:819-823), PIP / WPPA counters (:826-845), the thinned store (:848-882) and the
 * posterior assembly (:919-1040).  BSLMM (nk) and the single-step epsilon block (ne) are
 * out of scope and are not restated.
 *
 * The serial marker loop and the BLAS-1 shaped dot/axpy are kept on purpose: this file is
 * also the "port" CPU baseline that bench.py times beside the GPU.
 */
#define _GNU_SOURCE
#include "hb_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------- */
static int g_threads = 1;

/* ddot_/daxpy_ stand-ins (reference src/hibayes.h:21-29 declares the BLAS symbols).
 *
 * Threaded form = what a threaded BLAS amounts to for n-long level-1 calls: ONE persistent team of workers for the whole run
 * (created in hbo_bayes, never forked/joined per call), every worker owning a fixed chunk of the rows, woken by a generation
 * word it spins on, handing its partial sum back through its own cache line; the partial sums are combined in a two-level tree
 * (groups of 8) in a fixed order, so a threaded run is deterministic for a given thread count. Rounds 1-4 used an
 * `omp parallel for` per call, whose fork/join (and the passive wait policy the bench set) cost more than the 400 KB a call
 * streams: 64 threads ran ten times SLOWER than one. */
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#if defined(__x86_64__)
#include <immintrin.h>
#define HBO_PAUSE() _mm_pause()
#else
#define HBO_PAUSE() ((void)0)
#endif

#define HBO_TEAM_GROUP 8
/* one cache line per worker: what it hands back */
typedef struct {
    _Alignas(128) atomic_ulong gen;  /* generation this worker has FINISHED (written by the worker only) */
    double part;                     /* its partial dot */
    double gsum;                     /* leaders: the group's sum ... */
    atomic_ulong ggen;               /* ... of this generation */
    char pad[128 - 2 * sizeof(atomic_ulong) - 2 * sizeof(double)];
} team_slot_t;
/* one cache line per GROUP: the posted job, copied down the tree (the master posts to the root line, the eight-or-so group leaders
 * spin on that, each leader re-posts to its own line, its seven members spin on that: nobody's store has to invalidate 63 readers) */
typedef struct {
    _Alignas(128) atomic_ulong go;   /* generation posted */
    int op;                          /* 0 dot, 1 axpy, 2 quit; 3 / 4: dot / axpy on an int8 column (x points at int8_t) */
    double a;
    const void *x;
    const double *yc;
    double *y;
    char pad[128 - sizeof(atomic_ulong) - sizeof(int) - sizeof(double) - 3 * sizeof(void *) - 4];
} team_job_t;

typedef struct {
    int nthreads, n, ngroups;
    pthread_t *tid;
    team_slot_t *slot;
    team_job_t *job;                 /* [0] the root line, [1 + g] group g's line */
    int pinned;
    cpu_set_t master_mask;
} team_t;

typedef struct { team_t *t; int id; int cpu; } team_arg_t;
static team_t *g_team = NULL;

static inline void team_wait(atomic_ulong *w, unsigned long want)
{
    for (unsigned spins = 0; atomic_load_explicit(w, memory_order_acquire) != want; spins++) {
        if ((spins & 4095) == 4095) sched_yield();   /* oversubscribed box: let the thread we wait for run */
        else HBO_PAUSE();
    }
}

static inline void team_range(const team_t *t, int id, int *lo, int *hi)
{   /* chunks in multiples of 8 doubles (one cache line), the same for every call: a worker's rows stay in ITS cache */
    const long lines = ((long)t->n + 7) / 8;
    const long l0 = lines * id / t->nthreads, l1 = lines * (id + 1) / t->nthreads;
    *lo = (int)(l0 * 8);
    *hi = (int)(l1 * 8 > t->n ? t->n : l1 * 8);
}

/* worker `id` runs its share of job `jb` (generation gen); a leader first re-posts the job to its group */
static int team_do(team_t *t, int id, unsigned long gen, const team_job_t *jb)
{
    /* the job line is read HERE, before this worker publishes anything: once it has signalled completion the master (or its leader)
     * may already be rewriting the line for the next generation (advisor finding, round 5: the worker used to re-read op afterwards) */
    const int op = jb->op;
    const double a = jb->a;
    const void *xv = jb->x;
    const double *yc = jb->yc;
    double *y = jb->y;
    const int leader = id % HBO_TEAM_GROUP == 0;
    if (leader && id + 1 < t->nthreads) {
        team_job_t *gl = &t->job[1 + id / HBO_TEAM_GROUP];
        gl->op = op; gl->a = a; gl->x = xv; gl->yc = yc; gl->y = y;
        atomic_store_explicit(&gl->go, gen, memory_order_release);
    }
    if (op == 2) return op;
    int lo, hi;
    team_range(t, id, &lo, &hi);
    double s = 0.0;
    if (op == 0) {
        const double *x = (const double *)xv;
#pragma omp simd reduction(+ : s)
        for (int i = lo; i < hi; i++) s += x[i] * yc[i];
    } else if (op == 1) {
        const double *x = (const double *)xv;
#pragma omp simd
        for (int i = lo; i < hi; i++) y[i] += a * x[i];
    } else if (op == 3) {
        const int8_t *x = (const int8_t *)xv;
#pragma omp simd reduction(+ : s)
        for (int i = lo; i < hi; i++) s += (double)x[i] * yc[i];
    } else {
        const int8_t *x = (const int8_t *)xv;
#pragma omp simd
        for (int i = lo; i < hi; i++) y[i] += a * (double)x[i];
    }
    team_slot_t *me = &t->slot[id];
    me->part = s;
    if (!leader) { atomic_store_explicit(&me->gen, gen, memory_order_release); return op; }
    double g = s;       /* group leader: its members' partial sums, in member order */
    for (int k = id + 1; k < id + HBO_TEAM_GROUP && k < t->nthreads; k++) {
        team_wait(&t->slot[k].gen, gen);
        g += t->slot[k].part;
    }
    me->gsum = g;
    atomic_store_explicit(&me->ggen, gen, memory_order_release);
    return op;
}

static void *team_worker(void *p)
{
    team_arg_t *ta = (team_arg_t *)p;
    team_t *t = ta->t;
    const int id = ta->id;
    if (ta->cpu >= 0) {
        cpu_set_t cs;
        CPU_ZERO(&cs);
        CPU_SET(ta->cpu, &cs);
        pthread_setaffinity_np(pthread_self(), sizeof(cs), &cs);
    }
    /* a leader listens to the root line, a member to its leader's line */
    team_job_t *src = (id % HBO_TEAM_GROUP == 0) ? &t->job[0] : &t->job[1 + id / HBO_TEAM_GROUP];
    for (unsigned long gen = 1;; gen++) {
        team_wait(&src->go, gen);
        if (team_do(t, id, gen, src) == 2) break;
    }
    return NULL;
}

static team_arg_t *g_team_args = NULL;
static void team_start(int nthreads, int n)
{
    team_t *t = (team_t *)aligned_alloc(128, (sizeof(team_t) + 127) / 128 * 128);
    memset(t, 0, sizeof(*t));
    t->nthreads = nthreads;
    t->n = n;
    t->ngroups = (nthreads + HBO_TEAM_GROUP - 1) / HBO_TEAM_GROUP;
    t->slot = (team_slot_t *)aligned_alloc(128, sizeof(team_slot_t) * nthreads);
    memset(t->slot, 0, sizeof(team_slot_t) * nthreads);
    t->job = (team_job_t *)aligned_alloc(128, sizeof(team_job_t) * (1 + t->ngroups));
    memset(t->job, 0, sizeof(team_job_t) * (1 + t->ngroups));
    t->tid = (pthread_t *)calloc(nthreads, sizeof(pthread_t));
    g_team_args = (team_arg_t *)calloc(nthreads, sizeof(team_arg_t));
    /* one worker per CPU of the process's affinity mask, in mask order (on the usual numbering the first cores of one socket, no
     * SMT siblings) — a spinning team that the scheduler may stack on sibling threads or migrate is what made 64 threads slower
     * than one. Not pinned when there are fewer CPUs than workers (the yield in team_wait then keeps it live). */
    int cpus[1024], ncpu = 0;
    cpu_set_t cs;
    if (sched_getaffinity(0, sizeof(cs), &cs) == 0)
        for (int c = 0; c < CPU_SETSIZE && ncpu < 1024; c++)
            if (CPU_ISSET(c, &cs)) cpus[ncpu++] = c;
    t->pinned = ncpu >= nthreads && !getenv("HBO_TEAM_NOPIN");
    if (t->pinned) {
        pthread_getaffinity_np(pthread_self(), sizeof(t->master_mask), &t->master_mask);
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(cpus[0], &one);
        pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
    }
    for (int i = 1; i < nthreads; i++) {
        g_team_args[i].t = t;
        g_team_args[i].id = i;
        g_team_args[i].cpu = t->pinned ? cpus[i] : -1;
        pthread_create(&t->tid[i], NULL, team_worker, &g_team_args[i]);
    }
    g_team = t;
}

static double team_run(team_t *t, int op, double a, const void *x, const double *yc, double *y)
{
    team_job_t *root = &t->job[0];
    root->op = op; root->a = a; root->x = x; root->yc = yc; root->y = y;
    const unsigned long gen = atomic_load_explicit(&root->go, memory_order_relaxed) + 1;
    atomic_store_explicit(&root->go, gen, memory_order_release);
    team_do(t, 0, gen, root);                 /* the master is worker 0 and the leader of group 0 */
    if (op == 2) return 0.0;
    double s = t->slot[0].gsum;
    for (int g = HBO_TEAM_GROUP; g < t->nthreads; g += HBO_TEAM_GROUP) {
        team_wait(&t->slot[g].ggen, gen);     /* (also the completion barrier of an axpy) */
        s += t->slot[g].gsum;
    }
    return s;
}

static void team_stop(void)
{
    team_t *t = g_team;
    if (!t) return;
    team_run(t, 2, 0.0, NULL, NULL, NULL);
    for (int i = 1; i < t->nthreads; i++) pthread_join(t->tid[i], NULL);
    if (t->pinned) pthread_setaffinity_np(pthread_self(), sizeof(t->master_mask), &t->master_mask);
    free(t->tid); free(t->slot); free(t->job); free(t); free(g_team_args);
    g_team = NULL; g_team_args = NULL;
}

double hbo_ddot(int n, const double *x, const double *y)
{
    if (g_team && n == g_team->n) return team_run(g_team, 0, 0.0, x, y, NULL);
    double s = 0.0;
#pragma omp simd reduction(+ : s)
    for (int i = 0; i < n; i++) s += x[i] * y[i];
    return s;
}

static void hbo_daxpy(int n, double a, const double *x, double *y)
{
    if (g_team && n == g_team->n) { team_run(g_team, 1, a, x, NULL, y); return; }
#pragma omp simd
    for (int i = 0; i < n; i++) y[i] += a * x[i];
}

static double ddot_i8(int n, const int8_t *x, const double *y)
{
    if (g_team && n == g_team->n) return team_run(g_team, 3, 0.0, x, y, NULL);
    double s = 0.0;
#pragma omp simd reduction(+ : s)
    for (int i = 0; i < n; i++) s += (double)x[i] * y[i];
    return s;
}

static void daxpy_i8(int n, double a, const int8_t *x, double *y)
{
    if (g_team && n == g_team->n) { team_run(g_team, 4, a, x, NULL, y); return; }
#pragma omp simd
    for (int i = 0; i < n; i++) y[i] += a * (double)x[i];
}

/* column accessors: the reference holds X as doubles; the int8 form is a test convenience */
typedef struct {
    const double *X;
    const int8_t *X8;
    int n;
} xmat_t;

static inline double col_dot(const xmat_t *x, int j, const double *v)
{
    if (x->X) return hbo_ddot(x->n, x->X + (size_t)j * x->n, v);
    return ddot_i8(x->n, x->X8 + (size_t)j * x->n, v);
}

static inline void col_axpy(const xmat_t *x, int j, double a, double *v)
{
    if (x->X) hbo_daxpy(x->n, a, x->X + (size_t)j * x->n, v);
    else daxpy_i8(x->n, a, x->X8 + (size_t)j * x->n, v);
}

static inline double col_get(const xmat_t *x, int j, int i)
{
    return x->X ? x->X[(size_t)j * x->n + i] : (double)x->X8[(size_t)j * x->n + i];
}

/* Armadillo's arrayops::accumulate: two interleaved accumulators (matters only for the
 * exact compare sum(Pi) != 1 at src/Bayes.cpp:101). */
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

/* arma::var, norm_type 0: two-pass with the N-1 denominator */
static double var_n1(const double *v, int n)
{
    if (n < 2) return 0.0;
    double mean = arma_sum(v, n) / n;
    double acc2 = 0.0, acc3 = 0.0;
    for (int i = 0; i < n; i++) {
        double t = mean - v[i];
        acc2 += t * t;
        acc3 += t;
    }
    return (acc2 - acc3 * acc3 / n) / (n - 1);
}

static int fail(hbo_out *o, const char *msg)
{
    snprintf(o->error, sizeof(o->error), "%s", msg);
    return 1;
}

static int is_null(double v) { return isnan(v); }

static int cmp_str(const void *a, const void *b)
{
    return strcmp(*(const char *const *)a, *(const char *const *)b);
}

/* ---------------------------------------------------------------------------------- */
/* marker-level draws: sequential from the global stream (R kind) or addressed (Philox) */
typedef struct {
    int kind;
    hbo_stream_t *glob; /* R kind: the one global stream */
    uint64_t seed, sub; /* Philox marker stream of this iteration */
    int64_t off;
} mdraw_t;

static inline double md_unif(mdraw_t *d, int j)
{
    if (d->kind == HBO_RNG_R) return hbo_unif(d->glob);
    return hbo_philox_uniform(d->seed, d->sub, (uint64_t)(d->off + j) * HBO_BLK_PER_MARKER + 0);
}

static inline double md_norm(mdraw_t *d, int j)
{
    if (d->kind == HBO_RNG_R) return hbo_norm(d->glob);
    return hbo_philox_normal(d->seed, d->sub, (uint64_t)(d->off + j) * HBO_BLK_PER_MARKER + 1);
}

static inline double md_chisq(mdraw_t *d, int j, double df)
{
    if (d->kind == HBO_RNG_R) return hbo_chisq(d->glob, df);
    hbo_stream_t t;
    hbo_stream_init_philox(&t, d->seed, d->sub, (uint64_t)(d->off + j) * HBO_BLK_PER_MARKER + 4);
    return hbo_chisq(&t, df);
}

static inline double md_invgauss(mdraw_t *d, int j, double mu, double lambda)
{
    if (d->kind == HBO_RNG_R) return hbo_invgauss(d->glob, mu, lambda);
    hbo_stream_t t;
    hbo_stream_init_philox(&t, d->seed, d->sub, (uint64_t)(d->off + j) * HBO_BLK_PER_MARKER + 2);
    return hbo_invgauss(&t, mu, lambda);
}

/* src/stats.cpp:8-11 */
static inline double norm_sample(double z, double mean, double sd) { return mean + sd * z; }

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ---------------------------------------------------------------------------------- */
int hbo_bayes(const hbo_args *a, hbo_out *o)
{
    o->error[0] = 0;
    const int n = a->n, m = a->m;
    const double *y = a->y;
    xmat_t X = {a->X, a->X ? NULL : a->X8, n};
    const char *model = a->model;
    g_threads = a->threads > 0 ? a->threads : 1;
#ifdef _OPENMP
    if (a->threads <= 0) g_threads = omp_get_max_threads(); /* src/omp_set.h:10-22 */
#endif

    /* ---- validation, src/Bayes.cpp:92-117 ---- */
    for (int i = 0; i < n; i++)
        if (isnan(y[i])) return fail(o, "NAs are not allowed in y.");
    int model_index = !strcmp(model, "BayesRR") ? 1
                    : !strcmp(model, "BayesA")  ? 2
                    : (!strcmp(model, "BayesB") || !strcmp(model, "BayesBpi")) ? 3
                    : (!strcmp(model, "BayesC") || !strcmp(model, "BayesCpi") || !strcmp(model, "BSLMM")) ? 4
                    : !strcmp(model, "BayesL")  ? 5 : 6;
    int fixpi = (!strcmp(model, "BayesB") || !strcmp(model, "BayesC"));
    if (a->n_pi < 2) return fail(o, "Pi should be a vector.");
    double Pi[HBO_MAX_FOLD];
    if (a->n_pi > HBO_MAX_FOLD) return fail(o, "too many mixture classes for the oracle.");
    for (int i = 0; i < a->n_pi; i++) Pi[i] = a->Pi[i];
    const int n_pi = a->n_pi;
    if (arma_sum(Pi, n_pi) != 1) return fail(o, "sum of Pi should be 1.");
    if (Pi[0] == 1) return fail(o, "all markers have no effect size.");
    for (int i = 0; i < n_pi; i++)
        if (Pi[i] < 0 || Pi[i] > 1) return fail(o, "elements of Pi should be at the range of [0, 1]");
    double fold_[HBO_MAX_FOLD] = {0};
    int n_fold;
    if (a->fold) {
        if (a->n_fold > HBO_MAX_FOLD) return fail(o, "too many mixture classes for the oracle.");
        n_fold = a->n_fold;
        for (int i = 0; i < n_fold; i++) fold_[i] = a->fold[i];
    } else {
        if (!strcmp(model, "BayesR")) return fail(o, "'fold' should be provided for BayesR model.");
        n_fold = 2;
    }
    if (n_fold != n_pi) return fail(o, "length of Pi and fold not equals.");

    /* ---- sizes, :119-124 ---- */
    const double vary = var_n1(y, n);
    const double h2 = 0.5;
    const int niter = a->niter, nburn = a->nburn, thin = a->thin;
    const int n_records = (niter - nburn) / thin;
    o->n_records = n_records;

    /* ---- covariates, :126-147 ---- */
    const int nc = a->C ? a->nc : 0;
    double *beta = NULL, *cpc = NULL;
    if (nc) {
        for (size_t i = 0; i < (size_t)n * nc; i++)
            if (isnan(a->C[i]))
                return fail(o, "Individuals with phenotypic value should not have missing covariates.");
        beta = (double *)calloc(nc, sizeof(double));
        cpc = (double *)calloc(nc, sizeof(double));
        for (int i = 0; i < nc; i++) cpc[i] = hbo_ddot(n, a->C + (size_t)i * n, a->C + (size_t)i * n);
    }

    /* ---- environmental random effects, :149-201 with makeZ :29-57 ---- */
    const int nr = a->R ? a->nr : 0;
    double dfr = is_null(a->dfvr) ? -1 : a->dfvr;
    double s2r = is_null(a->s2vr) ? 0 : a->s2vr;
    double *vr = NULL, *vrtmp = NULL, *estR = NULL;
    int *R_first = NULL, *R_q = NULL; /* first level / number of levels per term */
    int *zid = NULL;                  /* n x nr: level index of each record   */
    double *zz = NULL;                /* per level: diag(Z'Z) = count         */
    int n_levels = 0;
    if (nr) {
        vr = (double *)calloc(nr, sizeof(double));
        vrtmp = (double *)calloc(nr, sizeof(double));
        R_first = (int *)calloc(nr, sizeof(int));
        R_q = (int *)calloc(nr, sizeof(int));
        zid = (int *)calloc((size_t)n * nr, sizeof(int));
        for (int i = 0; i < nr; i++) vrtmp[i] = vary * (1 - h2) / (nr + 1);
        const char **tmp = (const char **)malloc(sizeof(char *) * n);
        int *firsts = (int *)calloc(nr + 1, sizeof(int));
        /* first pass: count levels */
        for (int i = 0; i < nr; i++) {
            for (int k = 0; k < n; k++) tmp[k] = a->R[(size_t)i * n + k];
            qsort(tmp, n, sizeof(char *), cmp_str);
            int q = 0;
            for (int k = 0; k < n; k++)
                if (k == 0 || strcmp(tmp[k], tmp[k - 1]) != 0) tmp[q++] = tmp[k];
            if (q == n) { free(tmp); free(firsts);
                return fail(o, "number of class of environmental random effects should be less than population size."); }
            if (q == 1) { free(tmp); free(firsts);
                return fail(o, "number of class of environmental random effects should be bigger than 1."); }
            R_first[i] = n_levels;
            R_q[i] = q;
            for (int k = 0; k < n; k++) {
                const char *key = a->R[(size_t)i * n + k];
                int lo = 0, hi = q - 1;
                while (lo < hi) {
                    int mid = (lo + hi) / 2;
                    if (strcmp(tmp[mid], key) < 0) lo = mid + 1; else hi = mid;
                }
                zid[(size_t)i * n + k] = lo;
            }
            n_levels += q;
        }
        free(tmp);
        free(firsts);
        estR = (double *)calloc(n_levels, sizeof(double));
        zz = (double *)calloc(n_levels, sizeof(double));
        for (int i = 0; i < nr; i++)
            for (int k = 0; k < n; k++) zz[R_first[i] + zid[(size_t)i * n + k]] += 1.0;
    }
    o->n_levels = n_levels;

    /* ---- state, :277-308 ---- */
    int count = 0, nzct = 0, NnzSnp = 0;
    double *snptracker = NULL, *nzrate = NULL;
    if (model_index == 1 || model_index == 2 || model_index == 5) {
        NnzSnp = m;
        Pi[0] = 0; Pi[1] = 1;
        fixpi = 1;
    } else {
        if (strcmp(model, "BayesR") && n_pi != 2)
            return fail(o, "length of Pi should be 2, the first value is the proportion of non-effect markers.");
        nzrate = (double *)calloc(m, sizeof(double));
        snptracker = (double *)calloc(m, sizeof(double));
    }
    double *g = (double *)calloc(m, sizeof(double));
    double *alpha_sum = (double *)calloc(m, sizeof(double));
    double *u = (double *)calloc(n, sizeof(double));
    double *xpx = (double *)calloc(m, sizeof(double));
    double *vx = (double *)calloc(m, sizeof(double));
    double *beta_sum = nc ? (double *)calloc(nc, sizeof(double)) : NULL;
    double *estR_sum = n_levels ? (double *)calloc(n_levels, sizeof(double)) : NULL;
    double *vr_sum = nr ? (double *)calloc(nr, sizeof(double)) : NULL;
    double pi_sum[HBO_MAX_FOLD] = {0};
    double mu_sum = 0, vara_sum = 0, vare_sum = 0, hsq_sum = 0;

    /* ---- marker statistics, :310-317 ---- */
    {
        double *col = (double *)malloc(sizeof(double) * n);
        for (int i = 0; i < m; i++) {
            double s2 = 0;
            for (int k = 0; k < n; k++) { col[k] = col_get(&X, i, k); s2 += col[k] * col[k]; }
            xpx[i] = s2;
            vx[i] = var_n1(col, n);
        }
        free(col);
    }
    double sumvx = arma_sum(vx, m);
    int nvar0 = 0;
    for (int i = 0; i < m; i++) nvar0 += (vx[i] == 0);

    /* ---- prior defaults, :319-374 ---- */
    double dfvara_ = is_null(a->dfvg) ? 4 : a->dfvg;
    if (dfvara_ <= 2) return fail(o, "dfvg should not be less than 2.");
    double vara_ = is_null(a->vg) ? ((dfvara_ - 2) / dfvara_) * vary * h2 : a->vg;
    double vare_ = is_null(a->ve) ? vary * (1 - h2) / (nr + 1) : a->ve;
    double dfvare_ = is_null(a->dfve) ? -2 : a->dfve;
    double s2vara_ = is_null(a->s2vg) ? vara_ * (dfvara_ - 2) / dfvara_ : a->s2vg;
    double varg = vara_ / ((1 - Pi[0]) * sumvx);
    double s2varg_ = s2vara_ / ((1 - Pi[0]) * sumvx);
    double s2vare_ = is_null(a->s2ve) ? 0 : a->s2ve;
    if (niter < nburn)
        return fail(o, "Number of total iteration ('niter') shold be larger than burn-in ('nburn').");
    double R2 = (dfvara_ - 2) / dfvara_;
    double lambda2 = 2 * (1 - R2) / (R2) * sumvx;
    double lambda = sqrt(lambda2);
    double shape0 = 1.1;
    double rate0 = (shape0 - 1) / lambda2;
    double *vargL = NULL;
    if (model_index == 5) {
        vargL = (double *)malloc(sizeof(double) * m);
        for (int i = 0; i < m; i++) vargL[i] = varg;
    }
    double stemp[HBO_MAX_FOLD] = {0}, fold_snp_num[HBO_MAX_FOLD] = {0}, logpi[HBO_MAX_FOLD] = {0},
           s[HBO_MAX_FOLD] = {0}, vara_fold[HBO_MAX_FOLD] = {0}, vare_vara_fold[HBO_MAX_FOLD] = {0};
    for (int j = 0; j < n_fold; j++) vara_fold[j] = (vara_ / ((1 - Pi[0]) * sumvx)) * fold_[j];

    o->vary = vary; o->sumvx = sumvx; o->nvar0 = nvar0; o->varg0 = varg; o->s2varg = s2varg_;
    o->vara0 = vara_; o->s2vara = s2vara_; o->vare0 = vare_; o->lambda2_0 = lambda2; o->rate0 = rate0;
    if (o->xpx) memcpy(o->xpx, xpx, sizeof(double) * m);
    if (o->vx) memcpy(o->vx, vx, sizeof(double) * m);

    /* ---- GWAS windows, :376-391 ---- */
    int nw = 0;
    double *wppai = NULL;
    unsigned char *wflag = NULL;
    if (a->windindx) {
        for (int i = 0; i < m; i++) if ((int)a->windindx[i] > nw) nw = (int)a->windindx[i];
        wppai = (double *)calloc(nw, sizeof(double));
        wflag = (unsigned char *)calloc(nw, 1);
    }
    o->nw = nw;

    /* ---- RNG ---- */
    hbo_stream_t glob;
    if (a->rng_kind == HBO_RNG_R) hbo_stream_init_r(&glob, (uint32_t)a->seed);
    mdraw_t md = {a->rng_kind, &glob, a->seed, 0, a->marker_offset};

    /* ---- warm state (not in the reference; hbo_warm): the prior constants above stay the cold run's ---- */
    if (a->warm) {
        vare_ = a->warm->vare;
        if (model_index == 1 || model_index == 4 || model_index == 6) varg = a->warm->varg;
        if (model_index == 6) for (int j = 0; j < n_fold; j++) vara_fold[j] = varg * fold_[j];
        if (model_index == 5) {
            lambda2 = a->warm->lambda2;
            lambda = sqrt(lambda2);
            if (a->warm->vargL) for (int i = 0; i < m; i++) vargL[i] = a->warm->vargL[i];
        }
        if (!fixpi) for (int j = 0; j < n_fold; j++) Pi[j] = a->warm->pi[j];
    }

    /* ---- :469-472 ---- */
    double mu_, mu = a->warm ? a->warm->mu : arma_sum(y, n) / n;
    double *yadj = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) yadj[i] = y[i] - mu;
    double *one = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) one[i] = 1.0;
    if (a->g_init) { /* warm start: g = g_init, u = X g, yadj = y - mu - X g */
        for (int i = 0; i < m; i++) {
            if (!vx[i] || a->g_init[i] == 0.0) continue;
            g[i] = a->g_init[i];
            if (snptracker) snptracker[i] = 1;
            col_axpy(&X, i, g[i], u);
        }
        for (int i = 0; i < n; i++) yadj[i] -= u[i];
    }
    double *r_RHS = n_levels ? (double *)malloc(sizeof(double) * n_levels) : NULL;
    double *estR_new = n_levels ? (double *)malloc(sizeof(double) * n_levels) : NULL;

    double xx, oldgi, gi, gi_, rhs, lhs, logdetV, acceptProb, uhat, v, vargi;
    int indistflag;
    if (g_threads > 1 && n >= 16384) team_start(g_threads, n); /* the threaded-BLAS stand-in: one team for the whole loop */
    double t_start = now_sec();
    int iter;

    /* ================================ MCMC, :477-917 ================================ */
    for (iter = 0; iter < niter; iter++) {
        if (a->rng_kind == HBO_RNG_PHILOX) {
            hbo_stream_init_philox(&glob, a->seed, ((uint64_t)HBO_PURPOSE_HOST << 56) | (uint64_t)iter, 0);
            md.sub = ((uint64_t)HBO_PURPOSE_MARKER << 56) | (uint64_t)iter;
        }
        const int tracing = (a->trace_rhs && iter == a->trace_iter);

        /* sample intercept, :479-482 */
        mu_ = -norm_sample(hbo_norm(&glob), arma_sum(yadj, n) / n, sqrt(vare_ / n));
        mu -= mu_;
        hbo_daxpy(n, mu_, one, yadj);

        /* covariates, :484-494 */
        for (int i = 0; i < nc; i++) {
            const double *dci = a->C + (size_t)i * n;
            oldgi = beta[i];
            v = cpc[i];
            rhs = hbo_ddot(n, dci, yadj);
            rhs += v * oldgi;
            gi = norm_sample(hbo_norm(&glob), rhs / v, sqrt(vare_ / v));
            gi_ = oldgi - gi;
            hbo_daxpy(n, gi_, dci, yadj);
            beta[i] = gi;
        }

        /* environmental random effects, :496-516 */
        for (int i = 0; i < nr; i++) {
            const int q0 = R_first[i], qr = R_q[i];
            const int *zi = zid + (size_t)i * n;
            for (int qi = 0; qi < qr; qi++) r_RHS[qi] = 0;
            for (int k = 0; k < n; k++) r_RHS[zi[k]] += yadj[k];          /* Z' yadj     */
            for (int qi = 0; qi < qr; qi++) r_RHS[qi] += zz[q0 + qi] * estR[q0 + qi]; /* + ZZ estR */
            for (int qi = 0; qi < qr; qi++) {
                double l = zz[q0 + qi] + vare_ / vrtmp[i];
                estR_new[qi] = norm_sample(hbo_norm(&glob), r_RHS[qi] / l, sqrt(vare_ / l));
            }
            for (int k = 0; k < n; k++) yadj[k] += estR[q0 + zi[k]] - estR_new[zi[k]];
            vrtmp[i] = (hbo_ddot(qr, estR_new, estR_new) + s2r * dfr) / hbo_chisq(&glob, qr + dfr);
            vr[i] = var_n1(estR_new, qr);
            for (int qi = 0; qi < qr; qi++) estR[q0 + qi] = estR_new[qi];
        }

        /* ------------------------- marker sweep, :586-816 ------------------------- */
        switch (model_index) {
        case 1: /* BayesRR :587-606 */
            for (int i = 0; i < m; i++) {
                if (!vx[i]) continue;
                xx = xpx[i];
                oldgi = g[i];
                rhs = col_dot(&X, i, yadj);
                rhs += xx * oldgi;
                v = xx + vare_ / varg;
                gi = norm_sample(md_norm(&md, i), rhs / v, sqrt(vare_ / v));
                gi_ = oldgi - gi;
                col_axpy(&X, i, gi_, yadj);
                gi_ *= -1;
                col_axpy(&X, i, gi_, u);
                g[i] = gi;
                if (tracing) { a->trace_rhs[i] = rhs; a->trace_cls[i] = 1; a->trace_g[i] = gi; }
            }
            varg = (hbo_ddot(m, g, g) + s2varg_ * dfvara_) / hbo_chisq(&glob, dfvara_ + m - nvar0);
            break;
        case 2: /* BayesA :607-626 */
            for (int i = 0; i < m; i++) {
                if (!vx[i]) continue;
                xx = xpx[i];
                oldgi = g[i];
                varg = (oldgi * oldgi + s2varg_ * dfvara_) / md_chisq(&md, i, dfvara_ + 1);
                rhs = col_dot(&X, i, yadj);
                rhs += xx * oldgi;
                v = xx + vare_ / varg;
                gi = norm_sample(md_norm(&md, i), rhs / v, sqrt(vare_ / v));
                gi_ = oldgi - gi;
                col_axpy(&X, i, gi_, yadj);
                gi_ *= -1;
                col_axpy(&X, i, gi_, u);
                g[i] = gi;
                if (tracing) { a->trace_rhs[i] = rhs; a->trace_cls[i] = 1; a->trace_g[i] = gi; }
            }
            break;
        case 3: /* BayesB / BayesBpi :627-670 */
            for (int j = 0; j < n_fold; j++) logpi[j] = log(Pi[j]);
            s[0] = logpi[0];
            for (int i = 0; i < m; i++) {
                if (!vx[i]) continue;
                xx = xpx[i];
                oldgi = g[i];
                varg = (oldgi * oldgi + s2varg_ * dfvara_) / md_chisq(&md, i, dfvara_ + 1);
                rhs = col_dot(&X, i, yadj);
                if (oldgi) rhs += xx * oldgi;
                lhs = xx / vare_;
                logdetV = log(varg * lhs + 1);
                uhat = rhs / (xx + vare_ / varg);
                s[1] = -0.5 * (logdetV - (rhs * uhat / vare_)) + logpi[1];
                acceptProb = 1 / (exp(s[0] - s[0]) + exp(s[1] - s[0]));
                indistflag = (md_unif(&md, i)) < acceptProb ? 0 : 1;
                snptracker[i] = indistflag;
                if (indistflag) {
                    v = xx + vare_ / varg;
                    gi = norm_sample(md_norm(&md, i), rhs / v, sqrt(vare_ / v));
                    gi_ = oldgi - gi;
                    col_axpy(&X, i, gi_, yadj);
                    gi_ *= -1;
                    col_axpy(&X, i, gi_, u);
                } else {
                    gi = 0;
                    if (oldgi) {
                        gi_ = oldgi;
                        col_axpy(&X, i, gi_, yadj);
                        gi_ *= -1;
                        col_axpy(&X, i, gi_, u);
                    }
                }
                g[i] = gi;
                if (tracing) { a->trace_rhs[i] = rhs; a->trace_cls[i] = indistflag; a->trace_g[i] = gi; }
            }
            fold_snp_num[1] = arma_sum(snptracker, m);
            fold_snp_num[0] = m - nvar0 - fold_snp_num[1];
            NnzSnp = (int)fold_snp_num[1];
            if (!fixpi) { /* rdirichlet_sample, src/stats.cpp:69-76 */
                double xn[HBO_MAX_FOLD], sx;
                for (int j = 0; j < n_fold; j++) xn[j] = hbo_gamma(&glob, fold_snp_num[j] + 1, 1.0);
                sx = arma_sum(xn, n_fold);
                for (int j = 0; j < n_fold; j++) Pi[j] = xn[j] / sx;
            }
            break;
        case 4: /* BayesC / BayesCpi :671-717 */
            for (int j = 0; j < n_fold; j++) logpi[j] = log(Pi[j]);
            s[0] = logpi[0];
            vargi = 0;
            for (int i = 0; i < m; i++) {
                if (!vx[i]) continue;
                xx = xpx[i];
                oldgi = g[i];
                rhs = col_dot(&X, i, yadj);
                if (oldgi) rhs += xx * oldgi;
                lhs = xx / vare_;
                logdetV = log(varg * lhs + 1);
                uhat = rhs / (xx + vare_ / varg);
                s[1] = -0.5 * (logdetV - (rhs * uhat / vare_)) + logpi[1];
                acceptProb = 1 / (exp(s[0] - s[0]) + exp(s[1] - s[0]));
                indistflag = (md_unif(&md, i)) < acceptProb ? 0 : 1;
                snptracker[i] = indistflag;
                if (indistflag) {
                    v = xx + vare_ / varg;
                    gi = norm_sample(md_norm(&md, i), rhs / v, sqrt(vare_ / v));
                    gi_ = oldgi - gi;
                    col_axpy(&X, i, gi_, yadj);
                    gi_ *= -1;
                    col_axpy(&X, i, gi_, u);
                    vargi += (gi * gi);
                } else {
                    gi = 0;
                    if (oldgi) {
                        gi_ = oldgi;
                        col_axpy(&X, i, gi_, yadj);
                        gi_ *= -1;
                        col_axpy(&X, i, gi_, u);
                    }
                }
                g[i] = gi;
                if (tracing) { a->trace_rhs[i] = rhs; a->trace_cls[i] = indistflag; a->trace_g[i] = gi; }
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
        case 5: /* BayesL :718-742 */
            for (int i = 0; i < m; i++) {
                if (!vx[i]) continue;
                xx = xpx[i];
                oldgi = g[i];
                rhs = col_dot(&X, i, yadj);
                rhs += xx * oldgi;
                v = xx + 1 / vargL[i];
                gi = norm_sample(md_norm(&md, i), rhs / v, sqrt(vare_ / v));
                if (fabs(gi) < 1e-6) gi = 1e-6;
                vargi = 1 / md_invgauss(&md, i, sqrt(vare_) * lambda / fabs(gi), lambda2);
                if (vargi >= 0) vargL[i] = vargi;
                gi_ = oldgi - gi;
                col_axpy(&X, i, gi_, yadj);
                gi_ *= -1;
                col_axpy(&X, i, gi_, u);
                g[i] = gi;
                if (tracing) { a->trace_rhs[i] = rhs; a->trace_cls[i] = 1; a->trace_g[i] = gi; }
            }
            {
                double shape = shape0 + m - nvar0;
                double rate = rate0 + arma_sum(vargL, m) / 2;
                lambda2 = hbo_gamma(&glob, shape, 1 / rate);
                lambda = sqrt(lambda2);
            }
            break;
        case 6: /* BayesR :743-815 */
            for (int j = 0; j < n_fold; j++) logpi[j] = log(Pi[j]);
            s[0] = logpi[0];
            varg = 0;
            for (int j = 1; j < n_fold; j++) vare_vara_fold[j] = vare_ / vara_fold[j];
            for (int i = 0; i < m; i++) {
                if (!vx[i]) continue;
                xx = xpx[i];
                oldgi = g[i];
                rhs = col_dot(&X, i, yadj);
                if (oldgi) rhs += xx * oldgi;
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
                double rval = md_unif(&md, i);
                for (int j = 0; j < n_fold; j++) {
                    acceptProb += stemp[j];
                    if (rval < acceptProb) { indistflag = j; break; }
                }
                snptracker[i] = indistflag;
                if (indistflag) {
                    v = xx + vare_vara_fold[indistflag];
                    gi = norm_sample(md_norm(&md, i), rhs / v, sqrt(vare_ / v));
                    gi_ = oldgi - gi;
                    col_axpy(&X, i, gi_, yadj);
                    gi_ *= -1;
                    col_axpy(&X, i, gi_, u);
                    varg += (gi * gi / fold_[indistflag]);
                } else {
                    gi = 0;
                    if (oldgi) {
                        gi_ = oldgi;
                        col_axpy(&X, i, gi_, yadj);
                        gi_ *= -1;
                        col_axpy(&X, i, gi_, u);
                    }
                }
                g[i] = gi;
                if (tracing) { a->trace_rhs[i] = rhs; a->trace_cls[i] = indistflag; a->trace_g[i] = gi; }
            }
            for (int j = 0; j < n_fold; j++) {
                double c = 0;
                for (int i = 0; i < m; i++) c += (snptracker[i] == j);
                fold_snp_num[j] = c;
            }
            NnzSnp = m - (int)fold_snp_num[0];
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

        /* genetic and residual variance, :819-823 */
        vara_ = var_n1(u, n);
        vare_ = (hbo_ddot(n, yadj, yadj) + s2vare_ * dfvare_) / hbo_chisq(&glob, n + dfvare_);

        /* PIP / WPPA, :826-845 */
        if (iter >= nburn) {
            if (snptracker)
                for (int i = 0; i < m; i++)
                    if (snptracker[i]) nzrate[i] += 1;
            if (nw) {
                memset(wflag, 0, nw);
                for (int i = 0; i < m; i++)
                    if (snptracker[i]) wflag[a->windindx[i] - 1] = 1;
                for (int w = 0; w < nw; w++) wppai[w] += wflag[w];
            }
            nzct++;
        }

        /* thinned store, :848-882 */
        if (iter >= nburn && (iter + 1 - nburn) % thin == 0) {
            if (o->s_mu) o->s_mu[count] = mu;
            mu_sum += mu;
            if (!fixpi)
                for (int j = 0; j < n_fold; j++) {
                    if (o->s_pi) o->s_pi[(size_t)count * n_fold + j] = Pi[j];
                    pi_sum[j] += Pi[j];
                }
            if (o->s_Vg) o->s_Vg[count] = vara_;
            if (o->s_Ve) o->s_Ve[count] = vare_;
            vara_sum += vara_;
            vare_sum += vare_;
            if (o->s_alpha) memcpy(o->s_alpha + (size_t)count * m, g, sizeof(double) * m);
            for (int i = 0; i < m; i++) alpha_sum[i] += g[i];
            for (int i = 0; i < nc; i++) {
                if (o->s_beta) o->s_beta[(size_t)count * nc + i] = beta[i];
                beta_sum[i] += beta[i];
            }
            double vt = vara_ + vare_;
            if (nr) {
                for (int i = 0; i < nr; i++) {
                    vt += vr[i];
                    if (o->s_Vr) o->s_Vr[(size_t)count * nr + i] = vr[i];
                    vr_sum[i] += vr[i];
                }
                for (int q = 0; q < n_levels; q++) estR_sum[q] += estR[q];
            }
            if (o->s_h2) o->s_h2[count] = vara_ / vt;
            hsq_sum += vara_ / vt;
            count++;
        }
        if (count == n_records) { iter++; break; }
    }
    o->loop_seconds = now_sec() - t_start;
    team_stop();
    o->iters_done = iter;
    memset(&o->last, 0, sizeof(o->last));
    o->last.mu = mu; o->last.vare = vare_; o->last.varg = varg; o->last.lambda2 = lambda2;
    for (int j = 0; j < n_fold && j < 8; j++) o->last.pi[j] = Pi[j];
    o->last.vargL = o->vargL_last;
    if (o->g_last) memcpy(o->g_last, g, sizeof(double) * m);
    if (o->vargL_last && vargL) memcpy(o->vargL_last, vargL, sizeof(double) * m);

    /* ============================ posterior assembly, :919-1040 ============================ */
    const double R = (double)n_records;
    o->Vg = vara_sum / R;
    o->Ve = vare_sum / R;
    o->h2 = hsq_sum / R;
    double Mu = mu_sum / R;
    o->mu = Mu;
    double *e = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) e[i] = y[i] - Mu;
    if (nc) {
        for (int i = 0; i < nc; i++) {
            double b = beta_sum[i] / R;
            if (o->beta) o->beta[i] = b;
            hbo_daxpy(n, -b, a->C + (size_t)i * n, e);
        }
    }
    for (int i = 0; i < m; i++) alpha_sum[i] /= R;
    for (int i = 0; i < m; i++) if (alpha_sum[i] != 0) col_axpy(&X, i, -alpha_sum[i], e);
    if (o->alpha) memcpy(o->alpha, alpha_sum, sizeof(double) * m);
    if (!fixpi) {
        for (int j = 0; j < n_fold; j++) Pi[j] = pi_sum[j] / R;
    } else if (o->s_pi) { /* :979-983 */
        for (int c = 0; c < n_records; c++) {
            o->s_pi[(size_t)c * n_fold + 0] = Pi[0];
            o->s_pi[(size_t)c * n_fold + 1] = Pi[1];
        }
    }
    if (o->pi) for (int j = 0; j < n_pi; j++) o->pi[j] = Pi[j];
    if (nr) {
        for (int i = 0; i < nr; i++) if (o->Vr) o->Vr[i] = vr_sum[i] / R;
        for (int q = 0; q < n_levels; q++) estR_sum[q] /= R;
        for (int i = 0; i < nr; i++)
            for (int k = 0; k < n; k++) e[k] -= estR_sum[R_first[i] + zid[(size_t)i * n + k]];
        if (o->r_est) memcpy(o->r_est, estR_sum, sizeof(double) * n_levels);
    }
    if (o->g) memcpy(o->g, u, sizeof(double) * n);   /* :1023: the FINAL-iteration u */
    if (o->e) memcpy(o->e, e, sizeof(double) * n);
    if (o->pip) {
        if (!nzrate) {
            for (int i = 0; i < m; i++) o->pip[i] = 1.0;
        } else {
            for (int i = 0; i < m; i++) {
                double p = nzrate[i] / nzct;
                if (p == 1) p = (nzct - 1) / (double)nzct;
                o->pip[i] = p;
            }
        }
    }
    if (nw && o->gwas) {
        for (int w = 0; w < nw; w++) {
            double p = wppai[w] / nzct;
            if (p == 1) p = (nzct - 1) / (double)nzct;
            o->gwas[w] = p;
        }
    }
    o->nzct = nzct;

    free(e); free(yadj); free(one); free(g); free(alpha_sum); free(u); free(xpx); free(vx);
    free(snptracker); free(nzrate); free(beta); free(cpc); free(beta_sum); free(vr); free(vrtmp);
    free(estR); free(R_first); free(R_q); free(zid); free(zz); free(estR_sum); free(vr_sum);
    free(vargL); free(wppai); free(wflag); free(r_RHS); free(estR_new);
    return 0;
}

/* ---------------------------------------------------------------------------------- */
/* reference src/read_bed.cpp:116-120 (code map), :147-167 (unpack), :182-230 (impute) */
int hbo_decode_bed(const uint8_t *bed, int64_t nbytes, int32_t nind, int32_t nsnp,
                   int impute, int8_t *out)
{
    const int64_t bpc = (nind + 3) / 4; /* bytes per SNP */
    if (nbytes < 3 + bpc * nsnp) return 1;
    if (bed[0] != 0x6c || bed[1] != 0x1b || bed[2] != 0x01) return 2;
    static const int8_t code[4] = {2, -128, 1, 0}; /* code[0]=2, code[1]=NA, code[2]=1, code[3]=0 */
    for (int32_t j = 0; j < nsnp; j++) {
        const uint8_t *p = bed + 3 + (int64_t)j * bpc;
        int8_t *col = out + (int64_t)j * nind;
        int miss = 0;
        for (int32_t i = 0; i < nind; i++) {
            int8_t gg = code[(p[i >> 2] >> (2 * (i & 3))) & 0x03];
            col[i] = gg;
            miss |= (gg == -128);
        }
        if (impute && miss) {
            int64_t counts[3] = {0, 0, 0};
            for (int32_t i = 0; i < nind; i++)
                if (col[i] >= 0 && col[i] <= 2) counts[col[i]]++;
            int64_t max = 0;
            int8_t major = 0;
            for (int k = 0; k < 3; k++)
                if (counts[k] > max) { max = counts[k]; major = (int8_t)k; }
            for (int32_t i = 0; i < nind; i++)
                if (col[i] == -128) col[i] = major;
        }
    }
    return 0;
}
