/*
 * hibayes_gpu.h — C ABI of libhibayes_gpu.so, the MI355X (gfx950) engine for hibayes'
 * individual-level per-SNP Gibbs sampler.
 *
 * Drop-in boundary.  The reference reaches this path through ONE symbol:
 *     SEXP _hibayes_Bayes(SEXP x 27)          reference src/RcppExports.cpp:16-50
 * generated from `Rcpp::List Bayes(arma::vec& y, arma::mat& X, std::string model, ...)`
 * at reference src/Bayes.cpp:60-88 and called from R/RcppExports.R:4-6 by ibrm()
 * (R/bayes.r:293-296).  hb_bayes_run() below takes that argument list one-for-one
 * (plain pointers and sizes; Nullable<T> -> pointer-or-NULL / has_* flags) and returns
 * the fields of the Rcpp::List built at src/Bayes.cpp:919-1040.  The Rcpp shim a
 * maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * All functions return 0 on success and a non-zero hb_status otherwise; the message for
 * the calling thread's last failure is hb_last_error().  Validation failures carry the
 * reference's own exception texts (src/Bayes.cpp:92-117, :293, :325, :357).  Nothing in
 * this library aborts the process, and nothing here falls back to a CPU implementation:
 * without a HIP device every compute entry point fails with HB_ERR_NO_DEVICE.
 */
#ifndef HIBAYES_GPU_H
#define HIBAYES_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HB_ABI_VERSION 6
#define HB_MAX_FOLD 8

typedef enum {
    HB_OK = 0,
    HB_ERR_INVALID = 1,     /* argument validation (reference exception text in hb_last_error) */
    HB_ERR_NO_DEVICE = 2,   /* no usable HIP device / runtime */
    HB_ERR_HIP = 3,         /* a HIP call or kernel failed */
    HB_ERR_UNSUPPORTED = 4, /* argument outside the GPU path (BSLMM, epsilon block, non-integer X) */
    HB_ERR_COMM = 5,        /* the multi-GPU all-reduce callback failed */
    HB_ERR_INTERRUPT = 6,   /* interrupt callback asked to stop */
    HB_ERR_ABORTED = 7      /* a wait inside the device pipeline timed out and the sweep was abandoned (hb_ctx_sweep_end); the
                               sampler (hb_bayes_run / hb_run_step) restores the state it saved before the sweep and replays
                               it — the draws are counter-based, so the replay is the same chain — and only fails with this
                               status when the replays time out as well */
} hb_status;

int hb_abi_version(void);
const char *hb_version(void);
const char *hb_last_error(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int hb_device_count(void);

/* ------------------------------------------------------------------------------------
 * Multi-GPU exchange.  Markers are sharded over `world` processes (one per GPU); once per
 * sweep every rank contributes a vector of `count` doubles living in DEVICE memory and
 * needs the element-wise sum over ranks back in place.  The host process supplies the
 * collective (torch.distributed / RCCL all_reduce in hibayes_amd/dist.py); the library
 * enqueues its kernels on `hip_stream` and calls this with that stream quiesced.
 * Return 0 on success.
 * ------------------------------------------------------------------------------------ */
typedef int (*hb_allreduce_fn)(void *device_buf, size_t count, void *user);
/* The same collective INSIDE the library: an RCCL communicator (librccl is dlopen()ed on first use). With
 * hb_bayes_args.comm set, the per-sweep exchange is one ncclAllReduce enqueued on the sweep's HIP stream — no host
 * synchronisation and no callback, so it also works from the R shim. Rank 0 calls hb_comm_unique_id(), ships the
 * HB_COMM_ID_BYTES to the other ranks by any means (torch.distributed / MPI / a file), every rank calls hb_comm_init(). */
#define HB_COMM_ID_BYTES 128
typedef struct hb_comm hb_comm;
int hb_comm_unique_id(void *id_out /* HB_COMM_ID_BYTES */);
int hb_comm_init(hb_comm **out, const void *id, int32_t rank, int32_t world, int32_t device);
int hb_comm_world(const hb_comm *c);   /* ranks RCCL reports for the communicator */
int hb_comm_rank(const hb_comm *c);
int hb_comm_selftest(hb_comm *c);     /* one 16-double all-reduce with a known answer, synchronised (run it under a deadline) */
void hb_comm_destroy(hb_comm *c);
typedef struct hb_ctx hb_ctx; /* device context of one genotype shard, see the engine API below */
/* called once per iteration from the calling thread; non-zero return stops the run
 * (the shim wires it to R_CheckUserInterrupt; the reference cannot be interrupted) */
typedef int (*hb_interrupt_fn)(void *user);
/* console lines the reference prints through Rcpp::Rcout (src/Bayes.cpp:393-461, :884-914,
 * :1042-1091); NULL = print to stdout when verbose */
typedef void (*hb_log_fn)(const char *line, void *user);

/* The scalars (and BayesL's per-marker variances) the loop at src/Bayes.cpp:477 carries from one iteration to the next, beside the
 * effects g: what hb_run_state() reports after an iteration is what a second run needs — together with g_init = the effects — to go
 * on in the same regime (same markers in the model, same pi / marker variance / residual variance) rather than re-admit markers from
 * the prior defaults for its first sweeps. The prior's own constants (s2varg_, dfvara_, s2vare_, rate0 ...: :319-374) are still
 * derived from the arguments exactly as in a cold run. Draws are addressed by (seed, iteration, marker) from iteration 0 of the
 * new run. bench.py's side legs and the stationary-state parity tests (tests/test_gpu_configs.py) start runs this way; the oracle
 * takes the same state (hbo_args.warm). */
typedef struct hb_warm_state {
    double mu;               /* intercept (:469)                                           */
    double vare;             /* residual variance vare_ (:823)                             */
    double varg;             /* shared marker variance (RR / C / R: :603, :713, :807); ignored by A / B / L */
    double lambda2;          /* BayesL (:738-741); ignored elsewhere                       */
    double pi[HB_MAX_FOLD];  /* class proportions in the CALLER's class order (as hb_run_info.pi); used unless the model fixes pi
                                (BayesB / BayesC, :669 / :716) or has none (RR / A / L)   */
    const double *vargL;     /* BayesL: m per-marker variances (:730), NULL = the prior's fill (:364-368) */
} hb_warm_state;

/* ------------------------------------------------------------------------------------
 * Arguments of Bayes(), reference src/Bayes.cpp:60-88, same order.
 * ------------------------------------------------------------------------------------ */
typedef struct hb_bayes_args {
    int32_t n;               /* y.n_elem == X.n_rows                                       */
    int32_t m;               /* X.n_cols (LOCAL columns when sharded)                      */
    const double *y;         /* arma::vec& y                                        (:61) */
    /* arma::mat& X (:62).  Exactly one of the two must be non-NULL.  X_f64 is the
     * reference's layout (column-major doubles) and must hold integers in [-127,127]
     * (0/1/2 or -1/0/1 genotype codes); X_i8 is the fast path that skips the double
     * blow-up (e.g. the bigmemory .bin mapping, R/read_plink.r:57-65).                     */
    const double *X_f64;
    int64_t ld_f64;          /* leading dimension in elements (>= n)                       */
    const int8_t *X_i8;
    int64_t ld_i8;
    const char *model;       /* std::string model                                    (:63) */
    const double *Pi;        /* arma::vec Pi                                         (:64) */
    int32_t n_pi;
    const double *Kival;     /* BSLMM only (:65) — must be NULL (HB_ERR_UNSUPPORTED)        */
    const double *Ki;        /* BSLMM only (:66) — must be NULL                             */
    const double *C;         /* Nullable<arma::mat> C, n x nc column-major           (:67) */
    int32_t nc;
    const char *const *R;    /* Nullable<CharacterMatrix> R, n x nr column-major      (:68) */
    int32_t nr;
    const double *fold;      /* Nullable<arma::vec> fold                             (:69) */
    int32_t n_fold;
    int32_t niter;           /* (:70) default 50000 */
    int32_t nburn;           /* (:71) default 20000 */
    int32_t thin;            /* (:72) default 5     */
    const double *epsl_y_J;  /* single-step epsilon block (:73-75) — must be NULL           */
    const void *epsl_Gi;
    const uint32_t *epsl_index;
    /* Nullable<double> dfvr..s2ve (:76-83): value used only when the has_ flag is set      */
    int32_t has_dfvr, has_s2vr, has_vg, has_dfvg, has_s2vg, has_ve, has_dfve, has_s2ve;
    double dfvr, s2vr, vg, dfvg, s2vg, ve, dfve, s2ve;
    const uint32_t *windindx; /* Nullable<arma::uvec> windindx, m, 1-based           (:84) */
    int32_t outfreq;          /* (:85) */
    int32_t threads;          /* (:86) accepted and ignored: there is no OpenMP region     */
    int32_t verbose;          /* (:87) */

    /* ---- additions that have no reference counterpart ---- */
    uint64_t seed;            /* replaces R's global RNG state (set.seed(), R/bayes.r:151)  */
    int32_t device;           /* HIP device ordinal                                         */
    int32_t panel;            /* markers per block-Gibbs panel: 0 = auto, else 64..1024     */
    int32_t precise;          /* arithmetic of the panel mat-vec x_j . yadj: 2 exact fixed point (fp64 residual as
                                 7 int8 digit planes, int8 x int8 -> int32; recommended), 1 fp64 FMA, 0 fp32 image */
    int32_t store_alpha;      /* keep MCMCsamples$alpha (m x n_records) — see hb_bayes_out  */
    /* marker sharding: this process owns global columns [m_offset, m_offset + m)           */
    int32_t rank, world;
    int64_t m_global, m_offset;
    hb_allreduce_fn allreduce;
    void *allreduce_user;
    void *exchange_buf;       /* optional caller-owned DEVICE buffer the collective runs on,
                                 >= hb_exchange_count(n) doubles (a torch tensor)            */
    hb_interrupt_fn interrupt;
    void *interrupt_user;
    hb_log_fn log;
    void *log_user;
    /* optional pre-loaded context: genotypes already resident on the device (then X_f64 and X_i8
     * must be NULL and n, m must match). Lets one upload serve several model fits.            */
    hb_ctx *ctx;
    /* warm start (no reference counterpart; ABI 2): m effects the chain starts from instead of g = 0
     * (src/Bayes.cpp:297). yadj = y - mu - X g_init, u = X g_init; classes restart as (g != 0). */
    const double *g_init;
    /* in-library RCCL collective (takes precedence over `allreduce`); with world == 1 the exchange path still runs,
     * which is how a one-GPU box exercises it */
    hb_comm *comm;
    /* the sharded sweep's `sync_every_blocks` knob (ABI 3): the shards exchange their residual deltas this many
     * times per sweep, after equal runs of mat-vec groups, instead of once at its end (0 or 1). Inside a run a shard still does
     * not see the other shards' moves; more exchanges = less of that staleness (and of the bias it causes where the shards'
     * markers are correlated), at one all-reduce and one pipeline drain each. Unsharded, the chain does not depend on it. */
    int32_t sync_blocks;
    /* resident genotype layout of the sweep (ABI 4; 0 = auto since round 6): 8 = int8 columns; 2 = 2 bits per genotype, PLINK's
     * own density (reference src/read_bed.cpp:116-167), expanded to bytes in registers inside the mat-vec — a
     * quarter of the bytes per sweep. Needs genotype codes 0..3 and the fixed-point mat-vec (precise = 2); same chain bit for bit
     * (the dot products are exact integers either way). A context the run creates drops its int8 copy once the Gram blocks
     * are built; with a pre-loaded ctx the layout is the context's (hb_ctx_set_layout).
     * 0 = AUTO (what the Rcpp shim and ibrm() pass): 2 bits where that is exact and the faster sweep — every code in 0..3, precise = 2,
     * BayesB / BayesBpi / BayesC / BayesCpi (450 against 213 sweeps/s at n = 50k, m = 500k) and BayesR with up to four classes (103 against 98
     * converged, 64.5 against 58 cold) at panel 512, the band and the packed copy fitting the device — int8 columns otherwise (the models in which
     * every marker moves, other genotype codes, small problems). hb_bayes_out.resident_bits / hb_run_info.resident_bits
     * report the choice. HB_NO_AUTO_BITS=1 in the environment keeps int8. */
    int32_t genotype_bits;
    /* Exact multi-GPU cross-check mode (ABI 4; SURVEY §8e "alternative rejected ... keep only as a correctness cross-check mode"):
     * shard the INDIVIDUALS instead of the markers. This process holds rows [row_offset, row_offset + n) of all m markers and of y
     * (n_global individuals in total; row_offset and every n but the last rank's multiples of 256). The digit-plane sums of every
     * panel mat-vec are integers, so their all-reduce is exact and order-independent: every rank runs THE single-GPU chain
     * (same decisions, same effects), whatever the model — the dense models (RR / A / L) that a marker-sharded sweep biases
     * included. Price: one all-reduce per panel (through `allreduce` / `comm`, with a host round trip: the per-panel kernels, no
     * pipeline), so it scales capacity, not throughput. Needs precise = 2; covariates / random effects are refused here. */
    int32_t shard_rows;
    int64_t n_global, row_offset;
    /* Warm hyper-parameter state (ABI 6; no reference counterpart): with g_init, the state a chain is CONTINUED from instead of
     * started at the prior defaults of src/Bayes.cpp:319-374. NULL = the reference's start. See hb_warm_state above. */
    const hb_warm_state *warm;
} hb_bayes_args;


/* number of doubles exchanged per sweep for n individuals: the residual delta (u moves by its negative) + 16 scalar sums */
size_t hb_exchange_count(int32_t n);

/* ------------------------------------------------------------------------------------
 * Result, reference src/Bayes.cpp:919-1040.  Arrays are caller-allocated; a NULL pointer
 * means "not wanted".  R = n_records = (niter - nburn) / thin.
 * ------------------------------------------------------------------------------------ */
typedef struct hb_bayes_out {
    double Vg, Ve, h2, mu;   /* results["Vg"], ["Ve"], ["h2"], ["mu"]          (:934-944) */
    int32_t n_records;
    int32_t nzct;
    int32_t nw;              /* number of GWAS windows = max(windindx)                     */
    int32_t n_levels;        /* total random-effect levels                                 */
    double *beta;            /* nc                                              (:951)    */
    double *alpha;           /* m                                               (:972)    */
    double *pi;              /* n_pi                                            (:984)    */
    double *Vr;              /* nr                                              (:925)    */
    double *r_est;           /* n_levels, results["r"]$Estimation              (:1020)    */
    int32_t *r_term_nlevels; /* nr: levels per term; level names are the sorted unique
                                strings of each R column (makeZ, :36-37)                   */
    double *g;               /* n: FINAL-iteration u, not a mean               (:1023)    */
    double *e;               /* n                                              (:1024)    */
    double *pip;             /* m                                              (:1032)    */
    double *gwas;            /* nw                                             (:1037)    */
    /* MCMCsamples: */
    double *s_Vg, *s_Ve, *s_h2, *s_mu;   /* 1 x R each                         (:937-945) */
    double *s_beta;          /* nc x R col-major                                (:952)    */
    double *s_alpha;         /* m x R col-major, only when args.store_alpha     (:973)    */
    double *s_pi;            /* n_pi x R                                        (:985)    */
    double *s_Vr;            /* nr x R                                          (:926)    */
    double *s_r;             /* n_levels x R                                    (:1021)   */
    /* extras: posterior SD of alpha from the running second moment (m), timings */
    double *alpha_sd;
    double setup_seconds;    /* upload + marker statistics + Gram precompute              */
    double loop_seconds;     /* the MCMC loop only                                        */
    int32_t iters_done;
    double mean_events;      /* mean number of markers whose effect changed per sweep     */
    int32_t sweeps_replayed; /* (ABI 5) sweeps that timed out on the device and were replayed from the saved state */
    int32_t resident_bits;   /* (was reserved_) the genotype layout the sweep read: 8 or 2 — what genotype_bits = 0 ("auto") chose */
    /* (ABI 6) the chain's state after its LAST iteration — what hb_bayes_args.g_init / .warm of a following run take to continue it:
     * the scalars in `last` (last.vargL is set to vargL_last), the effects in g_last (m, caller-allocated or NULL) and, for BayesL,
     * the per-marker variances in vargL_last (m, caller-allocated or NULL). */
    hb_warm_state last;
    double *g_last;
    double *vargL_last;
} hb_bayes_out;

/* The whole sampler: replaces Bayes() (reference src/Bayes.cpp:60-1094). */
int hb_bayes_run(const hb_bayes_args *args, hb_bayes_out *out);

/* The same sampler, stepwise: create (validation, upload, marker statistics, Gram, priors) ->
 * step (iterations of the loop at src/Bayes.cpp:477) -> finish (posterior assembly, :919-1040).
 * hb_bayes_run() is exactly create + step-until-finished + finish. The caller's arrays are copied
 * at create time. */
typedef struct hb_run hb_run;
typedef struct hb_run_info {
    int32_t iter;            /* iterations done */
    int32_t records;         /* thinned records stored */
    double nnz;              /* NumNZSnp of the last sweep */
    double vara, vare, varg, mu;
    double pi[HB_MAX_FOLD];
    double mean_events;      /* mean markers changed per sweep so far */
    double mean_misses;      /* mean row-cache misses per sweep so far */
    double mean_redo;        /* mean rolled-back chain rounds per sweep so far */
    double loop_seconds, setup_seconds, gram_seconds;
    int32_t sweeps_replayed; /* (ABI 5) sweeps whose device pipeline timed out and that were replayed from the saved state (HB_ERR_ABORTED) */
    int32_t resident_bits;   /* (was reserved_) the genotype layout the sweep read: 8 or 2 — what genotype_bits = 0 ("auto") chose */
    double lambda2;          /* (ABI 6) BayesL's lambda^2 after the last iteration (hb_warm_state.lambda2) */
} hb_run_info;
int hb_run_create(const hb_bayes_args *args, hb_run **out);
int hb_run_step(hb_run *r, int32_t nsteps, int32_t *finished);
int hb_run_state(hb_run *r, hb_run_info *info);
hb_ctx *hb_run_ctx(hb_run *r);
int hb_run_finish(hb_run *r, hb_bayes_out *out);
void hb_run_destroy(hb_run *r);

/* ------------------------------------------------------------------------------------
 * Summary-level sampler on a dense LD matrix: replaces SBayesD(), reference src/SBayesD.cpp:5-609, reached
 * through _hibayes_SBayesD (src/RcppExports.cpp:53, arity 18) from sbrm() (R/sbayes.r:213). Arguments in the reference's
 * order (:5-26); errors carry its exception texts (:30-63, :123-125, :154-156). The same six conditionals as Bayes() with the
 * right-hand sides kept in Gram space, r_hat += n (g_old - g_new) ldm[:, i] (:262-266): groups of 512 markers, the candidate-
 * round chain kernel with the LD matrix as its Gram matrix, a whole-chip column update per group (hb_sbayes.hpp).
 * ------------------------------------------------------------------------------------ */
typedef struct hb_sbayes_args {
    int32_t m;               /* ldm.n_rows == sumstat.n_rows                                              */
    const double *sumstat;   /* arma::mat sumstat (:6): m x 4 column-major — the columns sbrm() keeps,
                                R/sbayes.r:207: MAF, BETA, SE, NMISS; NaN = NA (the marker is then skipped, :102-104) */
    int64_t ld_sumstat;      /* leading dimension (>= m)                                                  */
    const double *ldm;       /* arma::mat ldm (:7): m x m column-major dense LD variance-covariance matrix */
    int64_t ld_ldm;
    const char *model;       /* (:8)  */
    const double *Pi;        /* (:9)  */
    int32_t n_pi;
    int32_t niter, nburn, thin; /* (:10-12) defaults 50000 / 20000 / 5 */
    const double *fold;      /* Nullable (:13) */
    int32_t n_fold;
    const uint32_t *windindx; /* Nullable (:14), m, 1-based */
    int32_t has_vg, has_dfvg, has_s2vg, has_ve, has_dfve, has_s2ve; /* Nullable<double> (:15-20) */
    double vg, dfvg, s2vg, ve, dfve, s2ve;
    int32_t outfreq, threads, verbose; /* (:21-23); threads is accepted and ignored */
    /* ---- additions without a reference counterpart ---- */
    uint64_t seed;           /* replaces R's global RNG state (set.seed(), R/sbayes.r:132) */
    int32_t device;
    int32_t store_alpha;     /* keep MCMCsamples$alpha (m x n_records) */
    hb_interrupt_fn interrupt;
    void *interrupt_user;
    hb_log_fn log;
    void *log_user;
} hb_sbayes_args;

/* Result list of SBayesD(), src/SBayesD.cpp:541-580. Caller-allocated arrays, NULL = not wanted; R = (niter - nburn) / thin. */
typedef struct hb_sbayes_out {
    double Vg, Ve, h2;
    int32_t n_records, nzct, nw;
    int32_t n;               /* population size the sampler used: mean of the finite NMISS, truncated (:33-34) */
    int32_t count_y;         /* markers with summary statistics (:111) */
    double *alpha;           /* m  (:553) */
    double *pi;              /* n_pi (:566) */
    double *pip;             /* m  (:575) */
    double *gwas;            /* nw (:580) */
    double *s_Vg, *s_Ve, *s_h2; /* 1 x R (:547-549) */
    double *s_alpha;         /* m x R column-major, only with store_alpha (:554) */
    double *s_pi;            /* n_pi x R (:567) */
    double *r_hat;           /* extras: the Gram-space right-hand side and the effects after the last sweep (m each) */
    double *g_last;
    double setup_seconds, loop_seconds;
    int32_t iters_done;
    double mean_events;
} hb_sbayes_out;

int hb_sbayes_run(const hb_sbayes_args *args, hb_sbayes_out *out);

/* ====================================================================================
 * Fine-grained engine API.  hb_bayes_run() is built on it; the parity tests and bench.py
 * drive the device pieces through it one at a time.  A context owns all device state of
 * one genotype shard; calls on one context must be serialised by the caller.
 * ==================================================================================== */
typedef struct hb_ctx_params {
    int32_t device;
    int32_t n, m;            /* individuals, local markers */
    int32_t panel;           /* 0 = auto */
    int32_t precise;
    int64_t m_offset;        /* global index of local marker 0 (RNG addressing) */
    uint64_t seed;
} hb_ctx_params;

int hb_ctx_create(const hb_ctx_params *p, hb_ctx **out);
void hb_ctx_destroy(hb_ctx *c);
int hb_ctx_panel(const hb_ctx *c);
int64_t hb_ctx_ld(const hb_ctx *c);     /* device leading dimension of X in bytes */

/* Genotypes -> device int8, column-major. ncols columns starting at col0. */
int hb_ctx_upload_genotype_i8(hb_ctx *c, const int8_t *X, int64_t ld, int32_t col0, int32_t ncols);
/* doubles are checked for integrality and range, never rounded silently */
int hb_ctx_upload_genotype_f64(hb_ctx *c, const double *X, int64_t ld, int32_t col0, int32_t ncols);
/* PLINK .bed (SNP-major, 2 bits/genotype) decoded ON DEVICE with the code map of reference
 * src/read_bed.cpp:116-120 and its major-genotype imputation (:182-230); `rows` (n long,
 * indices into the .bed's nind individuals) selects/reorders individuals as ibrm() does
 * (R/bayes.r:165, :286-291); NULL = first n individuals. bed excludes no header: pass the
 * whole file image including the 3 magic bytes. */
int hb_ctx_upload_bed(hb_ctx *c, const uint8_t *bed, int64_t nbytes, int32_t nind,
                      const int32_t *rows, int32_t col0, int32_t ncols);
/* synthetic genotypes generated on device (SURVEY §8 d): p_j ~ U(0.05,0.5),
 * x ~ Binomial(2,p_j), every mono_every-th column forced monomorphic (0 = never) */
int hb_ctx_generate_genotype(hb_ctx *c, uint64_t seed, int32_t mono_every);
int hb_ctx_download_genotype(hb_ctx *c, int8_t *X, int64_t ld, int32_t col0, int32_t ncols);
/* Resident layout the sweep's kernels read: bits = 8 (int8 columns) or 2 (2 bits per genotype: 16 individuals per 32-bit word,
 * expanded in registers — hb_dotq2.hpp; codes must be 0..3). Packing happens on the device from the int8 columns. keep_int8 = 0
 * frees the int8 copy afterwards (every kernel of the run — mat-vec, residual update, X*alpha, the GEBV sample matrix, genotype
 * download — then reads the packed form; the Gram blocks must have been built for the widest geometry the run will use, a
 * rebuild unpacks panel by panel into a scratch buffer). bits = 8 on a context that dropped its int8 copy unpacks it again. */
int hb_ctx_set_layout(hb_ctx *c, int32_t bits, int32_t keep_int8);
int hb_ctx_get_layout(const hb_ctx *c, int32_t *bits, int32_t *int8_resident);
/* Which kernel computes the panel mat-vec on 2-bit resident genotypes (ABI 5; all three produce the same exact integers, hence the same
 * chain bit for bit): 2 = k_dotq2m, the seven digit planes of the residual as a skinny int8 GEMM on the matrix cores
 * (v_mfma_i32_16x16x64_i8) — THE DEFAULT since round 5 (12 us per 3584-column launch at n = 50k against 22: with the residual held as
 * seven int8 planes the product [columns x individuals] x [individuals x 7] is a dense integer contraction, and the v_dot4 form is bound
 * by vector-ALU issue, not by memory); 0 = k_dotq2, lane = column, v_dot4_i32_i8 (north_star's literal "coalesced loads with LDS-staged
 * wavefront reductions, no MFMA"; the default until round 4, kept selectable and measured beside the default by bench.py); 1 = k_dotq2r,
 * individuals across the lanes. Also HB_DOTQ2_KIND. The int8-column layout (the library's default layout) always runs k_dotq: v_dot4, no MFMA. */
int hb_ctx_set_matvec_kernel(hb_ctx *c, int32_t kind);

/* xpx_i = sum x^2, vx_i = var(x_i) (N-1), reference src/Bayes.cpp:310-317; integer-exact */
int hb_ctx_marker_stats(hb_ctx *c, double *xpx, double *vx, double *sumvx, int32_t *nvar0);
/* Pipeline geometry of the sweep: pipeline 0 = one kernel per step and panel, 1 = persistent
 * chain workgroup overlapped with the mat-vec stream; `lookahead` mat-vec groups of `dotgroup` panels each run
 * ahead of the chain. Results do not depend on it: the same chain — identical decisions and move lists, effects to 1e-9 (two
 * geometries add the band corrections to a right-hand side in different orders). A band wider than the stored one rebuilds the Gram blocks.
 * What hb_bayes_run picks for a context it creates (end of round 6): (1, 2, 7) for BayesB / BayesC, with (1, 2, 2) while many markers move; (1, 2, 2) /
 * (1, 2, 1) for BayesR; (1, 2, 2) for BayesRR / A / L at panel 512. A geometry the band limit cannot hold is narrowed (hb_ctx_get_pipeline tells): at most 20
 * panels of band, 27 with the forward workgroup beside the group chain — panel 512: (3, 7) and (2, 8). */
int hb_ctx_set_pipeline(hb_ctx *c, int32_t pipeline, int32_t lookahead, int32_t dotgroup);
/* per-panel Gram blocks G = X_p' X_p (int32, exact) plus the look-ahead band; also reports seconds spent */
int hb_ctx_build_gram(hb_ctx *c, double *seconds);
/* rows x cols window of panel p's Gram (row-major int32, P x P) for exactness tests */
int hb_ctx_download_gram(hb_ctx *c, int32_t panel_index, int32_t *G);
/* band block l of panel p (row-major int32, P x P): G[k][t] = x_{(p-l)P+k} . x_{pP+t}, l = 0..band; the look-ahead
 * pipeline folds the moves of panel p-l into panel p's right-hand sides with it */
int hb_ctx_download_gram_band(hb_ctx *c, int32_t panel_index, int32_t l, int32_t *G);
/* NULL when the persistent pipeline is available; otherwise the reason the context fell back to the event-ordered
 * per-panel kernels (kernels on two streams are not co-resident here: AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING, a
 * counter-collecting profiler ...). Probed once at hb_ctx_create(); hb_ctx_set_pipeline(1, ...) then keeps pipeline 0. */
const char *hb_ctx_pipeline_note(const hb_ctx *c);
/* Geometry by regime. The band Gram blocks are stored once, for the widest band built so far; hb_ctx_set_pipeline() to any
 * geometry whose band fits reuses them (and the sweeps captured for each geometry are cached). With adaptive != 0,
 * hb_bayes_run / hb_run_step choose the geometry of each sweep of a point-mass model (BayesB/C) from the number of markers that
 * moved in the previous one: a narrow band while many move (every move costs one band row per block), the wide band with its
 * big mat-vec launches once few do. Same chain either way. hb_bayes_run does this by itself for a context it creates. */
int hb_ctx_set_adaptive(hb_ctx *c, int32_t adaptive);
/* current geometry: pipeline flag, look-ahead groups, panels per mat-vec launch, band width (blocks l = 1..band) */
int hb_ctx_get_pipeline(const hb_ctx *c, int32_t *pipeline, int32_t *lookahead, int32_t *dotgroup, int32_t *band);
/* move lists of the last sweep: ev_count[npanels]; ev_idx / ev_delta are [npanels][P] with ev_count[p] valid entries
 * (marker index inside the panel, change of its effect), in the order the chain applied them */
int hb_ctx_get_events(hb_ctx *c, int32_t *ev_count, int32_t *ev_idx, double *ev_delta);

/* residual yadj and u = Xg (n each) */
int hb_ctx_set_residual(hb_ctx *c, const double *yadj, const double *u);
int hb_ctx_get_residual(hb_ctx *c, double *yadj, double *u);
int hb_ctx_set_effects(hb_ctx *c, const double *g, const uint8_t *tracker, const double *vargL);
int hb_ctx_get_effects(hb_ctx *c, double *g, uint8_t *tracker, double *vargL);
/* d[j] = x_j . yadj for ncols columns from col0 — the panel mat-vec kernel on its own */
int hb_ctx_dot(hb_ctx *c, int32_t col0, int32_t ncols, double *d);

/* out (n) = X * alpha (m): the product behind e = y - ... - X*alpha, reference src/Bayes.cpp:971 */
int hb_ctx_matvec(hb_ctx *c, const double *alpha, double *out);

/* out (n x R, column-major, leading dimension ldo) = X * A, A (m x R, leading dimension ldA) on the host: the GEBV sample
 * matrix MCMCsamples$g = M %*% MCMCsamples$alpha of reference R/bayes.r:303-305 (then g$gebv = its row means, :308).
 * Eight records per pass, only over the columns with a non-zero effect in any of them. */
int hb_ctx_matmul(hb_ctx *c, const double *A, int64_t ldA, int32_t R, double *out, int64_t ldo);

/* helpers for the host blocks that share yadj (reference src/Bayes.cpp:479-516) */
int hb_ctx_residual_sums(hb_ctx *c, double *sum_r, double *sum_r2);
int hb_ctx_residual_shift(hb_ctx *c, double a);                       /* yadj += a        */
int hb_ctx_set_covariates(hb_ctx *c, const double *Cmat, int32_t nc); /* n x nc, col-major */
int hb_ctx_cov_dot(hb_ctx *c, int32_t i, double *out);                /* C_i . yadj        */
int hb_ctx_cov_axpy(hb_ctx *c, int32_t i, double a);                  /* yadj += a C_i     */
int hb_ctx_set_levels(hb_ctx *c, const int32_t *zid, int32_t nr, const int32_t *nlev);
int hb_ctx_level_sums(hb_ctx *c, int32_t term, double *sums);         /* Z_t' yadj         */
int hb_ctx_level_axpy(hb_ctx *c, int32_t term, const double *delta);  /* yadj += Z_t delta */

/* The covariate and random-effect blocks of one iteration (src/Bayes.cpp:484-516) as device kernels with no host
 * synchronisation: the caller draws the deviates first, in the reference's order (they do not depend on the data) —
 * z_beta[nc], then per term its level normals (z_levels, all terms concatenated) and chisq[term] = chisq_sample(q_t + dfr) —
 * and reads the state back whenever it needs it (hb_ctx_blocks_state synchronises). set_covariates / set_levels first. */
int hb_ctx_blocks_setup(hb_ctx *c, const double *cpc /* nc */, const double *zz /* levels */, const double *vrtmp0 /* nr */);
int hb_ctx_blocks_step(hb_ctx *c, double vare, const double *z_beta, const double *z_levels, const double *chisq, double dfr, double s2r);
int hb_ctx_blocks_state(hb_ctx *c, double *beta, double *estR, double *vrtmp, double *vr);

/* One marker sweep (reference src/Bayes.cpp:586-816) plus the two reductions behind
 * :819 and :823.  Hyper-parameters come from the host, per-SNP draws happen on device. */
typedef struct hb_sweep_in {
    int32_t model_index;     /* 1 RR, 2 A, 3 B/Bpi, 4 C/Cpi, 5 L, 6 R   (:97)              */
    int32_t n_fold;
    int64_t iter;            /* addresses the marker RNG stream                             */
    double vare;
    double varg;             /* shared marker variance (RR, C, R)                           */
    double s2varg_df;        /* s2varg_ * dfvara_ (A, B)                                    */
    double dfvara;           /* (A, B)                                                      */
    double logpi[HB_MAX_FOLD];
    double fold[HB_MAX_FOLD];
    double vara_fold[HB_MAX_FOLD];
    double lambda, lambda2;  /* (L)                                                         */
    int32_t count_pip;       /* iter >= nburn: accumulate nzrate / window flags             */
    int32_t store;           /* thinned record: accumulate alpha moments                    */
} hb_sweep_in;

typedef struct hb_sweep_out {
    double sum_g2;            /* g.g (RR) | sum g^2 of included (C) | sum g^2/fold (R)      */
    double class_count[HB_MAX_FOLD]; /* markers per class among polymorphic ones, class 0
                                        EXCLUDES monomorphic markers                         */
    double sum_vargL;         /* (L) sum over markers incl. untouched monomorphic ones      */
    double sum_r, sum_r2;     /* over yadj after the sweep                                   */
    double var_u;             /* var(u), N-1                                                 */
    double n_events;          /* markers whose effect changed                                */
    double n_cache_miss;      /* ... of which the Gram row was not in the chain's LDS row cache */
    double n_redo;            /* speculative chain rounds that were rolled back and repeated    */
} hb_sweep_out;

int hb_ctx_sweep(hb_ctx *c, const hb_sweep_in *in, hb_sweep_out *out);
/* The same sweep in pieces, for a caller that exchanges residual deltas between them (hb_bayes_args.sync_blocks does exactly
 * this): block b of nblocks enqueues the panels of its share of the mat-vec groups as a self-contained pipeline — block 0 also
 * prepares the sweep, the last block closes it — and returns without waiting; hb_ctx_sweep_end() then fetches the sweep's sums.
 * The residual may be modified on the context's stream between two blocks (hb_ctx_set_residual / an all-reduce on it).
 * nblocks == 1 is hb_ctx_sweep() split into begin and end. */
int hb_ctx_sweep_range(hb_ctx *c, const hb_sweep_in *in, int32_t block, int32_t nblocks);
int hb_ctx_sweep_end(hb_ctx *c, hb_sweep_out *out);
/* nzrate counters (m, as doubles), alpha sum / sum of squares over stored records */
int hb_ctx_get_counters(hb_ctx *c, double *nzrate, double *alpha_sum, double *alpha_sq);
int hb_ctx_set_windows(hb_ctx *c, const uint32_t *windindx, int32_t nw);
int hb_ctx_get_windows(hb_ctx *c, double *wppa);

/* device timing of the last hb_ctx_sweep: milliseconds by phase, averaged per launch */
typedef struct hb_sweep_timing {
    double total_ms;
    double dot_ms;       /* summed over panel launches */
    int32_t dot_launches;
    double chain_ms;
    double update_ms;
    double other_ms;
} hb_sweep_timing;
int hb_ctx_last_timing(hb_ctx *c, hb_sweep_timing *t);
/* measurement helper: the panel mat-vec launches of one sweep, issued back to back exactly as the sweep issues them
 * (same columns per launch), timed with HIP events on the context's stream; average milliseconds per launch */
int hb_ctx_time_matvec(hb_ctx *c, int32_t reps, double *avg_ms, int32_t *launches_per_sweep, int32_t *cols_per_launch);
/* measurement helper ("fraction of a measured streaming-read kernel on the same buffer"): the resident genotype buffer
 * (int8 columns, or the 2-bit words when that layout is set) read once front to back by a plain 16-bytes-per-lane kernel, nothing
 * computed, nothing written; average milliseconds per pass over `reps` passes (one untimed pass first) and the bytes of one pass */
int hb_ctx_time_stream_read(hb_ctx *c, int32_t reps, double *avg_ms, int64_t *bytes);
/* on: bit 0 HIP-event timing of the per-panel kernels (hb_ctx_last_timing), bit 1 cycle stamps inside the chain kernel,
 * bit 2 chain-alone diagnostic, bit 3 in-situ stamps of the mat-vec launches (hb_ctx_matvec_stamps) */
int hb_ctx_set_profiling(hb_ctx *c, int32_t on);
/* In-situ duration of the dominant kernel. With hb_ctx_set_profiling(c, 8) every block of every mat-vec launch of a sweep
 * records the device's constant 100 MHz clock at its start and at its end — the launches of the REAL sweep, with the chain
 * workgroup and the update rows running beside them, not an isolated replay. A launch's duration is max(end) - min(start)
 * over its blocks (what a kernel trace reports); the statistics are over the launches of the last sweep. span_ms = first
 * launch's start to last launch's end, so launches * avg_ms <= span_ms <= the sweep's wall time. */
typedef struct hb_launch_stats {
    int32_t launches;        /* full-width mat-vec launches of the last sweep (the statistics are over these) */
    int32_t launches_all;    /* ... all of them (the last launch of a sweep may cover fewer columns) */
    int32_t blocks;          /* blocks of the last launch looked at (update rows + finalize + tiles) */
    int32_t cols_per_launch;
    double avg_ms, min_ms, max_ms, sum_ms, span_ms;
} hb_launch_stats;
int hb_ctx_matvec_stamps(hb_ctx *c, hb_launch_stats *out);
/* Debug hook for the replay of an aborted sweep (ABI 5): the next `times` sweeps the persistent pipeline runs on the context are
 * aborted in mid-flight — the abort flag a timed-out waiter raises is set once the chain has published `panel` panels — so
 * hb_ctx_sweep_end() returns HB_ERR_ABORTED for them; hb_run_step() restores the saved state and replays (twice on the pipeline,
 * then on the per-panel kernels, which the hook does not touch). times <= 0 disarms it. */
int hb_ctx_debug_inject_abort(hb_ctx *c, int32_t panel, int32_t times);

#ifdef __cplusplus
}
#endif
#endif
