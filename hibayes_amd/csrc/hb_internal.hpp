// hb_internal.hpp — private declarations shared by the engine translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/hibayes_gpu.h"

// thread-local last error (hb_last_error)
void hb_set_error(const std::string &msg);
int hb_fail(int status, const std::string &msg);

#define HB_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return hb_fail(HB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

// layout of the per-sweep scalar block the kernels accumulate into / the host reads back
#define HB_ND 7 /* int8 digits of the fixed-point residual: 55 bits + sign */
// Words that one workgroup writes through and others read behind a flag — the panels' move counts, the groups' bounds on max |yadj| —
// live ONE PER 128-BYTE LINE: a reader that touched the line for panel p while the chain was writing panel p + 1's word into it could
// otherwise be left with a copy that shows the neighbour's old value (the stale-line hazard of DESIGN.md §9.0), and such a word is not
// a sentinel a later look would recognise. ev_count[p] is at ev_count[p * HB_EVS], mb[i] at mb[i * HB_MBS].
#define HB_EVS 32
#define HB_MBS 16
#define HB_LSTAMP_BLOCKS 4608

enum {
    HB_ACC_SUMG2 = 0,
    HB_ACC_COUNT0 = 1, // .. +HB_MAX_FOLD
    HB_ACC_EVENTS = 9,
    HB_ACC_SUMVARGL = 10,
    HB_ACC_SUMR = 11,
    HB_ACC_SUMR2 = 12,
    HB_ACC_VARU = 13,
    HB_ACC_MISS = 14, // moves whose Gram row was not in the LDS row cache
    HB_ACC_REDO = 15, // chain rounds rolled back and repeated (a non-candidate crossed its threshold)
    HB_ACC_N = 16
};

struct hb_ctx {
    int device = 0;
    int n = 0, m = 0, P = 0, npanels = 0, m_pad = 0;
    int L = 0;     // Gram band: blocks l = 0..L per panel (L = Lv + D - 1)
    int Lv = 0;    // version lag: the mat-vec of group g reads the residual with panels <= g*D - Lv - 1 applied
    int D = 1;     // panels per mat-vec launch
    int NB = 1;    // residual versions kept = Lv + D
    int pipeline = 0;              // 0: serial kernels per panel; 1: persistent chain workgroup + flags
    int chain_kind = 1;            // group-granular chain (k_chain_group): bit 0 BayesB/C, bit 1 the dense models at one panel per group; 0 = k_chain_persist everywhere (HB_CHAIN=panel / all / <bits>)
    int num_cus = 256;             // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    bool warm_group = false;       // k_warm beside the group chain (HB_WARM_GROUP=1)
    int warm_g = 0;                // k_warm workgroups per XCD beside the wide group chain with k_fwd (HB_WARM_G)
    bool dense_chain = true;       // BayesRR / A / L at panel 512: k_chain_dense + k_fold_dense (hb_chain_dense.hpp; HB_DENSE=0: k_chain_persist)
    bool fwd_group = true;         // k_fwd beside the group chain: a second workgroup folds a move's rows for the group after next (HB_FWD=0: off)
    bool concurrent = true;        // kernels on two streams were seen running at the same time (probe at create)
    std::string pipeline_note;     // why the persistent pipeline is off, when it is
    unsigned int *flags = nullptr; // [0] chain_done (panels whose moves are published), [1] abort
    unsigned int *h_flags = nullptr;
    int *hot_slot = nullptr, *hot_list = nullptr, *hot_n = nullptr; // per-sweep row-cache lists (k_hotlist): hot_list is packed, 256 ints per panel
    float *thr0f = nullptr;                                          // per-sweep opening filter of the chain (k_hotlist)
    int4 *opn = nullptr;                                             // per-sweep, 16 bytes per marker: {candidate threshold (double), opening filter (float), gB}: what k_chain_group's opening needs that no other workgroup of the sweep writes — staged in its LDS a group ahead (k_hotlist writes it)
    int64_t ld = 0; // bytes per genotype column on device (multiple of 256)
    int precise = 0;
    int64_t m_offset = 0;
    uint64_t seed = 0;
    hipStream_t stream = nullptr;      // mat-vec stream (and everything outside the sweep)
    hipStream_t s_chain = nullptr;     // serial chain kernels
    hipStream_t s_upd = nullptr;       // residual updates
    hipStream_t s_warm = nullptr;      // k_warm where k_fwd has s_upd (BayesR)
    // round 6, overlapped launch stream (HB_OVERLAP): the mat-vec launches of a sweep alternate between `stream` and s_t2, the residual updates are kernels of
    // their own on s_uk and the finalize kernels on s_fk — every dependency a graph edge, two launches in flight (hb_kernels.hip: enqueue_sweep_pipeline)
    hipStream_t s_t2 = nullptr, s_uk = nullptr, s_fk = nullptr;
    std::vector<hipEvent_t> ev_ot, ev_ou; // per mat-vec group: its tiles are done / its residual update is done
    int overlap = 0;                      // 0 off, 1 on (HB_OVERLAP)
    // round 6: the persistent mat-vec (hb_mvp.hpp; HB_MVP): digit planes and exponents per residual VERSION (rq / vexp grown to rq_slots slots), hand-over counters
    int mvp = 0;
    int rq_slots = 8;
    unsigned *mvp_ho = nullptr;
    std::vector<hipEvent_t> ev_dot, ev_chain, ev_upd; // cross-stream dependencies of one sweep
    hipEvent_t ev_fork = nullptr;

    int8_t *X = nullptr;
    // 2-bit resident layout (hb_dotq2.hpp; hb_ctx_set_layout): layout == 2 makes the sweep's kernels read X2; X may then be dropped
    uint32_t *X2 = nullptr;
    int64_t ld2 = 0;     // bytes per column of X2 = 128 * ceil(ld / 512)
    int layout = 8;      // 8: int8 columns, 2: 2-bit columns
    int dotq2_cpl = 1, dotq2_tiles = 1600, dotq2_rs = 256; // (tiles per full-width k_dotq2 launch, HB_DOTQ2_TILES: 1568 of seven stages at n = 50k — with two waves per SIMD (Q2_TWO_PER_SIMD) 2048 waves are resident, and tiles + update rows + the chain's and k_fwd's compute units must fit; until that cap 2000 -> 1848 tiles of six stages: 296 against 300 sweeps/s)
    // which kernel computes the panel mat-vec on 2-bit resident genotypes (HB_DOTQ2_KIND / hb_ctx_set_matvec_kernel; all three give the same exact integers):
    // 2 (default since round 5) k_dotq2m, the seven digit planes as a skinny int8 GEMM on the matrix cores — 12.0 us per 3584-column launch isolated; 0 k_dotq2,
    // lane = column through LDS, v_dot4 (22 us: VALU-issue-bound; the default until round 4); 1 k_dotq2r, individuals across the lanes, no LDS, NC columns per tile (26 us)
    int dotq2_kind = 2, dotq2_nc = 16;
    double *xpx = nullptr, *vx = nullptr, *g = nullptr, *vargL = nullptr;
    double *s1 = nullptr; // column sums of the resident rows (k_stats), for the row-sharded mode's global statistics
    double *alpha_sum = nullptr, *alpha_sq = nullptr;
    uint8_t *tracker = nullptr;
    uint32_t *nzrate = nullptr;
    double *r = nullptr, *u = nullptr;
    float *r32 = nullptr;
    // precise == 2: exact fixed-point mat-vec. Residual version slot s is also kept as HB_ND balanced
    // base-256 digit planes rq[s][k][ld] (int8) of q = rint(yadj * 2^vexp[s]), |q| <= 2^54.
    int8_t *rq = nullptr;
    int *vexp = nullptr;          // [8] exponent of the digits in each residual slot
    int *gexp = nullptr;          // [npanels + 1] exponent the mat-vec launch of each group / panel used (for its finalize)
    double *mb = nullptr;         // [npanels + 2]: mb[0] = max |yadj| at sweep start, mb[1 + h] = bound on max |yadj| after group / panel h
    long long *accq = nullptr;    // [HB_ND][m_pad] exact digit-plane sums of the current sweep's mat-vecs
    int32_t *gram = nullptr;
    size_t gram_cap = 0; // ints allocated
    // Round 5: the band once more as "rank one + int16 residual" — G[k][j] = ga[k] * gB[j] + gram16[k][j] exactly, with ga = rint(s1 / 256),
    // gB = rint(256 s1 / n) (s1: column sums): the product is the part of x_k . x_j that every pair of markers shares (n mean_k mean_j), what is
    // left is n cov(k, j) plus rounding, a few hundred for unlinked markers and at most n var for a pair in full LD. The group chain and k_fwd
    // fold a move's rows from it: half the bytes of the phase that is 44 % of their time. Built when every residual fits (else null: the int32 band serves).
    int16_t *gram16 = nullptr;
    size_t gram16_cap = 0;
    int32_t *ga = nullptr, *gB = nullptr, *gcmax = nullptr; // G[k][j] = ga[k] gB[j] + c[k][j], |c[k][j]| <= gcmax[k]: the group chain's certificate (hb_build_gcert)
    bool gcert_ok = false, gcert_on = true;                 // (HB_CERT=0: off)
    int *g16_flag = nullptr;
    bool gram16_ok = false, gram16_on = false; // (HB_GRAM16=1: on. OFF by default: measured slower than the int32 band both ways it was read)
    bool env_pinned = false;
    int dot_lds = 0;     // dynamic LDS bytes requested by each mat-vec workgroup: caps the workgroups resident per CU
    int q2m_ct = 4, q2m_g = 0, q2m_sc = 1; // k_dotq2m's shape (HB_Q2M_CT / _G / _SC): column tiles of 16 per wave; stages requested together (1, 2) or 512-individual stages of whole-line
                                           // DMA pieces (0, the default since round 5: 12.3 against 15.3 us per launch; 3: the same with conflict-free lane order); per-scale accumulators
    double candf = 1.0;  // chain candidates: markers at zero with q >= candf * thr0 (tuning knob; <= 1)
    double kappa = 3.0; // row-cache prediction: markers with thr0 <= kappa * xx * vare get their Gram row prefetched
    bool gram_ready = false, stats_ready = false;
    int Lg = 0;       // band blocks per panel (minus one) the stored gram[] was built with: every geometry with L <= Lg runs on it
    bool adaptive = false; // hb_run picks the geometry per sweep from the number of moves (hb_ctx_set_adaptive)
    int home_lv = 0, home_d = 0; // the geometry its owner gave an adaptive context (hb_ctx_set_pipeline): a run that left it in its narrow geometry is followed by one that starts from this again
    int *xinfo = nullptr; // device: [0]=min value, [1]=max value over X
    int xmin = 0, xmax = 0;

    double *thr = nullptr, *invv = nullptr, *sdz = nullptr; // (HB_MAX_FOLD-1) x m_pad each
    double *partial = nullptr;                              // nsplit x m_pad
    double *dsum = nullptr;                                 // m_pad: the partials added up (by the next mat-vec launch)
    double *fcorr = nullptr;                                // m_pad: k_fwd's corrections for the group after next (sentinel-prefilled like dsum)
    double *fcorr2 = nullptr;                               // m_pad: k_fold_dense's sums of a panel's own far sub-blocks (sentinel-prefilled)
    double *ddense = nullptr;                               // m_pad: the dense chain's changes of effect by marker (k_chain_dense -> k_fold_dense; sentinel-prefilled)
    double *dots = nullptr;                                 // m_pad (hb_ctx_dot)
    int nchunks = 0, nsplit = 0;
    int32_t *ev_count = nullptr, *ev_idx = nullptr;
    double *ev_delta = nullptr;
    double *acc = nullptr;        // HB_ACC_N device scalars
    double *h_acc = nullptr;      // pinned mirror
    hb_sweep_in *d_in = nullptr;  // device copy of the sweep parameters
    hb_sweep_in *h_in = nullptr;  // pinned staging

    double *Cmat = nullptr;
    int nc = 0;
    int32_t *zid = nullptr;
    int nr = 0;
    std::vector<int> nlev, lev_first;
    double *lev_buf = nullptr;
    int lev_total = 0;
    // device-resident state of the covariate / random-effect blocks (hb_ctx_blocks_*): [beta nc][estR lev_total][vrtmp nr][vr nr]
    double *blk = nullptr, *h_blk = nullptr; // device, pinned mirror
    double *blk_zz = nullptr, *blk_z = nullptr, *h_z = nullptr; // level counts, this iteration's level deviates (device, pinned staging)
    int blk_n = 0;
    std::vector<double> blk_cpc; // C_i . C_i
    double *scratch = nullptr; // small device scratch (>= 4096 doubles)
    double *ru_ws = nullptr;   // k_reduce_ru's partial sums and tickets (128 doubles, zero at rest)
    long long *dbg = nullptr;  // optional chain-kernel cycle stamps, 32 per panel
    // optional (hb_ctx_set_profiling bit 3): start / end of every block of every mat-vec launch of the last sweep on the
    // constant 100 MHz clock, [launch][HB_LSTAMP_BLOCKS][2]; lstamp_nblk[launch] = blocks the launch had (0: not launched)
    unsigned long long *lstamp = nullptr;
    std::vector<int> lstamp_nblk, lstamp_cols;

    uint32_t *wind = nullptr;
    uint8_t *wflag = nullptr;
    double *wppa = nullptr;
    int nw = 0;

    // captured sweep graph, keyed by (model_index, n_fold)
    hipGraphExec_t gexec = nullptr;
    // captured sweeps by (model, classes, geometry): switching the geometry between sweeps (hb_run's adaptive choice) replays a
    // cached graph instead of capturing again; graph_model == -1 marks all of them stale (pointers changed)
    struct graph_entry { int model, fold, pipeline, Lv, D, pb, pe; hipGraph_t g; hipGraphExec_t e; };
    int rng_pb = 0, rng_pe = 0;                 // panels of the (partial) sweep being enqueued; pe == 0: the whole sweep
    bool rng_first = true, rng_last = true;
    std::vector<graph_entry> gcache;
    hipGraph_t graph = nullptr;
    int graph_model = -1, graph_fold = -1;
    bool use_graph = true;

    // Row-sharded exact cross-check mode (hb_bayes_args.shard_rows): this context holds a block of INDIVIDUALS of every marker; the
    // digit-plane sums of each panel mat-vec (and the few other n-long reductions) are summed over the shards by this hook, on
    // host arrays, in place. Integer sums are order-independent, so every shard runs the single-GPU chain.
    int (*row_reduce)(void *user, double *vals, size_t count) = nullptr;
    void *row_user = nullptr;
    int row_rank = 0, row_world = 1;
    bool row_failed = false;
    bool profiling = false;
    bool chain_alone = false; // hb_ctx_set_profiling bit 2: the pipeline's kernels, mat-vec launches first, the chain alone afterwards
    hb_sweep_timing timing{};
    std::vector<hipEvent_t> ev_pool;

    // ---- an aborted sweep is replayed ----
    // Every wait of the persistent pipeline is bounded; a waiter that gives up raises the abort flag, all kernels of the sweep
    // leave, and fetch_acc() reports HB_ERR_ABORTED. The state a sweep changes (effects, residual, u, the posterior counters) is
    // copied aside before each sweep of a run (hb_ctx_snapshot: one kernel, a few MB) and put back by hb_ctx_restore(); the
    // per-SNP draws are counter-based, so the replayed sweep is the same chain.
    struct snap_seg { void *live; size_t off, bytes; };
    char *snap = nullptr;
    size_t snap_cap = 0;
    std::vector<snap_seg> snap_segs; // what the last snapshot holds
    bool aborted = false;            // the last fetch found the abort flag raised
    int timeout_ms = 3000;           // a device-side wait gives up after this long. 3 s for whoever drives the context directly (hb_ctx_sweep:
                                     // nothing replays an aborted sweep there); hb_run_step, which does replay, runs its sweeps with 100 ms
    bool timeout_env = false;        // HB_TIMEOUT_MS given: that value everywhere, hb_run_step does not shorten it
    bool force_geometry = false;     // hb_run_step's fall-back to the per-panel kernels: hb_ctx_set_pipeline ignores env_pinned while set
    int inject_abort_panel = -1;     // debug hook (hb_ctx_debug_inject_abort): the next sweeps are aborted once chain_done reaches this panel
    int inject_abort_times = 0;
    hipStream_t s_dbg = nullptr;
    // HB_DEBUG_ABORT=1: start / latest end / finished blocks of every mat-vec launch of the sweep ([npanels + 2][4], hb_ldiag_note)
    unsigned long long *ldiag = nullptr;
    std::vector<int> ldiag_nblk;
};

extern "C" int hb_ctx_snapshot(hb_ctx *c, int model_index, bool store, bool count_pip);
extern "C" int hb_ctx_restore(hb_ctx *c);
unsigned hbk_long_wait_flushes();
int hbk_set_timeout(hb_ctx *c);
int hbk_time_stream_read(hb_ctx *c, int reps, double *avg_ms, int64_t *bytes);
int hbk_copy_segs(hb_ctx *c, const std::vector<hb_ctx::snap_seg> &segs, bool restore);

int hb_sweep_enqueue(hb_ctx *c, const hb_sweep_in *in, bool timed);
int hb_ctx_switch_geometry(hb_ctx *c, int32_t pipeline, int32_t lookahead, int32_t dotgroup); // hb_ctx_set_pipeline without touching home_lv / home_d (hb_run's per-sweep choice)
extern "C" int hb_ctx_sweep_range(hb_ctx *c, const hb_sweep_in *in, int block, int nblocks);
extern "C" int hb_ctx_sweep_begin(hb_ctx *c, const hb_sweep_in *in);
extern "C" int hb_ctx_sweep_end(hb_ctx *c, hb_sweep_out *out);
int hb_comm_allreduce_f64(hb_comm *c, double *buf, size_t count, hipStream_t st);
int hb_build_gram_impl(hb_ctx *c);
int hb_build_gram16(hb_ctx *c);
int hb_build_gcert(hb_ctx *c);

// device buffers of one summary-level run (hb_sbayes.hip owns them; the kernels are in hb_sbayes.hpp)
struct hb_sb_dev {
    int m = 0, m_pad = 0, n = 0;
    uint64_t seed = 0;
    hipStream_t stream = nullptr;
    double *ldm = nullptr, *r_hat = nullptr, *xy = nullptr, *g = nullptr, *xpx = nullptr, *vx = nullptr, *vargL = nullptr;
    double *thr = nullptr, *invv = nullptr, *sdz = nullptr, *acc = nullptr, *ev_gi = nullptr, *wppa = nullptr;
    uint8_t *tracker = nullptr, *wflag = nullptr;
    uint32_t *nzrate = nullptr, *wind = nullptr;
    int *ev_n = nullptr, *ev_col = nullptr;
    hb_sweep_in *d_in = nullptr;
    int nw = 0;
};
int hbk_sb_enqueue_sweep(hb_sb_dev *d, int model, int n_fold);
int hbk_sb_windows(hb_sb_dev *d);
