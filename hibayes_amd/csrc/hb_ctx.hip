// hb_ctx.hip — device context of one genotype shard and the fine-grained C ABI on top of it.
#include "hb_internal.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>

// launch wrappers implemented in hb_kernels.hip
int hbk_init_attrs();
int hbk_stats(hb_ctx *c);
int hbk_dot_all(hb_ctx *c);
int hbk_dot_panels(hb_ctx *c, int reps);
int hbk_reduce_ru(hb_ctx *c);
int hbk_shift(hb_ctx *c, double a);
int hbk_to_f32(hb_ctx *c);
int hbk_cov_dot(hb_ctx *c, int i, double *dev_out);
int hbk_cov_axpy(hb_ctx *c, int i, double a);
int hbk_level_sums(hb_ctx *c, int term, double *dev_sums, int nlev);
int hbk_level_axpy(hb_ctx *c, int term, const double *dev_delta);
int hbk_windows(hb_ctx *c);
int hbk_f64_to_i8(hb_ctx *c, const double *dsrc, int64_t lds, int ncols, int8_t *dst, int *dbad);
int hbk_bed_decode(hb_ctx *c, const uint8_t *dbed, int64_t bpc, int nind, const int32_t *drows, int col0, int ncols);
int hbk_generate(hb_ctx *c, uint64_t seed, int mono_every);
int hbk_xalpha(hb_ctx *c, const double *dev_alpha, double *dev_out);
int hbk_time_matvec(hb_ctx *c, int D, int reps, int as_pipeline, double *avg_us, int *launches);
int hbk_probe_concurrency(hb_ctx *c, int *concurrent);
int hbk_cov_step(hb_ctx *c, int i, double v, double vare, double z, double *beta_i);
int hbk_lev_step(hb_ctx *c, int term, int q0, int qr, const double *zz, double *estR, const double *z, double vare, double *vrtmp,
                 double *vr, double s2r_dfr, double chis);
int hbk_xmat(hb_ctx *c, const int *didx, const double *dval, int nnz, double *dout);
int hbk_pack2(hb_ctx *c);
int hbk_unpack2(hb_ctx *c, int col0, int ncols, int8_t *dst);

static thread_local std::string g_err;

void hb_set_error(const std::string &msg) { g_err = msg; }
int hb_fail(int status, const std::string &msg)
{
    g_err = msg;
    return status;
}

template <typename T>
static int dev_alloc(T **p, size_t count, bool zero = true)
{
    HB_HIP(hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T)));
    if (zero) HB_HIP(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(T)));
    return HB_OK;
}

// The few MB of words that one workgroup publishes write-through while workgroups on OTHER XCDs poll them (dots, correction sums,
// changes of effect, move counts and lists, bounds, the flag block). HB_HANDOFF_ALLOC selects what kind of device memory they
// live in: 0 = plain hipMalloc (coarse-grained: cached in every XCD's L2 — the polls and stores carry sc1 to get past it),
// 1 = hipDeviceMallocUncached (MTYPE UC: no L2 copy anywhere, a store goes straight to the memory side), 2 = fine-grained
// (coherent across agents). Round 5's experiment on the lost store of DESIGN.md §9.0; the arrays are small, the genotypes, Gram
// blocks and everything streamed stay plain.
static int g_handoff_kind = -1;
template <typename T>
static int dev_alloc_handoff(T **p, size_t count)
{
    if (g_handoff_kind < 0) {
        const char *e = getenv("HB_HANDOFF_ALLOC");
        g_handoff_kind = e ? std::max(0, std::min(2, atoi(e))) : 0;
    }
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (g_handoff_kind == 0) HB_HIP(hipMalloc(reinterpret_cast<void **>(p), bytes));
    else HB_HIP(hipExtMallocWithFlags(reinterpret_cast<void **>(p), bytes, g_handoff_kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    HB_HIP(hipMemset(*p, 0, bytes));
    return HB_OK;
}

// normalise (pipeline, Lv, D) and derive the Gram band / version ring sizes
static void hb_pipeline_geometry(hb_ctx *c)
{
    if (!c->concurrent) c->pipeline = 0; // the persistent pipeline needs co-resident kernels (hbk_probe_concurrency)
    c->Lv = std::max(0, std::min(6, c->Lv));
    c->D = c->pipeline ? std::max(1, std::min(8, c->D)) : 1;
    // Lv counts mat-vec GROUPS of look-ahead; the Gram band then spans (Lv + 1) * D - 1 earlier panels
    // (20 = HB_LBMAX in hb_kernels.hip, what k_chain_persist folds; 27 with k_fwd beside the group chain — three groups of seven
    // panels of look-ahead — which the sweep refuses for the models that run k_chain_persist)
    const int lbmax = (c->fwd_group && c->P == 512 && ((c->Lv == 3 && c->D == 7) || (c->Lv == 2 && c->D == 8))) ? 27 : 20; // ((2, 8): 23, round 6)
    while ((c->Lv + 1) * c->D - 1 > lbmax) {
        if (c->Lv > 1) c->Lv--; else c->D--;
    }
    c->L = std::max((c->Lv + 1) * c->D - 1, c->Lv);
    c->NB = c->Lv + 1;
}

extern "C" {

int hb_abi_version(void) { return HB_ABI_VERSION; }
const char *hb_version(void) { return "hibayes_amd 0.1 (gfx950)"; }
const char *hb_last_error(void) { return g_err.c_str(); }

int hb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

size_t hb_exchange_count(int32_t n) { return (size_t)n + 16; }

static int auto_panel(int m)
{
    if (m >= 4096) return 512;
    if (m >= 1024) return 256;
    if (m >= 256) return 128;
    return 64;
}

// development aid (HB_DEBUG_SEGV=1): a native backtrace on SIGSEGV — there is no debugger in the image
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void hb_segv_handler(int sig)
{
    void *bt[64];
    const int n = backtrace(bt, 64);
    const char msg[] = "hibayes_gpu: fatal signal, native backtrace:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int hb_ctx_create(const hb_ctx_params *p, hb_ctx **out)
{
    if (getenv("HB_DEBUG_SEGV")) signal(SIGSEGV, hb_segv_handler);
    if (!p || !out) return hb_fail(HB_ERR_INVALID, "hb_ctx_create: null argument");
    *out = nullptr;
    if (p->n < 2 || p->m < 1) return hb_fail(HB_ERR_INVALID, "hb_ctx_create: n >= 2 and m >= 1 required");
    const int ndev = hb_device_count();
    if (ndev <= 0) return hb_fail(HB_ERR_NO_DEVICE, "no HIP device available: the hibayes GPU engine has no CPU fallback");
    if (p->device < 0 || p->device >= ndev) return hb_fail(HB_ERR_INVALID, "hb_ctx_create: bad device ordinal");
    int P = p->panel ? p->panel : auto_panel(p->m);
    if (P != 64 && P != 128 && P != 256 && P != 512)
        return hb_fail(HB_ERR_INVALID, "hb_ctx_create: panel must be 64, 128, 256 or 512");
    HB_HIP(hipSetDevice(p->device));
    hb_ctx *c = new hb_ctx();
    c->device = p->device;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) == hipSuccess && ncu > 0) c->num_cus = ncu;
    }
    c->n = p->n;
    c->m = p->m;
    c->P = P;
    c->npanels = (p->m + P - 1) / P;
    c->m_pad = c->npanels * P;
    c->ld = ((int64_t)p->n + 255) / 256 * 256;
    c->precise = p->precise;
    c->m_offset = p->m_offset;
    c->seed = p->seed;
    c->nchunks = (int)((c->ld + 4095) / 4096);
    c->nsplit = c->nchunks;
    // pipeline geometry; hb_ctx_set_pipeline() changes it later
    c->pipeline = 1;
    c->Lv = 2;
    c->D = 7;
    if (const char *e = getenv("HB_PIPELINE")) c->pipeline = atoi(e) ? 1 : 0;
    if (const char *e = getenv("HB_LOOKAHEAD")) c->Lv = atoi(e);
    if (const char *e = getenv("HB_DOTGROUP")) c->D = atoi(e);
    if (const char *e = getenv("HB_GRAPH")) c->use_graph = atoi(e) != 0;
    if (const char *e = getenv("HB_DOTQ2_CPL")) c->dotq2_cpl = atoi(e) == 1 ? 1 : 2;
    if (const char *e = getenv("HB_DOTQ2_TILES")) c->dotq2_tiles = std::max(1, atoi(e));
    if (const char *e = getenv("HB_DOTQ2_KIND")) c->dotq2_kind = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("HB_GRAM16")) c->gram16_on = atoi(e) != 0;
    if (const char *e = getenv("HB_CERT")) c->gcert_on = atoi(e) != 0;
    if (const char *e = getenv("HB_Q2M_CT")) c->q2m_ct = atoi(e) >= 16 ? 16 : atoi(e) >= 8 ? 8 : 4;
    if (const char *e = getenv("HB_Q2M_G")) c->q2m_g = atoi(e) == 3 ? 3 : atoi(e) >= 2 ? 2 : atoi(e) == 0 ? 0 : 1;
    if (const char *e = getenv("HB_Q2M_SC")) c->q2m_sc = atoi(e) != 0;
    if (const char *e = getenv("HB_DOTQ2_NC")) c->dotq2_nc = std::max(4, atoi(e) / 4 * 4);
    if (const char *e = getenv("HB_DOTQ2_RS")) c->dotq2_rs = atoi(e) == 256 ? 256 : atoi(e) == 128 ? 128 : 512;
    if (const char *e = getenv("HB_OVERLAP")) c->overlap = atoi(e) != 0;
    if (const char *e = getenv("HB_MVP")) c->mvp = atoi(e) != 0;
    if (const char *e = getenv("HB_CHAIN")) c->chain_kind = std::strcmp(e, "panel") == 0 ? 0 : std::strcmp(e, "all") == 0 ? 3 : (atoi(e) ? atoi(e) : 1);
    if (const char *e = getenv("HB_WARM_GROUP")) c->warm_group = atoi(e) != 0;
    if (const char *e = getenv("HB_FWD")) c->fwd_group = atoi(e) != 0;
    if (const char *e = getenv("HB_DENSE")) c->dense_chain = atoi(e) != 0;
    if (const char *e = getenv("HB_TIMEOUT_MS")) { c->timeout_ms = std::max(1, std::min(60000, atoi(e))); c->timeout_env = true; }
    if (const char *e = getenv("HB_KAPPA")) c->kappa = atof(e);
    if (const char *e = getenv("HB_DOT_LDS")) c->dot_lds = std::min(65536, std::max(0, atoi(e)));
    if (const char *e = getenv("HB_CANDF")) c->candf = std::min(1.0, std::max(0.0, atof(e)));
    c->env_pinned = getenv("HB_PIPELINE") || getenv("HB_LOOKAHEAD") || getenv("HB_DOTGROUP");
    if (!c->pipeline && !getenv("HB_LOOKAHEAD")) c->Lv = 0;
    hb_pipeline_geometry(c);
    int rc = HB_OK;
#define TRY(x)                   \
    do {                         \
        rc = (x);                \
        if (rc) {                \
            hb_ctx_destroy(c);   \
            return rc;           \
        }                        \
    } while (0)
    {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return hb_fail(HB_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        }
        // HB_STREAM_PRIO=1 (an A/B for the dense stall): the streams of the persistent kernels at the highest
        // priority, so that they can never share a hardware queue with the stream of the mat-vec launches
        int plo = 0, phi = 0;
        const bool prio = getenv("HB_STREAM_PRIO") && atoi(getenv("HB_STREAM_PRIO")) != 0 && hipDeviceGetStreamPriorityRange(&plo, &phi) == hipSuccess;
        auto mk = [&](hipStream_t *st) { return prio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, phi) : hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
        if (mk(&c->s_chain) != hipSuccess ||
            mk(&c->s_upd) != hipSuccess ||
            mk(&c->s_warm) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) {
            hb_ctx_destroy(c);
            return hb_fail(HB_ERR_HIP, "hipStreamCreate failed");
        }
        c->ev_dot.resize(c->npanels);
        c->ev_chain.resize(c->npanels);
        c->ev_upd.resize(c->npanels);
        for (int p = 0; p < c->npanels; p++)
            if (hipEventCreateWithFlags(&c->ev_dot[p], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_chain[p], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_upd[p], hipEventDisableTiming) != hipSuccess) {
                hb_ctx_destroy(c);
                return hb_fail(HB_ERR_HIP, "hipEventCreate failed");
            }
    }
    TRY(hbk_init_attrs());
    const size_t mp = (size_t)c->m_pad;
    TRY(dev_alloc(&c->X, (size_t)c->ld * mp));
    TRY(dev_alloc(&c->xpx, mp));
    TRY(dev_alloc(&c->vx, mp));
    TRY(dev_alloc(&c->s1, mp));
    TRY(dev_alloc(&c->g, mp));
    TRY(dev_alloc(&c->vargL, mp));
    TRY(dev_alloc(&c->alpha_sum, mp));
    TRY(dev_alloc(&c->alpha_sq, mp));
    TRY(dev_alloc(&c->tracker, mp));
    TRY(dev_alloc(&c->nzrate, mp));
    TRY(dev_alloc(&c->r, (size_t)c->ld * 8));          // up to 8 versions; slot 0 is the residual between sweeps
    TRY(dev_alloc(&c->u, (size_t)c->ld));
    TRY(dev_alloc(&c->r32, (size_t)c->ld * 8));
    TRY(dev_alloc(&c->rq, (size_t)c->ld * 8 * HB_ND));
    TRY(dev_alloc(&c->vexp, 8));
    TRY(dev_alloc(&c->gexp, (size_t)c->npanels + 1));
    TRY(dev_alloc_handoff(&c->mb, ((size_t)c->npanels + 2) * HB_MBS));
    TRY(dev_alloc(&c->accq, (size_t)HB_ND * mp));
    TRY(dev_alloc(&c->xinfo, 2));
    TRY(dev_alloc(&c->thr, mp * (HB_MAX_FOLD - 1)));
    TRY(dev_alloc(&c->invv, mp * (HB_MAX_FOLD - 1)));
    TRY(dev_alloc(&c->sdz, mp * (HB_MAX_FOLD - 1)));
    TRY(dev_alloc(&c->partial, mp * (size_t)c->nsplit));
    TRY(dev_alloc_handoff(&c->dsum, mp));
    TRY(dev_alloc_handoff(&c->fcorr, mp));
    TRY(dev_alloc_handoff(&c->ddense, mp));
    TRY(dev_alloc_handoff(&c->fcorr2, mp));
    TRY(dev_alloc(&c->dots, mp));
    TRY(dev_alloc_handoff(&c->ev_count, (size_t)c->npanels * HB_EVS));
    TRY(dev_alloc_handoff(&c->ev_idx, mp));
    TRY(dev_alloc_handoff(&c->ev_delta, mp));
    TRY(dev_alloc(&c->acc, HB_ACC_N));
    TRY(dev_alloc(&c->d_in, 1));
    TRY(dev_alloc(&c->scratch, 8192));
    TRY(dev_alloc_handoff(&c->flags, (size_t)4096));
    TRY(dev_alloc(&c->hot_slot, mp));
    TRY(dev_alloc(&c->hot_list, (size_t)(c->npanels + 1) * 256));
    TRY(dev_alloc(&c->thr0f, mp + 1024));
    TRY(dev_alloc(&c->opn, mp + 1024));
    TRY(dev_alloc(&c->ru_ws, (size_t)128));
    TRY(dev_alloc(&c->hot_n, (size_t)c->npanels));
    if (getenv("HB_DEBUG_ABORT")) {
        TRY(dev_alloc(&c->ldiag, ((size_t)c->npanels + 2) * 4));
        c->ldiag_nblk.assign((size_t)c->npanels + 2, 0);
    }
    {
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&c->h_acc), sizeof(double) * HB_ACC_N);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_in), sizeof(hb_sweep_in));
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&c->h_flags), 256);
        if (e != hipSuccess) {
            hb_ctx_destroy(c);
            return hb_fail(HB_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        }
    }
#undef TRY
    // the allocation memsets ran on the null stream, which the context's non-blocking streams do not wait for
    HB_HIP(hipDeviceSynchronize());
    {   // can the persistent pipeline run here at all?
        const char *why = nullptr;
        if (getenv("AMD_SERIALIZE_KERNEL") && atoi(getenv("AMD_SERIALIZE_KERNEL")) != 0) why = "AMD_SERIALIZE_KERNEL is set";
        else if (getenv("HIP_LAUNCH_BLOCKING") && atoi(getenv("HIP_LAUNCH_BLOCKING")) != 0) why = "HIP_LAUNCH_BLOCKING is set";
        else if (getenv("HB_FORCE_SERIAL") && atoi(getenv("HB_FORCE_SERIAL")) != 0) why = "HB_FORCE_SERIAL is set";
        else {
            int conc = 1;
            const int prc = hbk_probe_concurrency(c, &conc);
            if (prc) {
                hb_ctx_destroy(c);
                return prc;
            }
            if (!conc) why = "kernels on two streams do not run concurrently here (serialising profiler or shared GPU)";
        }
        if (why) {
            c->concurrent = false;
            c->pipeline_note = why;
            if (!c->env_pinned || c->pipeline) c->Lv = std::min(c->Lv, 2);
            hb_pipeline_geometry(c);
            fprintf(stderr, "hibayes_gpu: %s: the sweep uses the event-ordered per-panel kernels instead of the persistent "
                            "pipeline (same chain, slower)\n", why);
        }
    }
    *out = c;
    return HB_OK;
}

static void blocks_free(hb_ctx *c);

void hb_ctx_destroy(hb_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &ge : c->gcache) {
        if (ge.e) (void)hipGraphExecDestroy(ge.e);
        if (ge.g) (void)hipGraphDestroy(ge.g);
    }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto e : c->ev_dot) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_chain) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_upd) if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->s_chain) (void)hipStreamDestroy(c->s_chain);
    if (c->s_upd) (void)hipStreamDestroy(c->s_upd);
    if (c->s_warm) (void)hipStreamDestroy(c->s_warm);
    if (c->s_dbg) (void)hipStreamDestroy(c->s_dbg);
    for (auto e : c->ev_ot) if (e) (void)hipEventDestroy(e);
    for (auto e : c->ev_ou) if (e) (void)hipEventDestroy(e);
    if (c->s_t2) (void)hipStreamDestroy(c->s_t2);
    if (c->s_uk) (void)hipStreamDestroy(c->s_uk);
    if (c->s_fk) (void)hipStreamDestroy(c->s_fk);
    void *ptrs[] = {c->X, c->X2, c->xpx, c->vx, c->s1, c->g, c->vargL, c->alpha_sum, c->alpha_sq, c->tracker, c->nzrate, c->r, c->u,
                    c->r32, c->rq, c->vexp, c->gexp, c->mb, c->accq, c->gram, c->xinfo, c->thr, c->invv, c->sdz, c->partial, c->dsum, c->fcorr, c->ddense, c->fcorr2, c->dots, c->ev_count, c->ev_idx,
                    c->ev_delta, c->acc, c->d_in, c->scratch, c->dbg, c->lstamp, c->flags, c->gram16, c->ga, c->gB, c->gcmax, c->g16_flag, c->hot_slot, c->hot_list, c->hot_n, c->thr0f, c->opn, c->ru_ws, c->mvp_ho, c->Cmat, c->zid, c->lev_buf, c->wind, c->wflag, c->wppa, c->snap, c->ldiag};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    blocks_free(c);
    if (c->h_acc) (void)hipHostFree(c->h_acc);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int hb_ctx_panel(const hb_ctx *c) { return c ? c->P : 0; }
int64_t hb_ctx_ld(const hb_ctx *c) { return c ? c->ld : 0; }

static int check_cols(hb_ctx *c, int col0, int ncols, const char *who)
{
    if (!c) return hb_fail(HB_ERR_INVALID, std::string(who) + ": null context");
    if (col0 < 0 || ncols < 0 || (int64_t)col0 + ncols > c->m)
        return hb_fail(HB_ERR_INVALID, std::string(who) + ": column range outside the shard");
    HB_HIP(hipSetDevice(c->device));
    return HB_OK;
}

static void invalidate(hb_ctx *c)
{
    c->gram_ready = false;
    c->stats_ready = false;
    if (c->layout == 2) { // new genotypes: the packed copy is stale
        c->layout = 8;
        c->graph_model = -1;
    }
}

static int need_int8(hb_ctx *c, const char *who)
{
    if (!c->X) return hb_fail(HB_ERR_INVALID, std::string(who) + ": the context holds its genotypes in the 2-bit layout only (hb_ctx_set_layout(c, 8, 1) unpacks them)");
    return HB_OK;
}

int hb_ctx_upload_genotype_i8(hb_ctx *c, const int8_t *X, int64_t ld, int32_t col0, int32_t ncols)
{
    int rc = check_cols(c, col0, ncols, "hb_ctx_upload_genotype_i8");
    if (rc) return rc;
    rc = need_int8(c, "hb_ctx_upload_genotype_i8");
    if (rc) return rc;
    if (!X || ld < c->n) return hb_fail(HB_ERR_INVALID, "hb_ctx_upload_genotype_i8: bad source");
    HB_HIP(hipMemcpy2DAsync(c->X + (int64_t)col0 * c->ld, (size_t)c->ld, X, (size_t)ld, (size_t)c->n, (size_t)ncols,
                            hipMemcpyHostToDevice, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    invalidate(c);
    return HB_OK;
}

int hb_ctx_upload_genotype_f64(hb_ctx *c, const double *X, int64_t ld, int32_t col0, int32_t ncols)
{
    int rc = check_cols(c, col0, ncols, "hb_ctx_upload_genotype_f64");
    if (rc) return rc;
    rc = need_int8(c, "hb_ctx_upload_genotype_f64");
    if (rc) return rc;
    if (!X || ld < c->n) return hb_fail(HB_ERR_INVALID, "hb_ctx_upload_genotype_f64: bad source");
    const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(ncols, (int64_t)(64 << 20) / std::max(1, c->n)));
    double *stage = nullptr;
    int *dbad = nullptr;
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&stage), (size_t)chunk * c->n * sizeof(double)));
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&dbad), sizeof(int)));
    HB_HIP(hipMemsetAsync(dbad, 0, sizeof(int), c->stream));
    for (int c0 = 0; c0 < ncols && rc == HB_OK; c0 += chunk) {
        const int nc = std::min(chunk, ncols - c0);
        hipError_t e = hipMemcpy2DAsync(stage, (size_t)c->n * sizeof(double), X + (int64_t)(col0 + c0) * ld,
                                        (size_t)ld * sizeof(double), (size_t)c->n * sizeof(double), (size_t)nc,
                                        hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = hb_fail(HB_ERR_HIP, hipGetErrorString(e)); break; }
        rc = hbk_f64_to_i8(c, stage, c->n, nc, c->X + (int64_t)(col0 + c0) * c->ld, dbad);
        if (rc == HB_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = hb_fail(HB_ERR_HIP, "sync failed");
    }
    int bad = 0;
    if (rc == HB_OK && hipMemcpy(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
        rc = hb_fail(HB_ERR_HIP, "hipMemcpy failed");
    (void)hipFree(stage);
    (void)hipFree(dbad);
    invalidate(c);
    if (rc) return rc;
    if (bad)
        return hb_fail(HB_ERR_UNSUPPORTED,
                       "genotype matrix holds non-integer or out-of-range values: the int8 GPU path needs integer "
                       "codes in [-127, 127] (imputed fractional genotypes are not supported)");
    return HB_OK;
}

int hb_ctx_upload_bed(hb_ctx *c, const uint8_t *bed, int64_t nbytes, int32_t nind, const int32_t *rows, int32_t col0,
                      int32_t ncols)
{
    int rc = check_cols(c, col0, ncols, "hb_ctx_upload_bed");
    if (rc) return rc;
    rc = need_int8(c, "hb_ctx_upload_bed");
    if (rc) return rc;
    const int64_t bpc = ((int64_t)nind + 3) / 4;
    if (!bed || nbytes < 3 + bpc * ((int64_t)col0 + ncols)) return hb_fail(HB_ERR_INVALID, "hb_ctx_upload_bed: file image too short");
    if (bed[0] != 0x6c || bed[1] != 0x1b || bed[2] != 0x01)
        return hb_fail(HB_ERR_INVALID, "hb_ctx_upload_bed: not a SNP-major PLINK .bed (magic 6c 1b 01)");
    if (!rows && nind < c->n) return hb_fail(HB_ERR_INVALID, "hb_ctx_upload_bed: fewer individuals than n");
    if (rows)
        for (int i = 0; i < c->n; i++)
            if (rows[i] < 0 || rows[i] >= nind) return hb_fail(HB_ERR_INVALID, "hb_ctx_upload_bed: row index out of range");
    uint8_t *dbed = nullptr;
    int32_t *drows = nullptr;
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&dbed), (size_t)(bpc * ncols)));
    HB_HIP(hipMemcpy(dbed, bed + 3 + bpc * col0, (size_t)(bpc * ncols), hipMemcpyHostToDevice));
    if (rows) {
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&drows), sizeof(int32_t) * c->n));
        HB_HIP(hipMemcpy(drows, rows, sizeof(int32_t) * c->n, hipMemcpyHostToDevice));
    }
    rc = hbk_bed_decode(c, dbed, bpc, nind, drows, col0, ncols);
    if (rc == HB_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = hb_fail(HB_ERR_HIP, "sync failed");
    (void)hipFree(dbed);
    if (drows) (void)hipFree(drows);
    invalidate(c);
    return rc;
}

int hb_ctx_generate_genotype(hb_ctx *c, uint64_t seed, int32_t mono_every)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_generate_genotype");
    if (rc) return rc;
    rc = need_int8(c, "hb_ctx_generate_genotype");
    if (rc) return rc;
    rc = hbk_generate(c, seed, mono_every);
    if (rc) return rc;
    HB_HIP(hipStreamSynchronize(c->stream));
    invalidate(c);
    return HB_OK;
}

int hb_ctx_download_genotype(hb_ctx *c, int8_t *X, int64_t ld, int32_t col0, int32_t ncols)
{
    int rc = check_cols(c, col0, ncols, "hb_ctx_download_genotype");
    if (rc) return rc;
    if (!c->X) { // 2-bit only: unpack a slab at a time into a scratch buffer
        const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(ncols, (int64_t)(256 << 20) / c->ld));
        int8_t *tmp = nullptr;
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&tmp), (size_t)chunk * c->ld));
        hipError_t e = hipSuccess;
        for (int c0 = 0; c0 < ncols && e == hipSuccess && rc == HB_OK; c0 += chunk) {
            const int nc = std::min(chunk, ncols - c0);
            rc = hbk_unpack2(c, col0 + c0, nc, tmp);
            if (rc == HB_OK) e = hipStreamSynchronize(c->stream);
            if (rc == HB_OK && e == hipSuccess)
                e = hipMemcpy2D(X + (int64_t)c0 * ld, (size_t)ld, tmp, (size_t)c->ld, (size_t)c->n, (size_t)nc, hipMemcpyDeviceToHost);
        }
        (void)hipFree(tmp);
        if (rc) return rc;
        if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("hb_ctx_download_genotype: ") + hipGetErrorString(e));
        return HB_OK;
    }
    HB_HIP(hipMemcpy2D(X, (size_t)ld, c->X + (int64_t)col0 * c->ld, (size_t)c->ld, (size_t)c->n, (size_t)ncols,
                       hipMemcpyDeviceToHost));
    return HB_OK;
}

// Armadillo's arrayops::accumulate order (two interleaved accumulators) so that sumvx agrees
// with sum(vx) of reference src/Bayes.cpp:316 to the last bit when vx does
static double arma_sum(const double *v, size_t n)
{
    double a1 = 0.0, a2 = 0.0;
    size_t j;
    for (j = 1; j < n; j += 2) {
        a1 += v[j - 1];
        a2 += v[j];
    }
    if ((j - 1) < n) a1 += v[j - 1];
    return a1 + a2;
}

int hb_ctx_marker_stats(hb_ctx *c, double *xpx, double *vx, double *sumvx, int32_t *nvar0)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_marker_stats");
    if (rc) return rc;
    if (c->X || !c->stats_ready) { // (2-bit only: xpx / vx were computed before the int8 copy was dropped and stay valid)
        rc = need_int8(c, "hb_ctx_marker_stats");
        if (rc) return rc;
        rc = hbk_stats(c);
        if (rc) return rc;
    }
    std::vector<double> hv(c->m);
    HB_HIP(hipMemcpy(hv.data(), c->vx, sizeof(double) * c->m, hipMemcpyDeviceToHost));
    if (vx) std::memcpy(vx, hv.data(), sizeof(double) * c->m);
    if (xpx) HB_HIP(hipMemcpy(xpx, c->xpx, sizeof(double) * c->m, hipMemcpyDeviceToHost));
    if (sumvx) *sumvx = arma_sum(hv.data(), hv.size());
    if (nvar0) {
        int z = 0;
        for (double v : hv) z += (v == 0.0);
        *nvar0 = z;
    }
    return HB_OK;
}

int hb_ctx_set_layout(hb_ctx *c, int32_t bits, int32_t keep_int8)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_set_layout");
    if (rc) return rc;
    if (bits == 0) bits = 8;
    if (bits != 8 && bits != 2) return hb_fail(HB_ERR_INVALID, "hb_ctx_set_layout: bits must be 8 or 2");
    HB_HIP(hipStreamSynchronize(c->stream));
    if (bits == 8) {
        if (!c->X) { // unpack the dropped int8 copy
            HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->X), (size_t)c->ld * c->m_pad));
            HB_HIP(hipMemsetAsync(c->X, 0, (size_t)c->ld * c->m_pad, c->stream));
            rc = hbk_unpack2(c, 0, c->m_pad, c->X);
            if (rc) return rc;
            HB_HIP(hipStreamSynchronize(c->stream));
        }
        if (c->layout != 8) c->graph_model = -1;
        c->layout = 8;
        return HB_OK;
    }
    if (c->precise != 2)
        return hb_fail(HB_ERR_UNSUPPORTED, "the 2-bit resident layout needs the fixed-point mat-vec (precise = 2)");
    if (c->layout != 2) {
        rc = need_int8(c, "hb_ctx_set_layout");
        if (rc) return rc;
        if (!c->stats_ready) {
            rc = hbk_stats(c);
            if (rc) return rc;
        }
        if (c->xmin < 0 || c->xmax > 3)
            return hb_fail(HB_ERR_UNSUPPORTED, "the 2-bit resident layout needs genotype codes 0..3 (this matrix holds " + std::to_string(c->xmin) +
                                                   ".." + std::to_string(c->xmax) + "): keep the int8 layout");
        c->ld2 = (c->ld + 511) / 512 * 128;
        if (!c->X2) HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->X2), (size_t)c->ld2 * c->m_pad));
        rc = hbk_pack2(c);
        if (rc) return rc;
        HB_HIP(hipStreamSynchronize(c->stream));
        c->layout = 2;
        c->graph_model = -1;
    }
    if (!keep_int8 && c->X) {
        (void)hipFree(c->X);
        c->X = nullptr;
        c->graph_model = -1;
    }
    return HB_OK;
}

int hb_ctx_get_layout(const hb_ctx *c, int32_t *bits, int32_t *int8_resident)
{
    if (!c) return hb_fail(HB_ERR_INVALID, "hb_ctx_get_layout: null context");
    if (bits) *bits = c->layout;
    if (int8_resident) *int8_resident = c->X != nullptr;
    return HB_OK;
}

int hb_ctx_set_pipeline(hb_ctx *c, int32_t pipeline, int32_t lookahead, int32_t dotgroup)
{
    int rc = hb_ctx_switch_geometry(c, pipeline, lookahead, dotgroup);
    if (!rc) { c->home_lv = c->Lv; c->home_d = c->D; } // (what an adaptive run returns to: hb_run.hip)
    return rc;
}

} // extern "C"
int hb_ctx_switch_geometry(hb_ctx *c, int32_t pipeline, int32_t lookahead, int32_t dotgroup)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_set_pipeline");
    if (rc) return rc;
    if (c->env_pinned && c->concurrent && !c->force_geometry) return HB_OK; // HB_PIPELINE / HB_LOOKAHEAD / HB_DOTGROUP in the environment win (tuning runs)
    const int op = c->pipeline, ol = c->Lv, od = c->D;
    c->pipeline = pipeline ? 1 : 0;
    c->Lv = lookahead;
    c->D = dotgroup;
    hb_pipeline_geometry(c);
    (void)op; (void)ol; (void)od;
    // the stored band serves every geometry whose band fits into it (the captured sweeps are cached per geometry)
    if (c->L > c->Lg) c->gram_ready = false;
    return HB_OK;
}
extern "C" {

int hb_ctx_set_adaptive(hb_ctx *c, int32_t on)
{
    if (!c) return hb_fail(HB_ERR_INVALID, "hb_ctx_set_adaptive: null context");
    c->adaptive = on != 0;
    return HB_OK;
}

int hb_ctx_build_gram(hb_ctx *c, double *seconds)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_build_gram");
    if (rc) return rc;
    // (a context that holds only the 2-bit layout: hb_build_gram_impl unpacks a window of panels at a time into a scratch buffer)
    c->Lg = c->L; // the band this build stores
    c->graph_model = -1; // the captured sweeps hold the old band's stride (and, after a re-allocation, its pointer)
    const size_t need = (size_t)c->m_pad * (size_t)c->P * (size_t)(c->Lg + 1);
    if (need > c->gram_cap) {
        if (c->gram) { (void)hipFree(c->gram); c->gram = nullptr; }
        c->gram_cap = 0;
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->gram), need * sizeof(int32_t)));
        c->gram_cap = need;
    }
    HB_HIP(hipMemsetAsync(c->gram, 0, need * sizeof(int32_t), c->stream));
    if (!c->stats_ready) {
        rc = hbk_stats(c);
        if (rc) return rc;
    }
    const double amax = std::max(std::abs((double)c->xmin), std::abs((double)c->xmax));
    if (amax * amax * (double)c->n >= 2147483647.0)
        return hb_fail(HB_ERR_UNSUPPORTED, "genotype codes too large for the exact int32 Gram matrix at this n");
    const auto t0 = std::chrono::steady_clock::now();
    rc = hb_build_gram_impl(c);
    if (rc) return rc;
    rc = hb_build_gcert(c); // (the rank-one part of the band and the bound on the rest: the group chain's certificate)
    if (rc) return rc;
    rc = hb_build_gram16(c); // (the compact copy the group chain folds from; the row-sharded mode sums int32 blocks over the ranks and never runs it)
    if (rc) return rc;
    HB_HIP(hipStreamSynchronize(c->stream));
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->gram_ready = true;
    return HB_OK;
}

int hb_ctx_download_gram(hb_ctx *c, int32_t panel_index, int32_t *G)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_download_gram");
    if (rc) return rc;
    if (!c->gram_ready) return hb_fail(HB_ERR_INVALID, "hb_ctx_download_gram: call hb_ctx_build_gram first");
    if (panel_index < 0 || panel_index >= c->npanels) return hb_fail(HB_ERR_INVALID, "hb_ctx_download_gram: bad panel");
    HB_HIP(hipMemcpy(G, c->gram + (size_t)panel_index * (c->Lg + 1) * c->P * c->P, sizeof(int32_t) * c->P * c->P, hipMemcpyDeviceToHost));
    return HB_OK;
}

int hb_ctx_download_gram_band(hb_ctx *c, int32_t panel_index, int32_t l, int32_t *G)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_download_gram_band");
    if (rc) return rc;
    if (!c->gram_ready) return hb_fail(HB_ERR_INVALID, "hb_ctx_download_gram_band: call hb_ctx_build_gram first");
    if (panel_index < 0 || panel_index >= c->npanels || l < 0 || l > c->Lg)
        return hb_fail(HB_ERR_INVALID, "hb_ctx_download_gram_band: bad panel or band index");
    HB_HIP(hipStreamSynchronize(c->stream));
    HB_HIP(hipMemcpy(G, c->gram + ((size_t)panel_index * (c->Lg + 1) + l) * c->P * c->P, sizeof(int32_t) * c->P * c->P,
                     hipMemcpyDeviceToHost));
    return HB_OK;
}

const char *hb_ctx_pipeline_note(const hb_ctx *c) { return (c && !c->concurrent) ? c->pipeline_note.c_str() : nullptr; }

int hb_ctx_get_pipeline(const hb_ctx *c, int32_t *pipeline, int32_t *lookahead, int32_t *dotgroup, int32_t *band)
{
    if (!c) return hb_fail(HB_ERR_INVALID, "hb_ctx_get_pipeline: null context");
    if (pipeline) *pipeline = c->pipeline;
    if (lookahead) *lookahead = c->Lv;
    if (dotgroup) *dotgroup = c->D;
    if (band) *band = c->L;
    return HB_OK;
}

int hb_ctx_get_events(hb_ctx *c, int32_t *ev_count, int32_t *ev_idx, double *ev_delta)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_get_events");
    if (rc) return rc;
    HB_HIP(hipStreamSynchronize(c->stream));
    if (ev_count) HB_HIP(hipMemcpy2D(ev_count, sizeof(int32_t), c->ev_count, sizeof(int32_t) * HB_EVS, sizeof(int32_t), (size_t)c->npanels, hipMemcpyDeviceToHost));
    if (ev_idx) HB_HIP(hipMemcpy(ev_idx, c->ev_idx, sizeof(int32_t) * (size_t)c->m_pad, hipMemcpyDeviceToHost));
    if (ev_delta) HB_HIP(hipMemcpy(ev_delta, c->ev_delta, sizeof(double) * (size_t)c->m_pad, hipMemcpyDeviceToHost));
    return HB_OK;
}

int hb_ctx_set_residual(hb_ctx *c, const double *yadj, const double *u)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_set_residual");
    if (rc) return rc;
    if (yadj) {
        HB_HIP(hipMemcpyAsync(c->r, yadj, sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
        rc = hbk_to_f32(c);
        if (rc) return rc;
    }
    if (u) HB_HIP(hipMemcpyAsync(c->u, u, sizeof(double) * c->n, hipMemcpyHostToDevice, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    return HB_OK;
}

int hb_ctx_get_residual(hb_ctx *c, double *yadj, double *u)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_get_residual");
    if (rc) return rc;
    HB_HIP(hipStreamSynchronize(c->stream));
    if (yadj) HB_HIP(hipMemcpy(yadj, c->r, sizeof(double) * c->n, hipMemcpyDeviceToHost));
    if (u) HB_HIP(hipMemcpy(u, c->u, sizeof(double) * c->n, hipMemcpyDeviceToHost));
    return HB_OK;
}

int hb_ctx_set_effects(hb_ctx *c, const double *g, const uint8_t *tracker, const double *vargL)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_set_effects");
    if (rc) return rc;
    if (g) HB_HIP(hipMemcpy(c->g, g, sizeof(double) * c->m, hipMemcpyHostToDevice));
    if (tracker) HB_HIP(hipMemcpy(c->tracker, tracker, c->m, hipMemcpyHostToDevice));
    if (vargL) HB_HIP(hipMemcpy(c->vargL, vargL, sizeof(double) * c->m, hipMemcpyHostToDevice));
    return HB_OK;
}

int hb_ctx_get_effects(hb_ctx *c, double *g, uint8_t *tracker, double *vargL)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_get_effects");
    if (rc) return rc;
    HB_HIP(hipStreamSynchronize(c->stream));
    if (g) HB_HIP(hipMemcpy(g, c->g, sizeof(double) * c->m, hipMemcpyDeviceToHost));
    if (tracker) HB_HIP(hipMemcpy(tracker, c->tracker, c->m, hipMemcpyDeviceToHost));
    if (vargL) HB_HIP(hipMemcpy(vargL, c->vargL, sizeof(double) * c->m, hipMemcpyDeviceToHost));
    return HB_OK;
}

int hb_ctx_dot(hb_ctx *c, int32_t col0, int32_t ncols, double *d)
{
    int rc = check_cols(c, col0, ncols, "hb_ctx_dot");
    if (rc) return rc;
    if (!c->stats_ready) {
        rc = hbk_stats(c);
        if (rc) return rc;
    }
    rc = hbk_dot_all(c);
    if (rc) return rc;
    HB_HIP(hipStreamSynchronize(c->stream));
    HB_HIP(hipMemcpy(d, c->dots + col0, sizeof(double) * ncols, hipMemcpyDeviceToHost));
    return HB_OK;
}

int hb_ctx_matvec(hb_ctx *c, const double *alpha, double *out)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_matvec");
    if (rc) return rc;
    if (!alpha || !out) return hb_fail(HB_ERR_INVALID, "hb_ctx_matvec: null argument");
    double *da = nullptr, *dout = nullptr;
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&da), sizeof(double) * (size_t)c->m_pad));
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&dout), sizeof(double) * (size_t)c->ld));
    hipError_t e = hipMemset(da, 0, sizeof(double) * (size_t)c->m_pad);
    if (e == hipSuccess) e = hipMemcpy(da, alpha, sizeof(double) * c->m, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = hbk_xalpha(c, da, dout);
        if (rc == HB_OK) e = hipStreamSynchronize(c->stream);
        if (rc == HB_OK && e == hipSuccess) e = hipMemcpy(out, dout, sizeof(double) * c->n, hipMemcpyDeviceToHost);
    }
    (void)hipFree(da);
    (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("hb_ctx_matvec: ") + hipGetErrorString(e));
    return HB_OK;
}

// out (n x R) = X (n x m) * A (m x R): the GEBV sample matrix MCMCsamples$g = M %*% MCMCsamples$alpha of reference
// R/bayes.r:303-305, eight records per pass over the columns that carry a non-zero effect in any of them
int hb_ctx_matmul(hb_ctx *c, const double *A, int64_t ldA, int32_t R, double *out, int64_t ldo)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_matmul");
    if (rc) return rc;
    if (!A || !out || R < 0 || ldA < c->m || ldo < c->n) return hb_fail(HB_ERR_INVALID, "hb_ctx_matmul: bad argument");
    if (R == 0) return HB_OK;
    const int RB = 8;
    int *didx = nullptr;
    double *dval = nullptr, *dout = nullptr;
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&didx), sizeof(int) * (size_t)c->m));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dval), sizeof(double) * (size_t)c->m * RB);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dout), sizeof(double) * (size_t)c->ld * RB);
    std::vector<int> idx;
    std::vector<double> val, ho((size_t)c->ld * RB);
    for (int r0 = 0; r0 < R && e == hipSuccess && rc == HB_OK; r0 += RB) {
        const int nr = std::min(RB, R - r0);
        idx.clear();
        val.clear();
        for (int j = 0; j < c->m; j++) {
            bool any = false;
            for (int r = 0; r < nr; r++) any |= A[(size_t)(r0 + r) * ldA + j] != 0.0;
            if (!any) continue;
            idx.push_back(j);
            for (int r = 0; r < RB; r++) val.push_back(r < nr ? A[(size_t)(r0 + r) * ldA + j] : 0.0);
        }
        const int nnz = (int)idx.size();
        if (nnz) {
            e = hipMemcpyAsync(didx, idx.data(), sizeof(int) * nnz, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dval, val.data(), sizeof(double) * (size_t)nnz * RB, hipMemcpyHostToDevice, c->stream);
        }
        if (e != hipSuccess) break;
        rc = hbk_xmat(c, didx, dval, nnz, dout);
        if (rc) break;
        e = hipMemcpyAsync(ho.data(), dout, sizeof(double) * (size_t)c->ld * RB, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) break;
        for (int r = 0; r < nr; r++) std::memcpy(out + (size_t)(r0 + r) * ldo, ho.data() + (size_t)r * c->ld, sizeof(double) * c->n);
    }
    (void)hipFree(didx);
    if (dval) (void)hipFree(dval);
    if (dout) (void)hipFree(dout);
    if (rc) return rc;
    if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("hb_ctx_matmul: ") + hipGetErrorString(e));
    return HB_OK;
}

// Diagnostics of a pipeline time-out (HB_DEBUG_ABORT=1): the abort log of the waiters (hb_abort_log), what memory holds now at the
// words they were waiting for, and the start / end of the mat-vec launches around the stall (hb_ldiag_note).
static void print_abort_diagnostics(hb_ctx *c)
{
    std::vector<unsigned> f(4096, 0u);
    if (hipMemcpy(f.data(), c->flags, sizeof(unsigned) * 4096, hipMemcpyDeviceToHost) != hipSuccess) return;
    const unsigned nlog = std::min<unsigned>(f[64], 480u);
    static const char *kinds[] = {"?", "chain<-dsum", "chain<-fcorr", "chain<-fcorr2", "fold<-dd", "update-rows<-dd/mb", "wait_ge", "group-chain"};
    unsigned long long first_clock = ~0ull;
    int stall_panel = -1;
    fprintf(stderr, "abort log: %u records (chain_done %u)\n", f[64], f[0]);
    for (unsigned i = 0; i < nlog; i++) {
        const unsigned *r = f.data() + 128 + 8 * i;
        const unsigned kind = r[0] & 0xffffu, own = (r[0] >> 16) & 1u, xcc = (r[0] >> 20) & 15u;
        const unsigned long long clk = ((unsigned long long)r[5] << 32) | r[4], seen = ((unsigned long long)r[7] << 32) | r[6];
        unsigned long long mem = 0;
        const double *src = kind == 1 ? c->dsum : kind == 2 ? c->fcorr : kind == 3 ? c->fcorr2 : kind == 4 ? c->ddense : nullptr;
        if (src && r[1] < (unsigned)c->m_pad) (void)hipMemcpy(&mem, src + r[1], 8, hipMemcpyDeviceToHost);
        if (own && clk < first_clock) first_clock = clk;
        if ((kind == 1 || kind == 2) && stall_panel < 0) stall_panel = (int)r[2];
        if (i < 24 || own)
            fprintf(stderr, "  [%3u] %-18s own=%u xcd=%u block=%u a=%u (panel %d, +%d) b=%u/%u clock=%llu saw=%016llx memory-now=%016llx\n", i,
                    kinds[std::min(kind, 7u)], own, xcc, r[3], r[1], (int)(r[1] / (unsigned)c->P), (int)(r[1] % (unsigned)c->P), r[2] & 0xffffu, r[2] >> 16, clk, seen, mem);
    }
    if (c->ldiag) {
        const int G = (c->npanels + c->D - 1) / c->D;
        std::vector<unsigned long long> ld((size_t)(c->npanels + 2) * 4, 0ull);
        (void)hipMemcpy(ld.data(), c->ldiag, sizeof(unsigned long long) * ld.size(), hipMemcpyDeviceToHost);
        const int gs = stall_panel >= 0 ? stall_panel / c->D : 0;
        fprintf(stderr, "mat-vec launches around group %d (first own time-out at clock %llu; 100 MHz):\n", gs, first_clock);
        for (int g = std::max(0, gs - 3); g <= std::min(G, gs + 4); g++)
            fprintf(stderr, "  launch %4d: start %llu (%+.1f us vs the time-out)  last block end %llu (%+.1f us)  sampled blocks (every 32nd) finished %llu, launch has %d\n", g, ld[4 * (size_t)g],
                    ((double)ld[4 * (size_t)g] - (double)first_clock) / 100.0, ld[4 * (size_t)g + 1], ((double)ld[4 * (size_t)g + 1] - (double)first_clock) / 100.0,
                    ld[4 * (size_t)g + 2], g < (int)c->ldiag_nblk.size() ? c->ldiag_nblk[g] : -1);
    }
}

static int fetch_acc(hb_ctx *c)
{
    HB_HIP(hipMemcpyAsync(c->h_acc, c->acc, sizeof(double) * HB_ACC_N, hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipMemcpyAsync(c->h_flags, c->flags, 256, hipMemcpyDeviceToHost, c->stream));
    if (c->blk_n) HB_HIP(hipMemcpyAsync(c->h_blk, c->blk, sizeof(double) * c->blk_n, hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    c->aborted = false;
    if (getenv("HB_DEBUG_STARTS") && c->h_flags[13]) { // (development aid: how long after the dense chain its fold workgroups started)
        const long long ch = ((long long)c->h_flags[17] << 32) | c->h_flags[16], fo = ((long long)c->h_flags[19] << 32) | c->h_flags[18];
        fprintf(stderr, "hibayes_gpu: k_fold_dense's first workgroup started %.1f us after k_chain_dense's first panel (%u fold workgroups started)\n", (double)(fo - ch) * 1e-2, c->h_flags[13]);
    }
    if (c->ldiag) { // (HB_DEBUG_ABORT) a long wait that flushed its L2 and went on leaves no other trace
        static unsigned seen = 0;
        const unsigned now = hbk_long_wait_flushes();
        if (now != seen) {
            fprintf(stderr, "hibayes_gpu: %u long waits flushed their L2 so far (+%u in this sweep)%s\n", now, now - seen, c->h_flags[1] ? " — and the sweep was aborted" : "");
            seen = now;
        }
    }
    // (a sharded sweep: the rank whose pipeline gave up poisons the sums it contributes, so every rank sees a NaN here — hb_run.hip)
    if (c->h_flags[1] || c->h_acc[HB_ACC_EVENTS] != c->h_acc[HB_ACC_EVENTS]) {
        c->aborted = true;
        char msg[320];
        snprintf(msg, sizeof msg, "device pipeline timed out waiting on a flag (sweep aborted) [chain_done %u; gave up: update rows waiting for %u, chain at panel %u (%u), fold at panel %u step %u, fold workgroups started %u, targets delivered %u%s]",
                 c->h_flags[0], c->h_flags[8], c->h_flags[9], c->h_flags[10], c->h_flags[11], c->h_flags[12], c->h_flags[13], c->h_flags[14],
                 c->h_flags[1] ? "" : "; raised on another rank");
        if (getenv("HB_DEBUG_ABORT") && c->h_flags[1]) print_abort_diagnostics(c);
        if (getenv("HB_DEBUG_ABORT") && c->mvp_ho) { // (the persistent mat-vec's hand-over counters: versions signed off, finalize tickets of the first groups)
            const int ng = (c->npanels + c->D - 1) / c->D;
            std::vector<unsigned> ho((size_t)ng + 1 + 4 * 64);
            if (hipMemcpy(ho.data(), c->mvp_ho, sizeof(unsigned) * ho.size(), hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "  chain progress markers (HB_CHAINDBG builds): group %u phase %u; waves %u %u %u %u %u %u %u %u\n", c->h_flags[20], c->h_flags[21], c->h_flags[24], c->h_flags[25], c->h_flags[26], c->h_flags[27], c->h_flags[28], c->h_flags[29], c->h_flags[30], c->h_flags[31]);
                fprintf(stderr, "  chain opening: looks %u, lanes still missing per wave %u %u %u %u %u %u %u %u, thread 0: dj[0] hi %08x fc[0] hi %08x far_in %u\n", c->h_flags[22], c->h_flags[32], c->h_flags[33], c->h_flags[34], c->h_flags[35], c->h_flags[36], c->h_flags[37], c->h_flags[38], c->h_flags[39], c->h_flags[40], c->h_flags[41], c->h_flags[42]);
                fprintf(stderr, "  chain clocks (low 32 bits, 100 MHz): staged %u, poll loop entered %u, last look began %u (the abort log's clocks modulo 2^32 are on the same scale)\n", c->h_flags[43], c->h_flags[44], c->h_flags[45]);
                fprintf(stderr, "  mvp: versions signed off by update workgroups: slots 0..7 =");
                for (int i = 0; i < 8 && i <= ng; i++) fprintf(stderr, " %u", ho[i]);
                fprintf(stderr, "; finalize tickets of group 0, column groups 0..15 =");
                for (int i = 0; i < 64; i++) fprintf(stderr, " %u", ho[(size_t)ng + 1 + i]);
                fprintf(stderr, "; of group 3 =");
                for (int i = 0; i < 8; i++) fprintf(stderr, " %u", ho[(size_t)ng + 1 + 3 * 64 + i]);
                std::vector<double> ds((size_t)c->D * c->P);
                if (hipMemcpy(ds.data(), c->dsum, sizeof(double) * ds.size(), hipMemcpyDeviceToHost) == hipSuccess) {
                    int bad = 0, first = -1;
                    for (size_t i = 0; i < ds.size(); i++)
                        if (ds[i] != ds[i]) { if (first < 0) first = (int)i; bad++; }
                    fprintf(stderr, "; dsum of group 0: %d of %zu words not a number (first at %d); [0..3] = %g %g %g %g", bad, ds.size(), first, ds[0], ds[1], ds[2], ds[3]);
                }
                fprintf(stderr, "\n");
            }
        }
        if (c->h_flags[15]) // (k_chain_group: every word of a group was delivered and one of them is not finite — a numerical fault upstream, not a lost hand-off)
            return hb_fail(HB_ERR_ABORTED, std::string("a non-finite right-hand side reached the chain (dot product or correction overflowed): ") + msg);
        return hb_fail(HB_ERR_ABORTED, msg);
    }
    return HB_OK;
}

// ---- snapshot / restore around a sweep (hb_internal.hpp) ----
int hb_ctx_snapshot(hb_ctx *c, int model_index, bool store, bool count_pip)
{
    std::vector<hb_ctx::snap_seg> segs;
    size_t off = 0;
    auto add = [&](void *p, size_t bytes) {
        if (!p || !bytes) return;
        segs.push_back({p, off, bytes});
        off += (bytes + 255) / 256 * 256;
    };
    const size_t mp = (size_t)c->m_pad;
    add(c->g, sizeof(double) * mp);
    add(c->tracker, mp);
    add(c->r, sizeof(double) * (size_t)c->ld);
    add(c->r32, sizeof(float) * (size_t)c->ld);
    add(c->u, sizeof(double) * (size_t)c->ld);
    if (model_index == 5) add(c->vargL, sizeof(double) * mp);
    if (store) {
        add(c->alpha_sum, sizeof(double) * mp);
        add(c->alpha_sq, sizeof(double) * mp);
    }
    if (count_pip) {
        add(c->nzrate, sizeof(uint32_t) * mp);
        if (c->nw) {
            add(c->wflag, (size_t)c->nw);
            add(c->wppa, sizeof(double) * (size_t)c->nw);
        }
    }
    if (off > c->snap_cap) {
        HB_HIP(hipStreamSynchronize(c->stream));
        if (c->snap) (void)hipFree(c->snap);
        c->snap = nullptr;
        c->snap_cap = 0;
        // (room for every segment a later sweep of the run may add: the counters and moments start after the burn-in)
        const size_t cap = off + (sizeof(double) * 3 + sizeof(uint32_t)) * mp + 9 * (size_t)c->nw + 8 * 256;
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->snap), cap));
        c->snap_cap = cap;
    }
    c->snap_segs = segs;
    return hbk_copy_segs(c, segs, false);
}

int hb_ctx_restore(hb_ctx *c)
{
    if (!c->snap || c->snap_segs.empty()) return hb_fail(HB_ERR_INVALID, "hb_ctx_restore: no snapshot");
    HB_HIP(hipStreamSynchronize(c->stream)); // (every kernel of the aborted sweep has left: the fetch waited for them)
    int rc = hbk_copy_segs(c, c->snap_segs, true);
    if (rc) return rc;
    HB_HIP(hipMemsetAsync(c->flags, 0, sizeof(unsigned) * 4096, c->stream));
    HB_HIP(hipMemsetAsync(c->ru_ws, 0, sizeof(double) * 128, c->stream)); // k_reduce_ru's tickets: zero at rest, also after a sweep that did not end
    c->aborted = false;
    return HB_OK;
}

int hb_ctx_set_matvec_kernel(hb_ctx *c, int32_t kind)
{
    if (!c) return hb_fail(HB_ERR_INVALID, "hb_ctx_set_matvec_kernel: null context");
    if (kind < 0 || kind > 2) return hb_fail(HB_ERR_INVALID, "hb_ctx_set_matvec_kernel: kind must be 0 (lane = column, v_dot4), 1 (individuals across the lanes, v_dot4) or 2 (matrix cores)");
    if (kind != c->dotq2_kind) {
        HB_HIP(hipStreamSynchronize(c->stream));
        c->dotq2_kind = kind;
        c->graph_model = -1; // the captured sweeps hold the other kernel
    }
    return HB_OK;
}

extern "C" int hb_ctx_debug_inject_abort(hb_ctx *c, int32_t panel, int32_t times)
{
    if (!c) return hb_fail(HB_ERR_INVALID, "hb_ctx_debug_inject_abort: null context");
    c->inject_abort_panel = times > 0 ? panel : -1;
    c->inject_abort_times = times;
    if (panel >= 0 && !c->s_dbg) HB_HIP(hipStreamCreateWithFlags(&c->s_dbg, hipStreamNonBlocking));
    return HB_OK;
}

int hb_ctx_residual_sums(hb_ctx *c, double *sum_r, double *sum_r2)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_residual_sums");
    if (rc) return rc;
    rc = hbk_reduce_ru(c);
    if (rc) return rc;
    rc = fetch_acc(c);
    if (rc) return rc;
    if (sum_r) *sum_r = c->h_acc[HB_ACC_SUMR];
    if (sum_r2) *sum_r2 = c->h_acc[HB_ACC_SUMR2];
    return HB_OK;
}

int hb_ctx_residual_shift(hb_ctx *c, double a)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_residual_shift");
    if (rc) return rc;
    return hbk_shift(c, a);
}

int hb_ctx_set_covariates(hb_ctx *c, const double *Cmat, int32_t nc)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_set_covariates");
    if (rc) return rc;
    if (c->Cmat) { (void)hipFree(c->Cmat); c->Cmat = nullptr; }
    c->nc = 0;
    if (!Cmat || nc <= 0) return HB_OK;
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->Cmat), sizeof(double) * (size_t)c->n * nc));
    HB_HIP(hipMemcpy(c->Cmat, Cmat, sizeof(double) * (size_t)c->n * nc, hipMemcpyHostToDevice));
    c->nc = nc;
    return HB_OK;
}

int hb_ctx_cov_dot(hb_ctx *c, int32_t i, double *out)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_cov_dot");
    if (rc) return rc;
    if (i < 0 || i >= c->nc) return hb_fail(HB_ERR_INVALID, "hb_ctx_cov_dot: bad covariate index");
    rc = hbk_cov_dot(c, i, c->scratch);
    if (rc) return rc;
    HB_HIP(hipMemcpyAsync(out, c->scratch, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    return HB_OK;
}

int hb_ctx_cov_axpy(hb_ctx *c, int32_t i, double a)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_cov_axpy");
    if (rc) return rc;
    if (i < 0 || i >= c->nc) return hb_fail(HB_ERR_INVALID, "hb_ctx_cov_axpy: bad covariate index");
    return hbk_cov_axpy(c, i, a);
}

int hb_ctx_set_levels(hb_ctx *c, const int32_t *zid, int32_t nr, const int32_t *nlev)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_set_levels");
    if (rc) return rc;
    if (c->zid) { (void)hipFree(c->zid); c->zid = nullptr; }
    if (c->lev_buf) { (void)hipFree(c->lev_buf); c->lev_buf = nullptr; }
    c->nr = 0;
    c->nlev.clear();
    c->lev_first.clear();
    if (!zid || nr <= 0) return HB_OK;
    int tot = 0, mx = 0;
    for (int t = 0; t < nr; t++) {
        if (nlev[t] < 1) return hb_fail(HB_ERR_INVALID, "hb_ctx_set_levels: bad level count");
        c->lev_first.push_back(tot);
        c->nlev.push_back(nlev[t]);
        tot += nlev[t];
        mx = std::max(mx, nlev[t]);
    }
    for (int t = 0; t < nr; t++)
        for (int i = 0; i < c->n; i++)
            if (zid[(size_t)t * c->n + i] < 0 || zid[(size_t)t * c->n + i] >= nlev[t])
                return hb_fail(HB_ERR_INVALID, "hb_ctx_set_levels: level index out of range");
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->zid), sizeof(int32_t) * (size_t)c->n * nr));
    HB_HIP(hipMemcpy(c->zid, zid, sizeof(int32_t) * (size_t)c->n * nr, hipMemcpyHostToDevice));
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->lev_buf), sizeof(double) * (size_t)mx));
    c->nr = nr;
    c->lev_total = tot;
    return HB_OK;
}

int hb_ctx_level_sums(hb_ctx *c, int32_t term, double *sums)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_level_sums");
    if (rc) return rc;
    if (term < 0 || term >= c->nr) return hb_fail(HB_ERR_INVALID, "hb_ctx_level_sums: bad term");
    rc = hbk_level_sums(c, term, c->lev_buf, c->nlev[term]);
    if (rc) return rc;
    HB_HIP(hipMemcpyAsync(sums, c->lev_buf, sizeof(double) * c->nlev[term], hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    return HB_OK;
}

int hb_ctx_level_axpy(hb_ctx *c, int32_t term, const double *delta)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_level_axpy");
    if (rc) return rc;
    if (term < 0 || term >= c->nr) return hb_fail(HB_ERR_INVALID, "hb_ctx_level_axpy: bad term");
    HB_HIP(hipMemcpyAsync(c->lev_buf, delta, sizeof(double) * c->nlev[term], hipMemcpyHostToDevice, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream)); // delta may be a pageable host temporary
    return hbk_level_axpy(c, term, c->lev_buf);
}

// one sweep in two halves, so that a caller can enqueue more work on the stream (the multi-GPU exchange) before the single
// host synchronisation of the iteration
static void blocks_free(hb_ctx *c)
{
    if (c->blk) (void)hipFree(c->blk);
    if (c->blk_zz) (void)hipFree(c->blk_zz);
    if (c->blk_z) (void)hipFree(c->blk_z);
    if (c->h_blk) (void)hipHostFree(c->h_blk);
    if (c->h_z) (void)hipHostFree(c->h_z);
    c->blk = c->blk_zz = c->blk_z = c->h_blk = c->h_z = nullptr;
    c->blk_n = 0;
}

int hb_ctx_blocks_setup(hb_ctx *c, const double *cpc, const double *zz, const double *vrtmp0)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_blocks_setup");
    if (rc) return rc;
    blocks_free(c);
    c->blk_cpc.assign(cpc, cpc + c->nc);
    const int nb = c->nc + c->lev_total + 2 * c->nr;
    if (nb == 0) return HB_OK;
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->blk), sizeof(double) * nb));
    HB_HIP(hipMemsetAsync(c->blk, 0, sizeof(double) * nb, c->stream)); // (on the context's stream: a null-stream memset is not ordered before it)
    HB_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_blk), sizeof(double) * nb));
    std::memset(c->h_blk, 0, sizeof(double) * nb);
    if (c->lev_total) {
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->blk_zz), sizeof(double) * c->lev_total));
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->blk_z), sizeof(double) * c->lev_total));
        HB_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_z), sizeof(double) * c->lev_total));
        HB_HIP(hipMemcpyAsync(c->blk_zz, zz, sizeof(double) * c->lev_total, hipMemcpyHostToDevice, c->stream));
        HB_HIP(hipMemcpyAsync(c->blk + c->nc + c->lev_total, vrtmp0, sizeof(double) * c->nr, hipMemcpyHostToDevice, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream)); // (the caller's arrays may be temporaries)
    }
    c->blk_n = nb;
    return HB_OK;
}

int hb_ctx_blocks_step(hb_ctx *c, double vare, const double *z_beta, const double *z_levels, const double *chisq, double dfr, double s2r)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_blocks_step");
    if (rc) return rc;
    if (c->nc + c->nr == 0) return HB_OK;
    if (!c->blk) return hb_fail(HB_ERR_INVALID, "hb_ctx_blocks_step: call hb_ctx_blocks_setup first");
    for (int i = 0; i < c->nc; i++) {
        rc = hbk_cov_step(c, i, c->blk_cpc[i], vare, z_beta[i], c->blk + i);
        if (rc) return rc;
    }
    if (c->nr) {
        std::memcpy(c->h_z, z_levels, sizeof(double) * c->lev_total); // pinned staging: the caller's array may be a temporary
        HB_HIP(hipMemcpyAsync(c->blk_z, c->h_z, sizeof(double) * c->lev_total, hipMemcpyHostToDevice, c->stream));
        double *estR = c->blk + c->nc, *vrtmp = estR + c->lev_total, *vr = vrtmp + c->nr;
        for (int t = 0; t < c->nr; t++) {
            rc = hbk_lev_step(c, t, c->lev_first[t], c->nlev[t], c->blk_zz, estR, c->blk_z, vare, vrtmp + t, vr + t, s2r * dfr, chisq[t]);
            if (rc) return rc;
        }
    }
    return HB_OK;
}

int hb_ctx_blocks_state(hb_ctx *c, double *beta, double *estR, double *vrtmp, double *vr)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_blocks_state");
    if (rc) return rc;
    if (!c->blk_n) return HB_OK;
    HB_HIP(hipMemcpyAsync(c->h_blk, c->blk, sizeof(double) * c->blk_n, hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    const double *h = c->h_blk;
    if (beta) std::memcpy(beta, h, sizeof(double) * c->nc);
    if (estR) std::memcpy(estR, h + c->nc, sizeof(double) * c->lev_total);
    if (vrtmp) std::memcpy(vrtmp, h + c->nc + c->lev_total, sizeof(double) * c->nr);
    if (vr) std::memcpy(vr, h + c->nc + c->lev_total + c->nr, sizeof(double) * c->nr);
    return HB_OK;
}

int hb_ctx_sweep_begin(hb_ctx *c, const hb_sweep_in *in)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_sweep");
    if (rc) return rc;
    if (!in) return hb_fail(HB_ERR_INVALID, "hb_ctx_sweep: null argument");
    if (in->model_index < 1 || in->model_index > 6) return hb_fail(HB_ERR_INVALID, "hb_ctx_sweep: bad model_index");
    if (in->n_fold < 2 || in->n_fold > HB_MAX_FOLD) return hb_fail(HB_ERR_INVALID, "hb_ctx_sweep: bad n_fold");
    if (in->model_index == 6)
        for (int k = 2; k < in->n_fold; k++)
            if (!(in->fold[k] > in->fold[k - 1]))
                return hb_fail(HB_ERR_UNSUPPORTED, "BayesR on the GPU path needs 'fold' in strictly increasing order");
    if (!c->stats_ready) {
        rc = hbk_stats(c);
        if (rc) return rc;
    }
    if (!c->gram_ready) {
        rc = hb_ctx_build_gram(c, nullptr);
        if (rc) return rc;
    }
    rc = hb_sweep_enqueue(c, in, c->profiling);
    if (rc) return rc;
    if (in->count_pip && c->nw && (c->rng_pe == 0 || c->rng_last)) {
        rc = hbk_windows(c);
        if (rc) return rc;
    }
    return HB_OK;
}

// One block of a sweep that comes in `nblocks` ranges of whole mat-vec groups (sync_every_blocks of the sharded sweep: the
// caller exchanges the shards' residual deltas between the blocks). Block 0 prepares the sweep, the last block closes it;
// hb_ctx_sweep_end() fetches the results as usual. Where the persistent pipeline is not available the first block runs the
// whole sweep with the per-panel kernels and the others do nothing (the exchange between them then moves zeros).
int hb_ctx_sweep_range(hb_ctx *c, const hb_sweep_in *in, int block, int nblocks)
{
    if (!c) return hb_fail(HB_ERR_INVALID, "hb_ctx_sweep_range: null context");
    if (nblocks <= 1) return block == 0 ? hb_ctx_sweep_begin(c, in) : HB_OK;
    if (!c->pipeline || c->profiling) return block == 0 ? hb_ctx_sweep_begin(c, in) : HB_OK; // (the HIP-event timing mode runs the whole sweep with the per-panel kernels)
    const int G = (c->npanels + c->D - 1) / c->D;
    const int nb = std::min(nblocks, G);
    if (block >= nb) return HB_OK;
    const int g_lo = (int)((long long)G * block / nb), g_hi = (int)((long long)G * (block + 1) / nb);
    c->rng_pb = g_lo * c->D;
    c->rng_pe = std::min(c->npanels, g_hi * c->D);
    c->rng_first = block == 0;
    c->rng_last = block == nb - 1;
    const int rc = hb_ctx_sweep_begin(c, in);
    c->rng_pe = 0;
    c->rng_pb = 0;
    return rc;
}

int hb_ctx_sweep_end(hb_ctx *c, hb_sweep_out *out)
{
    if (!out) return hb_fail(HB_ERR_INVALID, "hb_ctx_sweep: null argument");
    int rc = fetch_acc(c);
    if (rc) return rc;
    const double *a = c->h_acc;
    out->sum_g2 = a[HB_ACC_SUMG2];
    for (int k = 0; k < HB_MAX_FOLD; k++) out->class_count[k] = a[HB_ACC_COUNT0 + k];
    out->sum_vargL = a[HB_ACC_SUMVARGL];
    out->sum_r = a[HB_ACC_SUMR];
    out->sum_r2 = a[HB_ACC_SUMR2];
    out->var_u = a[HB_ACC_VARU];
    out->n_events = a[HB_ACC_EVENTS];
    out->n_cache_miss = a[HB_ACC_MISS];
    out->n_redo = a[HB_ACC_REDO];
    return HB_OK;
}

int hb_ctx_sweep(hb_ctx *c, const hb_sweep_in *in, hb_sweep_out *out)
{
    if (!in || !out) return hb_fail(HB_ERR_INVALID, "hb_ctx_sweep: null argument");
    int rc = hb_ctx_sweep_begin(c, in);
    if (rc) return rc;
    return hb_ctx_sweep_end(c, out);
}

int hb_ctx_get_counters(hb_ctx *c, double *nzrate, double *alpha_sum, double *alpha_sq)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_get_counters");
    if (rc) return rc;
    HB_HIP(hipStreamSynchronize(c->stream));
    if (nzrate) {
        std::vector<uint32_t> t(c->m);
        HB_HIP(hipMemcpy(t.data(), c->nzrate, sizeof(uint32_t) * c->m, hipMemcpyDeviceToHost));
        for (int i = 0; i < c->m; i++) nzrate[i] = (double)t[i];
    }
    if (alpha_sum) HB_HIP(hipMemcpy(alpha_sum, c->alpha_sum, sizeof(double) * c->m, hipMemcpyDeviceToHost));
    if (alpha_sq) HB_HIP(hipMemcpy(alpha_sq, c->alpha_sq, sizeof(double) * c->m, hipMemcpyDeviceToHost));
    return HB_OK;
}

int hb_ctx_set_windows(hb_ctx *c, const uint32_t *windindx, int32_t nw)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_set_windows");
    if (rc) return rc;
    if (c->wind) { (void)hipFree(c->wind); c->wind = nullptr; }
    if (c->wflag) { (void)hipFree(c->wflag); c->wflag = nullptr; }
    if (c->wppa) { (void)hipFree(c->wppa); c->wppa = nullptr; }
    c->nw = 0;
    c->graph_model = -1; // the captured graph holds the old pointers
    if (!windindx || nw <= 0) return HB_OK;
    for (int i = 0; i < c->m; i++)
        if (windindx[i] < 1 || (int)windindx[i] > nw) return hb_fail(HB_ERR_INVALID, "hb_ctx_set_windows: window id out of range");
    std::vector<uint32_t> w(c->m_pad, 1u);
    std::memcpy(w.data(), windindx, sizeof(uint32_t) * c->m);
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->wind), sizeof(uint32_t) * c->m_pad));
    HB_HIP(hipMemcpy(c->wind, w.data(), sizeof(uint32_t) * c->m_pad, hipMemcpyHostToDevice));
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->wflag), (size_t)nw));
    HB_HIP(hipMemsetAsync(c->wflag, 0, (size_t)nw, c->stream));
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->wppa), sizeof(double) * nw));
    HB_HIP(hipMemsetAsync(c->wppa, 0, sizeof(double) * nw, c->stream));
    c->nw = nw;
    return HB_OK;
}

int hb_ctx_get_windows(hb_ctx *c, double *wppa)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_get_windows");
    if (rc) return rc;
    if (!c->nw) return hb_fail(HB_ERR_INVALID, "hb_ctx_get_windows: no windows set");
    HB_HIP(hipStreamSynchronize(c->stream));
    HB_HIP(hipMemcpy(wppa, c->wppa, sizeof(double) * c->nw, hipMemcpyDeviceToHost));
    return HB_OK;
}

int hb_ctx_time_matvec(hb_ctx *c, int32_t reps, double *avg_ms, int32_t *launches_per_sweep, int32_t *cols_per_launch)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_time_matvec");
    if (rc) return rc;
    if (!c->stats_ready) {
        rc = hbk_stats(c);
        if (rc) return rc;
    }
    double us = 0;
    int nl = 0;
    int D = c->pipeline ? c->D : 1, as_pipeline = c->pipeline;
    // (a counter-collecting profiler serialises kernels, the probe then turns the pipeline off and with it the launch width:
    // the counter passes of tools/matvec_only.py ask for the pipeline's launch shape explicitly)
    if (const char *e = getenv("HB_TIME_MATVEC_D")) {
        D = std::max(1, std::min(8, atoi(e)));
        as_pipeline = 1;
    }
    rc = hbk_time_matvec(c, D, reps > 0 ? reps : 1, as_pipeline, &us, &nl);
    if (rc) return rc;
    if (avg_ms) *avg_ms = us * 1e-3;
    if (launches_per_sweep) *launches_per_sweep = nl;
    if (cols_per_launch) *cols_per_launch = D * c->P;
    return HB_OK;
}

int hb_ctx_time_stream_read(hb_ctx *c, int32_t reps, double *avg_ms, int64_t *bytes)
{
    if (!c || !avg_ms || !bytes) return hb_fail(HB_ERR_INVALID, "hb_ctx_time_stream_read: null argument");
    return hbk_time_stream_read(c, reps > 0 ? reps : 1, avg_ms, bytes);
}

int hb_ctx_last_timing(hb_ctx *c, hb_sweep_timing *t)
{
    if (!c || !t) return hb_fail(HB_ERR_INVALID, "hb_ctx_last_timing: null argument");
    *t = c->timing;
    return HB_OK;
}

int hb_ctx_set_profiling(hb_ctx *c, int32_t on)
{
    if (!c) return hb_fail(HB_ERR_INVALID, "hb_ctx_set_profiling: null context");
    c->profiling = (on & 1) != 0;
    if ((on & 2) && !c->dbg) { // bit 1: cycle stamps inside k_chain (development aid)
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->dbg), sizeof(long long) * 32 * (size_t)c->npanels));
        HB_HIP(hipMemsetAsync(c->dbg, 0, sizeof(long long) * 32 * (size_t)c->npanels, c->stream));
        c->graph_model = -1;
    }
    if (((on & 8) != 0) != (c->lstamp != nullptr)) {
        // bit 3: every block of every mat-vec launch of a sweep records the 100 MHz clock at its start and end
        // (hb_ctx_matvec_stamps): the in-situ duration of the pipeline's launches, chain and update rows running beside them
        if (on & 8) {
            const size_t cnt = (size_t)(c->npanels + 1) * HB_LSTAMP_BLOCKS * 2;
            HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->lstamp), sizeof(unsigned long long) * cnt));
            HB_HIP(hipMemsetAsync(c->lstamp, 0, sizeof(unsigned long long) * cnt, c->stream));
            c->lstamp_nblk.assign((size_t)c->npanels + 1, 0);
            c->lstamp_cols.assign((size_t)c->npanels + 1, 0);
        } else {
            HB_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->lstamp);
            c->lstamp = nullptr;
        }
        c->graph_model = -1;
    }
    if (((on & 4) != 0) != c->chain_alone) {
        // bit 2: the persistent pipeline's kernels with the mat-vec launches first and the chain workgroup alone afterwards —
        // no co-residency needed, so it also runs where kernels serialise (rocprofv3 --pmc). The sweeps it runs are NOT the
        // chain (the residual updates see empty move lists): for cycle stamps and hardware counters of k_chain_persist only.
        c->chain_alone = (on & 4) != 0;
        if (c->chain_alone) c->concurrent = true; // (the probe's verdict does not apply to this mode)
        c->graph_model = -1;
    }
    return HB_OK;
}

int hb_ctx_matvec_stamps(hb_ctx *c, hb_launch_stats *o)
{
    int rc = check_cols(c, 0, 0, "hb_ctx_matvec_stamps");
    if (rc) return rc;
    if (!o) return hb_fail(HB_ERR_INVALID, "hb_ctx_matvec_stamps: null argument");
    if (!c->lstamp) return hb_fail(HB_ERR_INVALID, "hb_ctx_matvec_stamps: call hb_ctx_set_profiling(c, 8) before the sweep");
    HB_HIP(hipStreamSynchronize(c->stream));
    *o = hb_launch_stats{};
    std::vector<unsigned long long> h((size_t)HB_LSTAMP_BLOCKS * 2);
    unsigned long long first = ~0ull, last = 0;
    double sum = 0, mn = 1e300, mx = 0;
    int nl = 0;
    for (int g = 0; g <= c->npanels; g++) {
        const int nb = c->lstamp_nblk[g];
        if (nb <= 0) continue;
        const bool full = c->lstamp_cols[g] == (c->pipeline ? c->D : 1) * c->P; // (the sweep's last launch may be narrower)
        HB_HIP(hipMemcpy(h.data(), c->lstamp + (size_t)g * HB_LSTAMP_BLOCKS * 2, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
        unsigned long long s0 = ~0ull, e1 = 0;
        for (int b = 0; b < nb; b++) {
            if (h[2 * b] == 0) continue; // (a block that did not run in the last sweep)
            s0 = std::min(s0, h[2 * b]);
            e1 = std::max(e1, h[2 * b + 1]);
        }
        if (e1 == 0) continue;
        const double ms = (double)(e1 - s0) * 1e-5; // 100 MHz ticks -> ms
        first = std::min(first, s0);
        last = std::max(last, e1);
        o->launches_all++;
        if (!full) continue;
        sum += ms;
        mn = std::min(mn, ms);
        mx = std::max(mx, ms);
        o->blocks = nb;
        nl++;
    }
    if (!nl) return hb_fail(HB_ERR_INVALID, "hb_ctx_matvec_stamps: no stamped launch (the fixed-point mat-vec of the pipeline only)");
    o->launches = nl;
    o->avg_ms = sum / nl;
    o->min_ms = mn;
    o->max_ms = mx;
    o->sum_ms = sum;
    o->span_ms = (double)(last - first) * 1e-5;
    o->cols_per_launch = (c->pipeline ? c->D : 1) * c->P;
    return HB_OK;
}

// development aid (not in the header): row-cache prediction width and chain-candidate factor
int hb_ctx_debug_tune(hb_ctx *c, double kappa, double candf)
{
    if (!c) return hb_fail(HB_ERR_INVALID, "hb_ctx_debug_tune: null context");
    c->kappa = kappa;
    c->candf = std::min(1.0, std::max(0.0, candf));
    c->graph_model = -1;
    return HB_OK;
}

// development aid (not in the header): the raw block stamps of mat-vec launch g of the last sweep (set_profiling(8)), 2 per block
int hb_ctx_debug_launch_stamps(hb_ctx *c, int g, unsigned long long *out, int cap, int *nblocks)
{
    if (!c || !c->lstamp || g < 0 || g > c->npanels || !out || !nblocks) return hb_fail(HB_ERR_INVALID, "hb_ctx_debug_launch_stamps: bad argument");
    HB_HIP(hipStreamSynchronize(c->stream));
    const int nb = std::min(cap, c->lstamp_nblk[g]);
    *nblocks = nb;
    if (nb > 0) HB_HIP(hipMemcpy(out, c->lstamp + (size_t)g * HB_LSTAMP_BLOCKS * 2, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
    return HB_OK;
}

// development aid (not in the header): the per-launch diagnostics of the last sweep (HB_DEBUG_ABORT=1), 4 words per launch
int hb_ctx_debug_ldiag(hb_ctx *c, unsigned long long *out)
{
    if (!c || !c->ldiag || !out) return hb_fail(HB_ERR_INVALID, "hb_ctx_debug_ldiag: start the process with HB_DEBUG_ABORT=1");
    HB_HIP(hipStreamSynchronize(c->stream));
    HB_HIP(hipMemcpy(out, c->ldiag, sizeof(unsigned long long) * 4 * ((size_t)c->npanels + 2), hipMemcpyDeviceToHost));
    return HB_OK;
}

int hb_ctx_debug_stamps(hb_ctx *c, long long *out)
{
    if (!c || !c->dbg) return hb_fail(HB_ERR_INVALID, "hb_ctx_debug_stamps: stamps not enabled");
    HB_HIP(hipStreamSynchronize(c->stream));
    HB_HIP(hipMemcpy(out, c->dbg, sizeof(long long) * 32 * (size_t)c->npanels, hipMemcpyDeviceToHost));
    return HB_OK;
}

} // extern "C"
