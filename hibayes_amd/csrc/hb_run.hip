// hb_run.hip — the host side of Bayes() (reference src/Bayes.cpp:60-1094) as a stepwise run object.
//
// What stays on the host, as in the reference: argument validation (:92-117, :293, :325, :357), prior
// defaults (:319-374), the outer MCMC loop (:477), the intercept / covariate / random-effect draws
// (:479-516), the hyper-parameter draws after the marker sweep (:603, :666-669, :710-716, :738-741,
// :803-814, :819-823), the thinned store (:848-882) and the posterior assembly (:919-1040).
// What runs on the device: everything that touches an n- or m-long vector (hb_kernels.hip).
// BSLMM (nk) and the single-step epsilon block (ne) are refused with HB_ERR_UNSUPPORTED.
//
// hb_bayes_run() == hb_run_create() + hb_run_step(niter) + hb_run_finish(); bench.py drives the
// three separately so that exactly K iterations sit between its barriers.
#include "hb_internal.hpp"
#include "hb_rng.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>

int hbk_delta_pack(hb_ctx *c, const double *r0, const double *u0, double *buf);
int hbk_delta_unpack(hb_ctx *c, const double *r0, const double *u0, const double *buf);
int hbk_reduce_ru(hb_ctx *c);
int hbk_abort_poison(hb_ctx *c, double *sums);
int hbk_xalpha(hb_ctx *c, const double *dev_alpha, double *dev_out);
extern "C" int hb_comm_world(const hb_comm *c);
extern "C" int hb_comm_rank(const hb_comm *c);

namespace {

using clk = std::chrono::steady_clock;

double arma_sum(const double *v, size_t n)
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

// arma::var, N-1 (two-pass)
double var_n1(const double *v, size_t n)
{
    if (n < 2) return 0.0;
    const double mean = arma_sum(v, n) / n;
    double a2 = 0, a3 = 0;
    for (size_t i = 0; i < n; i++) {
        const double t = mean - v[i];
        a2 += t * t;
        a3 += t;
    }
    return (a2 - a3 * a3 / n) / (n - 1);
}

} // namespace

struct hb_run {
    // ---- arguments (deep copies: the caller's arrays need not outlive hb_run_create) ----
    hb_bayes_args a{};
    std::string model;
    std::vector<double> y, Cmat, Pi, fold_, g_init;
    bool has_warm = false;     // hb_bayes_args.warm: continue from a reported state instead of the prior defaults
    hb_warm_state warm_{};
    std::vector<double> warm_vargL;
    std::vector<uint32_t> wind;
    int n = 0, m = 0, model_index = 0, n_pi = 0, n_fold = 0, nc = 0, nr = 0, world = 1;
    bool fixpi = false, always_in = false, sharded = false;
    bool rowmode = false;      // hb_bayes_args.shard_rows: individuals sharded, exact digit-sum all-reduce per panel
    int64_t n_glob = 0, row_off = 0;
    int64_t m_global = 0;
    int niter = 0, nburn = 0, thin = 1, n_records = 0;
    // ---- device ----
    hb_ctx *c = nullptr;
    bool own_ctx = false;
    double *r0 = nullptr, *u0 = nullptr, *xbuf_own = nullptr, *xbuf = nullptr;
    size_t xcount = 0;
    // ---- state of the chain ----
    double vary = 0, sumvx = 0, ymean_glob = 0;
    int nvar0 = 0, nw = 0, n_levels = 0;
    std::vector<double> beta, cpc, beta_sum, vr, vrtmp, vr_sum, zz, estR, estR_sum, vara_fold, fold_snp_num, pi_sum;
    std::vector<int32_t> zid, nlev, lev_first;
    std::vector<int> cls_of; // device class index -> the caller's (BayesR with an unsorted `fold`)
    double dfr = -1, s2r = 0, dfvara_ = 4, vara_ = 0, vare_ = 0, dfvare_ = -2, s2vara_ = 0, varg = 0, s2varg_ = 0,
           s2vare_ = 0, lambda2 = 0, lambda = 0, shape0 = 1.1, rate0 = 0, mu = 0;
    double sum_r = 0, sum_r2 = 0;
    int iter = 0, count = 0, nzct = 0;
    long long NnzSnp = 0;
    double mu_sum = 0, vara_sum = 0, vare_sum = 0, hsq_sum = 0, events_sum = 0, miss_sum = 0, redo_sum = 0;
    int sync_blocks = 1;       // runs of mat-vec groups per sweep, an exchange after each (hb_bayes_args.sync_blocks)
    bool recover_on = true;    // replay a sweep whose pipeline timed out (HB_RECOVER=0: fail the run, as before round 4)
    int pipeline_setup = -1;   // sharded runs: the context's pipeline switch as the ranks agreed on it in setup (checked at every step)
    int aborts = 0;            // sweeps replayed so far
    int abort_win_start = 0, abort_win_count = 0, slow_timeout = 0; // three aborts within 64 iterations: a slow device, the run's time-out is raised
    bool adaptive_geo = false; // choose (Lv, D) per sweep from the previous sweep's moves (BayesB/C; round 6: BayesR)
    int geo_wide_lv = 2;       // look-ahead groups of the wide geometry (3 with k_fwd beside the chain, else 2)
    int geo_cur = 0;           // 0: the wide geometry — (2 | 3, 7) for BayesB / C, (2, 2) for BayesR; 1: the narrow one — (2, 2), (2, 1)
    int geo_wide_d = 7, geo_narrow_lv = 2, geo_narrow_d = 2;
    double geo_to_wide = 2.0, geo_to_narrow = 2.6; // moves per panel of the previous sweep below / above which the geometry changes
    double last_events_pp = 0;
    bool done = false;
    double setup_seconds = 0, gram_seconds = 0, loop_seconds = 0;
    // MCMC sample stores kept inside the run (copied out by finish)
    std::vector<double> zb_, zl_, ch_; // this iteration's pre-drawn deviates for the covariate / random-effect blocks
    std::vector<double> s_mu, s_Vg, s_Ve, s_h2, s_pi, s_beta, s_Vr, s_r, s_alpha;

    ~hb_run()
    {
        if (r0) (void)hipFree(r0);
        if (u0) (void)hipFree(u0);
        if (xbuf_own) (void)hipFree(xbuf_own);
        if (c && own_ctx) hb_ctx_destroy(c);
    }

    void line(const char *fmt, ...) const
    {
        if (!a.verbose) return;
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        if (a.log) a.log(buf, a.log_user);
        else { fputs(buf, stdout); fputc('\n', stdout); fflush(stdout); }
    }

    // sum over ranks of xbuf[0 .. xcount), on the sweep stream: RCCL inside the library when a communicator was given
    // (nothing but an enqueue), else the host language's callback (needs the stream quiesced on both sides)
    int exchange()
    {
        if (a.comm) return hb_comm_allreduce_f64(a.comm, xbuf, xcount, c->stream);
        HB_HIP(hipStreamSynchronize(c->stream));
        if (a.allreduce(xbuf, xcount, a.allreduce_user)) return hb_fail(HB_ERR_COMM, "all-reduce callback failed");
        return HB_OK;
    }

    int allreduce_host(double *vals, int cnt)
    { // a few host scalars through the device exchange buffer
        if (!sharded && !rowmode) return HB_OK;
        if (rowmode && world == 1 && !a.comm && !a.allreduce) return HB_OK; // (one shard: the sum over the ranks is the value itself)
        if ((size_t)cnt > xcount) return hb_fail(HB_ERR_INVALID, "allreduce_host: too many values");
        HB_HIP(hipMemsetAsync(xbuf, 0, sizeof(double) * xcount, c->stream));
        HB_HIP(hipMemcpyAsync(xbuf, vals, sizeof(double) * cnt, hipMemcpyHostToDevice, c->stream));
        int rc = exchange();
        if (rc) return rc;
        HB_HIP(hipMemcpyAsync(vals, xbuf, sizeof(double) * cnt, hipMemcpyDeviceToHost, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
        return HB_OK;
    }

    // any number of host values summed over the ranks, in place (chunks of the exchange buffer)
    int allreduce_chunks(double *vals, size_t cnt)
    {
        for (size_t k0 = 0; k0 < cnt; k0 += xcount) {
            const int rc = allreduce_host(vals + k0, (int)std::min(xcount, cnt - k0));
            if (rc) return rc;
        }
        return HB_OK;
    }
    static int row_hook(void *user, double *vals, size_t cnt) { return static_cast<hb_run *>(user)->allreduce_chunks(vals, cnt); }
    // maximum over the ranks of one non-negative value each (a sum over one-hot slots)
    int allreduce_max(double *v)
    {
        std::vector<double> slot(world, 0.0);
        slot[a.rank] = *v;
        const int rc = allreduce_chunks(slot.data(), slot.size());
        for (double s : slot) *v = std::max(*v, s);
        return rc;
    }
    int row_stats_and_gram();
    int row_sums(double *sr, double *sr2, double *varu);
    int setup(const hb_bayes_args *args);
    int step();
    int finish(hb_bayes_out *o);
};

// ---- row-sharded exact mode (hb_bayes_args.shard_rows) ----
// Marker statistics and Gram blocks are sums over individuals: the shards' integer sums are added up (exact in doubles), so
// xpx, vx and every Gram entry are the single-GPU numbers on every rank.
int hb_run::row_stats_and_gram()
{
    int rc = hb_ctx_marker_stats(c, nullptr, nullptr, nullptr, nullptr); // local S1 (c->s1), S2 (c->xpx), min / max
    if (rc) return rc;
    std::vector<double> s12((size_t)2 * m);
    HB_HIP(hipMemcpy(s12.data(), c->s1, sizeof(double) * m, hipMemcpyDeviceToHost));
    HB_HIP(hipMemcpy(s12.data() + m, c->xpx, sizeof(double) * m, hipMemcpyDeviceToHost));
    rc = allreduce_chunks(s12.data(), s12.size());
    if (rc) return rc;
    std::vector<double> vxh(m);
    const double N = (double)n_glob;
    for (int j = 0; j < m; j++) { // src/Bayes.cpp:310-317 from the exact integer sums: vx = (N S2 - S1^2) / (N (N - 1))
        const double num = N * s12[m + j] - s12[j] * s12[j];
        vxh[j] = (num == 0 || n_glob < 2) ? 0.0 : num / (N * (N - 1));
    }
    HB_HIP(hipMemcpy(c->xpx, s12.data() + m, sizeof(double) * m, hipMemcpyHostToDevice));
    HB_HIP(hipMemcpy(c->vx, vxh.data(), sizeof(double) * m, hipMemcpyHostToDevice));
    sumvx = arma_sum(vxh.data(), vxh.size());
    nvar0 = 0;
    for (double v : vxh) nvar0 += (v == 0.0);
    double lo = std::abs((double)c->xmin), hi = std::abs((double)c->xmax), neg = c->xmin < 0 ? 1.0 : 0.0;
    rc = allreduce_max(&lo);
    if (!rc) rc = allreduce_max(&hi);
    if (!rc) rc = allreduce_max(&neg);
    if (rc) return rc;
    c->xmax = (int)std::max(lo, hi); // (only max |x| matters from here on: the bound on max |yadj|, the Gram range check)
    c->xmin = neg != 0.0 ? -c->xmax : 0;
    rc = hb_ctx_build_gram(c, &gram_seconds);
    if (rc) return rc;
    const size_t cnt = (size_t)c->m_pad * c->P * (c->Lg + 1);
    std::vector<int32_t> gi(cnt);
    HB_HIP(hipMemcpy(gi.data(), c->gram, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));
    std::vector<double> gd(cnt);
    for (size_t i = 0; i < cnt; i++) gd[i] = (double)gi[i];
    rc = allreduce_chunks(gd.data(), cnt);
    if (rc) return rc;
    for (size_t i = 0; i < cnt; i++) {
        if (std::fabs(gd[i]) >= 2147483647.0) return hb_fail(HB_ERR_UNSUPPORTED, "genotype codes too large for the exact int32 Gram matrix at this n");
        gi[i] = (int32_t)gd[i];
    }
    HB_HIP(hipMemcpy(c->gram, gi.data(), sizeof(int32_t) * cnt, hipMemcpyHostToDevice));
    return HB_OK;
}

// sum(yadj), yadj.yadj, var(u) over ALL individuals, the same numbers on every rank AND for every number of shards: partial
// sums of fixed 256-row chunks (sequential inside a chunk), gathered over the ranks, added in chunk order.
int hb_run::row_sums(double *sr, double *sr2, double *varu)
{
    std::vector<double> r(n), u(n);
    int rc = hb_ctx_get_residual(c, r.data(), u.data());
    if (rc) return rc;
    const size_t nch = (size_t)((n_glob + 255) / 256), c0 = (size_t)(row_off / 256);
    std::vector<double> part(3 * nch, 0.0);
    for (int i = 0; i < n; i++) {
        const size_t ch = c0 + (size_t)i / 256;
        part[ch] += r[i];
        part[nch + ch] = std::fma(r[i], r[i], part[nch + ch]);
        part[2 * nch + ch] += u[i];
    }
    rc = allreduce_chunks(part.data(), part.size());
    if (rc) return rc;
    double a = 0, b = 0, su = 0;
    for (size_t k = 0; k < nch; k++) { a += part[k]; b += part[nch + k]; su += part[2 * nch + k]; }
    *sr = a;
    *sr2 = b;
    const double mean = su / (double)n_glob;
    std::vector<double> p2(2 * nch, 0.0);
    for (int i = 0; i < n; i++) {
        const size_t ch = c0 + (size_t)i / 256;
        const double d = mean - u[i];
        p2[ch] = std::fma(d, d, p2[ch]);
        p2[nch + ch] += d;
    }
    rc = allreduce_chunks(p2.data(), p2.size());
    if (rc) return rc;
    double a2 = 0, a3 = 0;
    for (size_t k = 0; k < nch; k++) { a2 += p2[k]; a3 += p2[nch + k]; }
    *varu = n_glob > 1 ? (a2 - a3 * a3 / (double)n_glob) / (double)(n_glob - 1) : 0.0; // arma::var, two-pass, N - 1
    return HB_OK;
}

int hb_run::setup(const hb_bayes_args *args)
{
    const auto t0 = clk::now();
    a = *args;
    n = a.n;
    m = a.m;
    if (n < 2 || m < 1 || !a.y) return hb_fail(HB_ERR_INVALID, "Number of individuals not equals.");
    if (!a.model) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: model is NULL");
    model = a.model;
    a.model = model.c_str();
    y.assign(a.y, a.y + n);

    // ---- validation, same order and texts as src/Bayes.cpp:92-117 ----
    for (int i = 0; i < n; i++)
        if (std::isnan(y[i])) return hb_fail(HB_ERR_INVALID, "NAs are not allowed in y.");
    const bool have_x = a.X_f64 || a.X_i8;
    if (a.X_f64 && a.X_i8) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: give X_f64 or X_i8, not both");
    if (!have_x && !a.ctx) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: no genotype matrix (X_f64, X_i8 or a loaded ctx)");
    if (have_x && a.ctx) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: a pre-loaded ctx excludes X_f64 / X_i8");
    if ((a.X_f64 && a.ld_f64 < n) || (a.X_i8 && a.ld_i8 < n)) return hb_fail(HB_ERR_INVALID, "Number of individuals not equals.");
    if (a.ctx && (a.ctx->n != n || a.ctx->m != m)) return hb_fail(HB_ERR_INVALID, "Number of individuals not equals.");
    model_index = model == "BayesRR" ? 1 : model == "BayesA" ? 2 : (model == "BayesB" || model == "BayesBpi") ? 3
                : (model == "BayesC" || model == "BayesCpi" || model == "BSLMM") ? 4 : model == "BayesL" ? 5 : 6;
    fixpi = (model == "BayesB" || model == "BayesC");
    if (a.n_pi < 2 || !a.Pi) return hb_fail(HB_ERR_INVALID, "Pi should be a vector.");
    if (a.n_pi > HB_MAX_FOLD) return hb_fail(HB_ERR_UNSUPPORTED, "more mixture classes than HB_MAX_FOLD");
    Pi.assign(a.Pi, a.Pi + a.n_pi);
    n_pi = a.n_pi;
    if (arma_sum(Pi.data(), Pi.size()) != 1) return hb_fail(HB_ERR_INVALID, "sum of Pi should be 1.");
    if (Pi[0] == 1) return hb_fail(HB_ERR_INVALID, "all markers have no effect size.");
    for (double p : Pi)
        if (p < 0 || p > 1) return hb_fail(HB_ERR_INVALID, "elements of Pi should be at the range of [0, 1]");
    if (a.fold) fold_.assign(a.fold, a.fold + a.n_fold);
    else {
        if (model == "BayesR") return hb_fail(HB_ERR_INVALID, "'fold' should be provided for BayesR model.");
        fold_.assign(2, 0.0);
    }
    if ((int)fold_.size() != n_pi) return hb_fail(HB_ERR_INVALID, "length of Pi and fold not equals.");
    n_fold = (int)fold_.size();
    if (a.Ki || a.Kival || model == "BSLMM") return hb_fail(HB_ERR_UNSUPPORTED, "BSLMM (Ki/Kival) is not part of the GPU path");
    if (a.epsl_index || a.epsl_Gi || a.epsl_y_J)
        return hb_fail(HB_ERR_UNSUPPORTED, "the single-step epsilon block is not part of the GPU path");

    world = a.world > 1 ? a.world : 1;
    if (a.comm) {
        world = hb_comm_world(a.comm);
        a.rank = hb_comm_rank(a.comm);
    }
    rowmode = a.shard_rows != 0;
    sharded = !rowmode && (world > 1 || a.comm != nullptr); // (a one-rank communicator still runs the exchange path)
    n_glob = rowmode ? a.n_global : n;
    row_off = rowmode ? a.row_offset : 0;
    if (rowmode) {
        if (!a.allreduce && !a.comm && world > 1) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: shard_rows needs a communicator (comm or allreduce)");
        if (n_glob < n || row_off < 0 || row_off + n > n_glob || row_off % 256 || (row_off + n < n_glob && n % 256))
            return hb_fail(HB_ERR_INVALID, "hb_bayes_run: shard_rows needs row_offset and every shard but the last in multiples of 256 individuals");
        if (a.precise != 2) return hb_fail(HB_ERR_UNSUPPORTED, "shard_rows needs the exact fixed-point mat-vec (precise = 2): only integer sums are order-independent");
        if ((a.C && a.nc) || (a.R && a.nr)) return hb_fail(HB_ERR_UNSUPPORTED, "shard_rows: covariates and random effects are not part of the cross-check mode");
        if (a.genotype_bits == 2) return hb_fail(HB_ERR_UNSUPPORTED, "shard_rows runs on the int8 layout");
    }
    sync_blocks = std::max(1, std::min(64, (int)a.sync_blocks));
    if (const char *e = getenv("HB_RECOVER")) recover_on = atoi(e) != 0;
    const int wide_lv = (getenv("HB_WIDE_LV") && atoi(getenv("HB_WIDE_LV")) == 3) ? 3 : 2; // look-ahead groups of the point-mass models' wide geometry
    m_global = (!rowmode && (world > 1 || a.m_global > 0)) ? a.m_global : m;
    if (!rowmode && world > 1 && ((!a.allreduce && !a.comm) || m_global < m))
        return hb_fail(HB_ERR_INVALID, "hb_bayes_run: sharded run needs a communicator (comm or allreduce) and m_global");
    if (m_global < m) m_global = m;

    // ---- sizes, :119-124 ----
    vary = var_n1(y.data(), n); // (row-sharded mode: replaced by the variance over all shards once the exchange is up)
    const double h2 = 0.5;
    niter = a.niter;
    nburn = a.nburn;
    thin = a.thin;
    if (thin < 1) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: thin must be >= 1");
    n_records = std::max(0, (niter - nburn) / thin);

    // ---- covariates, :126-147 ----
    nc = a.C ? a.nc : 0;
    beta.assign(nc, 0.0);
    cpc.assign(nc, 0.0);
    beta_sum.assign(nc, 0.0);
    if (nc) {
        Cmat.assign(a.C, a.C + (size_t)n * nc);
        for (double v : Cmat)
            if (std::isnan(v)) return hb_fail(HB_ERR_INVALID, "Individuals with phenotypic value should not have missing covariates.");
        for (int i = 0; i < nc; i++) {
            const double *ci = Cmat.data() + (size_t)i * n;
            double s = 0;
            for (int k = 0; k < n; k++) s += ci[k] * ci[k];
            cpc[i] = s;
        }
    }

    // ---- environmental random effects, :149-201 with makeZ :29-57 ----
    nr = a.R ? a.nr : 0;
    dfr = a.has_dfvr ? a.dfvr : -1;
    s2r = a.has_s2vr ? a.s2vr : 0;
    vr.assign(nr, 0.0);
    vrtmp.assign(nr, vary * (1 - h2) / (nr + 1));
    vr_sum.assign(nr, 0.0);
    zid.assign((size_t)n * nr, 0);
    nlev.assign(nr, 0);
    lev_first.assign(nr, 0);
    for (int t = 0; t < nr; t++) {
        std::vector<std::string> vals(n);
        for (int k = 0; k < n; k++) {
            const char *s = a.R[(size_t)t * n + k];
            if (!s) return hb_fail(HB_ERR_INVALID, "Individuals with phenotypic value should not have missing environmental random effects.");
            vals[k] = s;
        }
        std::vector<std::string> lev(vals);
        std::stable_sort(lev.begin(), lev.end());
        lev.erase(std::unique(lev.begin(), lev.end()), lev.end());
        if ((int)lev.size() == n) return hb_fail(HB_ERR_INVALID, "number of class of environmental random effects should be less than population size.");
        if (lev.size() == 1) return hb_fail(HB_ERR_INVALID, "number of class of environmental random effects should be bigger than 1.");
        std::map<std::string, int> idx;
        for (size_t q = 0; q < lev.size(); q++) idx[lev[q]] = (int)q;
        lev_first[t] = n_levels;
        nlev[t] = (int)lev.size();
        zz.resize(n_levels + lev.size(), 0.0);
        for (int k = 0; k < n; k++) {
            const int q = idx[vals[k]];
            zid[(size_t)t * n + k] = q;
            zz[n_levels + q] += 1.0;
        }
        n_levels += (int)lev.size();
    }
    a.R = nullptr; // consumed
    estR.assign(n_levels, 0.0);
    estR_sum.assign(n_levels, 0.0);

    // ---- :288-296 ----
    always_in = (model_index == 1 || model_index == 2 || model_index == 5);
    if (always_in) {
        Pi[0] = 0;
        Pi[1] = 1;
        fixpi = true;
    } else if (model != "BayesR" && n_pi != 2) {
        return hb_fail(HB_ERR_INVALID, "length of Pi should be 2, the first value is the proportion of non-effect markers.");
    }
    dfvara_ = a.has_dfvg ? a.dfvg : 4; // :319-326
    if (dfvara_ <= 2) return hb_fail(HB_ERR_INVALID, "dfvg should not be less than 2.");
    if (niter < nburn) return hb_fail(HB_ERR_INVALID, "Number of total iteration ('niter') shold be larger than burn-in ('nburn').");
    // BayesR: the device evaluates the class boundaries as nested thresholds on q = rhs^2, which needs the non-null classes in
    // order of increasing variance (P(class <= c | q) is then decreasing in q for every c). The reference takes `fold` in any
    // order (src/Bayes.cpp:743-815) and walks the classes as given (:773-781) — with another order the same uniform picks another
    // class, so no formulation can be that walk draw for draw AND monotone. The run is therefore the reference's chain for the
    // classes SORTED by fold (the same posterior: the mixture does not depend on how its components are numbered); cls_of[]
    // maps the internal class index back to the caller's for everything reported: pi, MCMCsamples$pi, the progress line.
    cls_of.resize(n_fold);
    for (int k = 0; k < n_fold; k++) cls_of[k] = k;
    if (model_index == 6) {
        std::stable_sort(cls_of.begin() + 1, cls_of.end(), [&](int x, int z) { return fold_[x] < fold_[z]; }); // class 0 is the null class (:759)
        std::vector<double> f2(n_fold), p2(n_fold);
        for (int k = 0; k < n_fold; k++) { f2[k] = fold_[cls_of[k]]; p2[k] = Pi[cls_of[k]]; }
        fold_ = f2;
        Pi = p2;
        for (int k = 2; k < n_fold; k++)
            if (!(fold_[k] > fold_[k - 1]))
                return hb_fail(HB_ERR_UNSUPPORTED, "BayesR on the GPU path needs distinct 'fold' values for the non-null classes");
    }
    if (a.windindx) wind.assign(a.windindx, a.windindx + m);
    if (a.g_init) {
        g_init.assign(a.g_init, a.g_init + m);
        for (double v : g_init)
            if (!std::isfinite(v)) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: g_init must be finite");
        a.g_init = nullptr; // consumed
    }
    if (a.warm) {
        has_warm = true;
        warm_ = *a.warm;
        // (advisor finding, round 5) hb_warm_state carries mu, vare, varg, pi, lambda2 and the per-marker variances — NOT the fixed-effect or
        // random-effect state (beta, the level effects, Vr): with C or R the run would restart those at their defaults beside warm marker
        // effects, which is not a continuation of the same chain. Refused rather than silently half-continued.
        if (a.nc > 0 || a.nr > 0)
            return hb_fail(HB_ERR_UNSUPPORTED, "hb_bayes_run: a warm start (hb_warm_state) does not carry the fixed / random effect state: not available with C or R");
        if (!(std::isfinite(warm_.mu) && warm_.vare > 0 && std::isfinite(warm_.vare)))
            return hb_fail(HB_ERR_INVALID, "hb_bayes_run: warm state needs a finite mu and vare > 0");
        if ((model_index == 1 || model_index == 4 || model_index == 6) && !(warm_.varg > 0 && std::isfinite(warm_.varg)))
            return hb_fail(HB_ERR_INVALID, "hb_bayes_run: warm state needs varg > 0 for this model");
        if (model_index == 5 && !(warm_.lambda2 > 0 && std::isfinite(warm_.lambda2)))
            return hb_fail(HB_ERR_INVALID, "hb_bayes_run: warm state needs lambda2 > 0 for BayesL");
        if (!always_in && !fixpi) {
            double sp = 0;
            for (int j = 0; j < n_fold; j++) {
                if (!(warm_.pi[j] > 0 && warm_.pi[j] < 1)) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: warm state needs every pi in (0, 1)");
                sp += warm_.pi[j];
            }
            if (std::fabs(sp - 1.0) > 1e-9) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: warm pi must sum to 1");
        }
        if (model_index == 5 && warm_.vargL) {
            warm_vargL.assign(warm_.vargL, warm_.vargL + m);
            for (double v : warm_vargL)
                if (!(v >= 0 && std::isfinite(v))) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: warm vargL must be finite and >= 0");
        }
        warm_.vargL = nullptr;
        a.warm = nullptr; // consumed
    }

    // =========================== device set-up ===========================
    int rc;
    if (a.ctx) {
        c = a.ctx;
        own_ctx = false;
        c->seed = a.seed;
        if (sharded) c->m_offset = a.m_offset; // (a single-process run keeps the context's own marker addressing)
        c->precise = a.precise;
        c->graph_model = -1;
    } else {
        hb_ctx_params cp{};
        cp.device = a.device;
        cp.n = n;
        cp.m = m;
        cp.panel = a.panel;
        // every marker moves: panels of 512 run k_chain_dense (hb_chain_dense.hpp: static order, the band folded by other compute
        // units); small problems keep small panels, whose Gram rows are all LDS-resident in k_chain_persist
        if (!cp.panel && always_in) cp.panel = m >= 4096 ? 512 : (m >= 128 ? 128 : 64);
        cp.precise = a.precise;
        cp.m_offset = sharded ? a.m_offset : 0;
        cp.seed = a.seed;
        rc = hb_ctx_create(&cp, &c);
        if (rc) return rc;
        own_ctx = true;
        // few markers move per sweep in the point-mass models: long look-ahead, big mat-vec launches; where many or
        // all markers move the forward corrections dominate: one panel per launch, two groups of look-ahead (with one, the
        // chain idles for an update + launch boundary per panel)
        if (rowmode) rc = hb_ctx_set_pipeline(c, 0, 0, 1); // per-panel kernels: an exchange sits between each mat-vec and its chain
        else if (model_index == 3 || model_index == 4) // (2, 7): seven panels per launch, two groups of look-ahead. (Round 5 ran three on the 2-bit layout — 2 % faster then; with
            rc = hb_ctx_set_pipeline(c, 1, wide_lv, 7);   // round 6's chain it is 2.4 % SLOWER, 445-449 against 456-462 sweeps/s, and its band is 28 blocks instead of 21: HB_WIDE_LV=3 brings it back)
        else if (always_in && c->P == 512) // k_chain_dense: two panels per launch (45.6 against 39.4 sweeps/s at n=50k, m=500k; (1, 1) 24.8, (1, 2) 27.7)
            rc = hb_ctx_set_pipeline(c, 1, 2, 2);
        else if (model_index == 6 && n_fold <= 4 && c->P == 512 && !getenv("HB_NO_ADAPTIVE_R")) // BayesR: (2, 2) stored, (2, 1) while many markers move (geometry by regime, below)
            rc = hb_ctx_set_pipeline(c, 1, 2, 2);
        else rc = hb_ctx_set_pipeline(c, 1, 2, 1); // (BayesR with more classes; RR / A / L on small panels: the second group of look-ahead hides the update + launch boundary)
        if (rc) return rc;
        if (a.X_i8) rc = hb_ctx_upload_genotype_i8(c, a.X_i8, a.ld_i8, 0, m);
        else rc = hb_ctx_upload_genotype_f64(c, a.X_f64, a.ld_f64, 0, m);
        if (rc) return rc;
    }
    a.X_i8 = nullptr;
    a.X_f64 = nullptr;
    HB_HIP(hipSetDevice(c->device));

    // (row-sharded mode: the shards hold different numbers of individuals, the message must not depend on n)
    xcount = rowmode ? hb_exchange_count(4096) : hb_exchange_count(n);
    if (sharded || rowmode) {
        if (a.exchange_buf) xbuf = static_cast<double *>(a.exchange_buf);
        else {
            HB_HIP(hipMalloc(reinterpret_cast<void **>(&xbuf_own), sizeof(double) * xcount));
            xbuf = xbuf_own;
        }
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&r0), sizeof(double) * n));
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&u0), sizeof(double) * n));
    }
    if (rowmode) {
        c->row_reduce = &hb_run::row_hook;
        c->row_user = this;
        c->row_rank = a.rank;
        c->row_world = world;
        c->graph_model = -1;
        if (c->pipeline) { // (a pre-loaded context: the exchange needs the per-panel kernels)
            rc = hb_ctx_set_pipeline(c, 0, 0, 1);
            if (rc) return rc;
        }
        // var(y) over all shards, two-pass (arma::var), in the shard-count-independent order of row_sums(): partial sums of
        // fixed 256-row chunks, gathered, added in chunk order
        const size_t nch = (size_t)((n_glob + 255) / 256), c0 = (size_t)(row_off / 256);
        std::vector<double> p1(nch, 0.0);
        for (int i = 0; i < n; i++) p1[c0 + (size_t)i / 256] += y[i];
        rc = allreduce_chunks(p1.data(), p1.size());
        if (rc) return rc;
        double tot = 0;
        for (double v : p1) tot += v;
        ymean_glob = tot / (double)n_glob;
        std::vector<double> p2(2 * nch, 0.0);
        for (int i = 0; i < n; i++) {
            const size_t ch = c0 + (size_t)i / 256;
            const double t = ymean_glob - y[i];
            p2[ch] += t * t;
            p2[nch + ch] += t;
        }
        rc = allreduce_chunks(p2.data(), p2.size());
        if (rc) return rc;
        double a2 = 0, a3 = 0;
        for (size_t k = 0; k < nch; k++) { a2 += p2[k]; a3 += p2[nch + k]; }
        vary = n_glob > 1 ? (a2 - a3 * a3 / (double)n_glob) / (double)(n_glob - 1) : 0.0;
    }

    // ---- marker statistics, :310-317 ----
    std::vector<double> vx_host(g_init.empty() ? 0 : m);
    if (rowmode) {
        rc = row_stats_and_gram();
        if (rc) return rc;
        if (!g_init.empty()) HB_HIP(hipMemcpy(vx_host.data(), c->vx, sizeof(double) * m, hipMemcpyDeviceToHost));
    } else
    rc = hb_ctx_marker_stats(c, nullptr, g_init.empty() ? nullptr : vx_host.data(), &sumvx, &nvar0);
    if (rc) return rc;
    if (sharded) { // marker shards: the statistics of all markers (row-sharded mode already holds the global ones)
        double sv[2] = {sumvx, (double)nvar0};
        rc = allreduce_host(sv, 2);
        if (rc) return rc;
        sumvx = sv[0];
        nvar0 = (int)sv[1];
    }
    if (sharded && world > 1) {
        // every rank must take the same replay decision (an aborting rank poisons the exchanged sums; a rank that would not replay
        // would fail while the others re-enter the all-reduce and hang): recovery is on only where ALL ranks have it — the
        // persistent pipeline available (the concurrency probe is per process) and HB_RECOVER not switched off
        double ok[1] = {(recover_on && c->pipeline) ? 1.0 : 0.0};
        rc = allreduce_host(ok, 1);
        if (rc) return rc;
        if (ok[0] < (double)world) {
            if (recover_on && c->pipeline)
                fprintf(stderr, "hibayes_gpu: rank %d: sweep replay switched off — not every rank can replay (HB_RECOVER / pipeline availability differ)\n", a.rank);
            recover_on = false;
        }
        pipeline_setup = c->pipeline ? 1 : 0;
    }
    // ---- resident layout (round 6): genotype_bits = 0 is "auto" — 2 bits per genotype where that is exact AND the faster sweep: codes
    // 0..3 (PLINK's own alphabet, src/read_bed.cpp:116-120), the fixed-point mat-vec, and the point-mass models' wide launches
    // (BayesB / BayesC at panel 512: 450 against 213 sweeps/s at n = 50k, m = 500k; the same chain bit for bit). The models whose
    // launches cover one or two panels are bound by their chain workgroup and run the lighter int8 kernel beside it. 8 forces int8. ----
    int bits_run = a.genotype_bits == 2 ? 2 : 8;
    const bool sparse_bc = model_index == 3 || model_index == 4, mix_r = model_index == 6 && n_fold <= 4;
    if (a.genotype_bits == 0 && own_ctx && !rowmode && a.precise == 2 && (sparse_bc || mix_r) && c->P == 512 &&
        c->pipeline && c->xmin >= 0 && c->xmax <= 3 && !getenv("HB_NO_AUTO_BITS")) {
        // (BayesR, measured late in round 6 with k_dotq2m beside both of its chains: 64.5 against 57.8 sweeps/s 300 sweeps after a cold start, 102.8 against 98.3
        // converged — a quarter of the genotype bytes streaming past the chain workgroup's own round trips; round 4's "the 2-bit kernel only lengthens the
        // launches" was the v_dot4 kernel)
        // the band of the geometry — (2, 7): 21 blocks ((3, 7): 28), BayesR's (2, 2): 6 — and the packed genotypes must fit beside the int8 columns the band is built from
        size_t fr = 0, tot = 0;
        const size_t band = (size_t)(sparse_bc ? 7 * (wide_lv + 1) : 6) * c->m_pad * c->P * sizeof(int32_t), x2 = (size_t)((c->ld + 511) / 512 * 128) * c->m_pad;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > band + x2 + ((size_t)2 << 30)) bits_run = 2;
        else (void)hipGetLastError();
    }
    if (bits_run == 2 && own_ctx && a.genotype_bits == 0 && sparse_bc && wide_lv != 2) {
        rc = hb_ctx_set_pipeline(c, 1, wide_lv, 7); // (HB_WIDE_LV=3: round 5's third group of look-ahead on the 2-bit layout)
        if (rc) return rc;
    }
    if (!c->gram_ready) {
        rc = hb_ctx_build_gram(c, &gram_seconds);
        if (rc) return rc;
    }
    if (a.genotype_bits != 0 && a.genotype_bits != 8 && a.genotype_bits != 2)
        return hb_fail(HB_ERR_INVALID, "hb_bayes_run: genotype_bits must be 0 (auto), 8 or 2");
    if (bits_run == 2 && own_ctx) { // the Gram blocks (built from the int8 columns) are in place: pack, drop the int8 copy
        rc = hb_ctx_set_layout(c, 2, 0);
        if (rc) return rc;
    }
    if (!own_ctx && c->adaptive && c->pipeline && c->home_d > 0 && (c->home_lv != c->Lv || c->home_d != c->D) && !rowmode) {
        // an adaptive context that an earlier run left in its narrow geometry: start from the geometry its owner set (round 6: a second fit
        // on the same context used to stay narrow for good, because only the wide geometry switches adaptivity on)
        rc = hb_ctx_switch_geometry(c, 1, c->home_lv, c->home_d);
        if (rc) return rc;
    }
    {   // geometry by regime: only from the wide-band geometry of the point-mass models, whose stored band serves the narrow one
        int32_t gp = 0, gl = 0, gd = 0, gb = 0;
        (void)hb_ctx_get_pipeline(c, &gp, &gl, &gd, &gb);
        adaptive_geo = (model_index == 3 || model_index == 4) && (own_ctx || c->adaptive) && gp == 1 && (((gl == 2 || gl == 3) && gd == 7) || (gl == 2 && gd == 8)) && c->Lg >= 20;
        geo_wide_lv = gl;
        if (adaptive_geo) geo_wide_d = gd;
        geo_cur = 0;
        // round 6, re-measured at n = 50k, m = 500k from a cold start with this round's chains (profiles/r06_regime_bayescpi*.txt; round 3's 2.0 / 2.6 were taken when
        // the wide geometry ran 166 sweeps/s): 2-bit genotypes — at 3.6 moves a panel (2, 2) 148 against (2, 7) 141 sweeps/s, at 2.6: 173 against 189, at 2.1: 187
        // against 233; int8 columns — at 5.8: 97 against 80, at 3.6: 124 against 131, at 2.6: 131 against 168
        // One pair of thresholds for both layouts (the geometries cross at 3.2 moves a panel on 2-bit genotypes, at 4.2 on int8 columns): a run on 2-bit genotypes
        // is the int8 run BIT FOR BIT only if both take the same geometry in every sweep (tests/test_gpu_depth.py test_two_bit_resident_layout_is_the_same_chain;
        // per-layout thresholds broke exactly that), and between 3.2 and 4.2 the int8 run loses 5 % for a handful of sweeps.
        if (model_index == 3 || model_index == 4) {
            geo_to_wide = 3.2;
            geo_to_narrow = 4.0;
        }
        // round 6, BayesR with up to four classes at panel 512: two panels per launch and the certified group chain (k_chain_group<3, 2, 2, 15> + k_fwd + warmers)
        // once fewer than ~22 markers a panel move, one panel per launch and the per-panel chain with its row cache (k_chain_persist) above ~27
        // (measured at n = 50k, m = 500k from a cold start, profiles/r06_bayesr_regime.txt: they cross at 19 moves per panel — 47.5 sweeps/s both;
        // at 51: 37 against 54; at 11: 68 against 60; at 8: 85 against 69)
        if (model_index == 6 && n_fold <= 4 && c->P == 512 && (own_ctx || c->adaptive) && gp == 1 && gl == 2 && gd == 2 && c->Lg >= 5 && !getenv("HB_NO_ADAPTIVE_R")) {
            adaptive_geo = true;
            geo_wide_d = 2;
            geo_narrow_lv = 2;
            geo_narrow_d = 1;
            geo_to_wide = 22.0;   // (re-measured with k_fwd and the warmers beside the group chain, profiles/r06_bayesr_regime2.txt: at 19.6 moves a panel 51.4 against 49.6
            geo_to_narrow = 27.0; //  sweeps/s, at 11: 78 against 63; at 47: 39 against 54 — no measurement in between)
        }
        if (adaptive_geo) { // the first sweep: as many moves as markers are expected in the model (a cold start) or are in it
            double nz = 0;
            for (double gv : g_init) nz += gv != 0.0;
            if (g_init.empty()) nz = (1.0 - Pi[0]) * (double)m;
            last_events_pp = nz / std::max(1, c->npanels);
        }
    }

    // ---- prior defaults, :327-374 ----
    vara_ = a.has_vg ? a.vg : ((dfvara_ - 2) / dfvara_) * vary * h2;
    vare_ = a.has_ve ? a.ve : vary * (1 - h2) / (nr + 1);
    dfvare_ = a.has_dfve ? a.dfve : -2;
    s2vara_ = a.has_s2vg ? a.s2vg : vara_ * (dfvara_ - 2) / dfvara_;
    varg = vara_ / ((1 - Pi[0]) * sumvx);
    s2varg_ = s2vara_ / ((1 - Pi[0]) * sumvx);
    s2vare_ = a.has_s2ve ? a.s2ve : 0;
    const double R2 = (dfvara_ - 2) / dfvara_;
    lambda2 = 2 * (1 - R2) / (R2)*sumvx;
    lambda = std::sqrt(lambda2);
    rate0 = (shape0 - 1) / lambda2;
    vara_fold.assign(n_fold, 0.0);
    fold_snp_num.assign(n_fold, 0.0);
    pi_sum.assign(n_fold, 0.0);
    for (int j = 0; j < n_fold; j++) vara_fold[j] = (vara_ / ((1 - Pi[0]) * sumvx)) * fold_[j];
    {
        std::vector<double> g0(m, 0.0), vl(m, varg);
        std::vector<uint8_t> t0v(m, 0);
        rc = hb_ctx_set_effects(c, g0.data(), t0v.data(), warm_vargL.empty() ? vl.data() : warm_vargL.data()); // vargL.fill(varg), :364-368
        if (rc) return rc;
        HB_HIP(hipMemsetAsync(c->nzrate, 0, sizeof(uint32_t) * (size_t)c->m_pad, c->stream));
        HB_HIP(hipMemsetAsync(c->alpha_sum, 0, sizeof(double) * (size_t)c->m_pad, c->stream));
        HB_HIP(hipMemsetAsync(c->alpha_sq, 0, sizeof(double) * (size_t)c->m_pad, c->stream));
    }
    if (!wind.empty()) {
        for (int i = 0; i < m; i++) nw = std::max(nw, (int)wind[i]);
        if (world > 1) { // window ids are global: every rank needs the same nw (max over ranks)
            if ((size_t)world > xcount) return hb_fail(HB_ERR_INVALID, "too many ranks for the exchange buffer");
            std::vector<double> slots(world, 0.0);
            slots[a.rank] = nw;
            rc = allreduce_host(slots.data(), world);
            if (rc) return rc;
            for (double s : slots) nw = std::max(nw, (int)s);
        }
        rc = hb_ctx_set_windows(c, wind.data(), nw);
        if (rc) return rc;
    } else {
        rc = hb_ctx_set_windows(c, nullptr, 0);
        if (rc) return rc;
    }
    rc = hb_ctx_set_covariates(c, nc ? Cmat.data() : nullptr, nc);
    if (rc) return rc;
    rc = hb_ctx_set_levels(c, nr ? zid.data() : nullptr, nr, nr ? nlev.data() : nullptr);
    if (rc) return rc;
    rc = hb_ctx_blocks_setup(c, cpc.data(), zz.data(), vrtmp.data());
    if (rc) return rc;

    // ---- console, :393-461 ----
    line("Prior parameters:");
    line("    Model fitted at [%s]", model == "BayesRR" ? "Bayes Ridge Regression" : model.c_str());
    line("    Number of observations %d", n);
    line("    Number of covariates %d", nc + 1);
    line("    Number of envir-random effects %d", nr);
    line("    Number of markers %lld", (long long)m_global);
    line("    Total number of iteration %d", niter);
    line("    Total number of burn-in %d", nburn);
    line("    Frequency of collecting %d", thin);
    line("    Phenotypic var %f", vary);
    line("    Genetic var %f", vara_);
    line("    Inv-Chisq gpar %f %f", dfvara_, s2vara_);
    line("    Residual var %f", vare_);
    line("    Inv-Chisq epar %f %f", dfvare_, s2vare_);
    line("    Marker var %f", varg);
    line("    Inv-Chisq alpar %f %f", dfvara_, s2varg_);
    if (nw) line("    Number of windows for GWAS analysis %d", nw);
    if (world > 1 && always_in)
        line("    WARNING: %d marker shards with a model in which every marker moves every sweep (%s): shards are stale with "
             "respect to each other inside a sweep, variance components are biased (DESIGN.md section 8)", world, model.c_str());
    line("MCMC started: ");
    line(" Iter  NumNZSnp  pi  %sVg  Ve  h2  Timeleft", model == "BayesL" ? "Lambda  " : "");

    // ---- warm state: the scalars of a chain that is continued (the constants above stay the cold run's) ----
    if (has_warm) {
        vare_ = warm_.vare;
        if (model_index == 1 || model_index == 4 || model_index == 6) varg = warm_.varg;
        if (model_index == 6)
            for (int j = 0; j < n_fold; j++) vara_fold[j] = varg * fold_[j]; // :808
        if (model_index == 5) {
            lambda2 = warm_.lambda2;
            lambda = std::sqrt(lambda2);
        }
        if (!always_in && !fixpi)
            for (int j = 0; j < n_fold; j++) Pi[j] = warm_.pi[cls_of[j]]; // (internal class j is the caller's cls_of[j])
    }

    // ---- :469-472 ----
    mu = has_warm ? warm_.mu : (rowmode ? ymean_glob : arma_sum(y.data(), n) / n);
    {
        std::vector<double> yadj(n), zero(n, 0.0);
        for (int i = 0; i < n; i++) yadj[i] = y[i] - mu;
        if (!g_init.empty()) {
            // warm start: monomorphic markers are never visited by the sweep (:589), so they start (and stay) at zero
            std::vector<uint8_t> trk(m, 0);
            for (int i = 0; i < m; i++) {
                if (vx_host[i] == 0.0) g_init[i] = 0.0;
                trk[i] = g_init[i] != 0.0;
            }
            rc = hb_ctx_set_effects(c, g_init.data(), trk.data(), nullptr);
            if (rc) return rc;
            rc = hb_ctx_matvec(c, g_init.data(), zero.data()); // u = X g
            if (rc) return rc;
            if (sharded) { // every rank starts from the residual of ALL shards' effects (u and yadj are replicated)
                rc = allreduce_host(zero.data(), n);
                if (rc) return rc;
            }
            for (int i = 0; i < n; i++) yadj[i] -= zero[i];
        }
        rc = hb_ctx_set_residual(c, yadj.data(), zero.data());
        if (rc) return rc;
    }
    if (rowmode) {
        double vu = 0;
        rc = row_sums(&sum_r, &sum_r2, &vu);
    } else {
        rc = hb_ctx_residual_sums(c, &sum_r, &sum_r2);
    }
    if (rc) return rc;
    NnzSnp = always_in ? m_global : 0;
    s_mu.assign(n_records, 0.0);
    s_Vg.assign(n_records, 0.0);
    s_Ve.assign(n_records, 0.0);
    s_h2.assign(n_records, 0.0);
    s_pi.assign((size_t)n_records * n_fold, 0.0);
    s_beta.assign((size_t)n_records * nc, 0.0);
    s_Vr.assign((size_t)n_records * nr, 0.0);
    s_r.assign((size_t)n_records * n_levels, 0.0);
    if (a.store_alpha) s_alpha.assign((size_t)n_records * m, 0.0);
    setup_seconds = std::chrono::duration<double>(clk::now() - t0).count();
    return HB_OK;
}

// one iteration of the loop at src/Bayes.cpp:477-917
int hb_run::step()
{
    if (done || iter >= niter) {
        done = true;
        return HB_OK;
    }
    const auto t0 = clk::now();
    int rc;
    HB_HIP(hipSetDevice(c->device));
    if (a.interrupt && a.interrupt(a.interrupt_user)) return hb_fail(HB_ERR_INTERRUPT, "interrupted");
    hb_stream hs(a.seed, hb_sub(HB_PURPOSE_HOST, (uint64_t)iter), 0);

    // sample intercept, :479-482
    const double ng = (double)n_glob; // (= n unless the individuals are sharded)
    const double mu_ = -(sum_r / ng + std::sqrt(vare_ / ng) * hs.norm());
    mu -= mu_;
    rc = hb_ctx_residual_shift(c, mu_);
    if (rc) return rc;

    // covariates (:484-494) and environmental random effects (:496-516): enqueued as device kernels. Their deviates do not
    // depend on the data, so they are drawn here first, in the reference's order — one normal per covariate, then per term
    // its level normals followed by its chisq — and the state comes back with the sweep's fetch: one host sync per iteration
    if (nc + nr) {
        zb_.resize(nc);
        zl_.resize(n_levels);
        ch_.resize(nr);
        for (int i = 0; i < nc; i++) zb_[i] = hs.norm();
        for (int t = 0; t < nr; t++) {
            for (int q = 0; q < nlev[t]; q++) zl_[lev_first[t] + q] = hs.norm();
            ch_[t] = hs.chisq(nlev[t] + dfr);
        }
        rc = hb_ctx_blocks_step(c, vare_, zb_.data(), zl_.data(), ch_.data(), dfr, s2r);
        if (rc) return rc;
    }

    // ---------------- marker sweep on the device, :586-816 ----------------
    hb_sweep_in in{};
    in.model_index = model_index;
    in.n_fold = n_fold;
    in.iter = iter;
    in.vare = vare_;
    in.varg = varg;
    in.s2varg_df = s2varg_ * dfvara_;
    in.dfvara = dfvara_;
    for (int j = 0; j < n_fold; j++) {
        in.logpi[j] = std::log(Pi[j]);
        in.fold[j] = fold_[j];
        in.vara_fold[j] = vara_fold[j];
    }
    in.lambda = lambda;
    in.lambda2 = lambda2;
    in.count_pip = (iter >= nburn) && !always_in;
    in.store = (iter >= nburn) && ((iter + 1 - nburn) % thin == 0);
    hb_sweep_out so{};
    // geometry by regime (point-mass models, when the context is ours or its owner asked for it): while many markers move
    // every move costs one band row per block of the band, so a narrow band wins; once few move, the wide band with its
    // big mat-vec launches does (measured at n=50k, m=500k: (2,2) 114 vs (2,7) 92 sweeps/s at 3.5 moves per panel,
    // 136 vs 166 at 1.4). The stored band serves both; each geometry's captured sweep is cached.
    if (adaptive_geo) {
        const double pp = last_events_pp; // moves per panel of the previous sweep on this shard
        int want = geo_cur;
        if (geo_cur == 1 && pp < geo_to_wide) want = 0;         // -> (2 | 3, 7); BayesR: (2, 2)
        else if (geo_cur == 0 && pp > geo_to_narrow) want = 1;  // -> (2, 2); BayesR: (2, 1)
        if (want != geo_cur) {
            rc = hb_ctx_switch_geometry(c, 1, want == 0 ? geo_wide_lv : geo_narrow_lv, want == 0 ? geo_wide_d : geo_narrow_d);
            if (rc) return rc;
            geo_cur = want;
        }
    }
    // The sweep, in sync_blocks runs of mat-vec groups (1: the whole sweep). After each run the shards sum their residual
    // deltas (n doubles; u moves by the negative) — and, after the last one, the 16 scalar sums — all enqueued on the sweep
    // stream behind the run itself; the iteration's only host synchronisation is the fetch below.
    // A sweep whose pipeline gave up waiting (HB_ERR_ABORTED: every wait is bounded) is replayed from the state
    // saved here: effects, residual, u and the posterior counters, a few MB copied by one kernel. The draws are counter-based, so the
    // replay is the same chain. A second failure of the same sweep replays it on the event-ordered per-panel kernels, which wait
    // for nothing on the device; sharded, the abort reaches every rank through the exchange and all of them replay.
    // (advisor finding, round 5) the replay decision was agreed by all ranks in setup on the context's pipeline switch as it was then; a caller that flips
    // it on one rank afterwards (hb_ctx_set_pipeline between steps) would make that rank's decision differ and the others hang in the exchange: refuse loudly.
    // (The adaptive geometry changes Lv and D, never this switch; the replay's own fallback flips it inside this function and puts it back.)
    if (pipeline_setup >= 0 && (c->pipeline ? 1 : 0) != pipeline_setup)
        return hb_fail(HB_ERR_INVALID, "the context's pipeline switch changed during a sharded run (the ranks agreed on their replay decision with the setting at set-up)");
    const bool recover = recover_on && c->pipeline && !rowmode;
    // Whatever way this iteration ends, the context gets back the wait bound and the geometry it came with (a caller-owned context
    // outlives the run): a scope guard, not a line after the loop that the error returns inside it would skip.
    struct restore_ctx {
        hb_ctx *c;
        int timeout_ms;
        int geo[4] = {0, 0, 0, 0};
        bool fell_back = false;
        ~restore_ctx()
        {
            c->timeout_ms = timeout_ms;
            if (fell_back) {
                c->force_geometry = true;
                (void)hb_ctx_switch_geometry(c, geo[0], geo[1], geo[2]);
                c->force_geometry = false;
            }
        }
    } guard{c, c->timeout_ms};
    // A sweep that is replayed when it times out may give up early: a healthy hand-off takes microseconds, the device's own pauses a
    // millisecond, so 100 ms instead of the context's 3 s — unless HB_TIMEOUT_MS fixed the bound, or the run has
    // learnt that this device is slow (three aborts within 64 iterations raise slow_timeout). Without recovery the context's bound stays.
    if (recover && !c->timeout_env) c->timeout_ms = std::max(100, slow_timeout);
    if (recover) {
        rc = hb_ctx_snapshot(c, model_index, in.store != 0, in.count_pip != 0);
        if (rc) return rc;
    }
    for (int attempt = 0;; attempt++) {
        for (int b = 0; b < sync_blocks; b++) {
            if (sharded) {
                HB_HIP(hipMemcpyAsync(r0, c->r, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
                HB_HIP(hipMemcpyAsync(u0, c->u, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
            }
            rc = hb_ctx_sweep_range(c, &in, b, sync_blocks);
            if (rc) return rc;
            if (sharded) {
                const bool last = b == sync_blocks - 1;
                rc = hbk_delta_pack(c, r0, u0, xbuf);
                if (rc) return rc;
                // (the sums ride along every time — the message has one shape — but only the last run's are the sweep's)
                HB_HIP(hipMemcpyAsync(xbuf + (size_t)n, c->acc, sizeof(double) * HB_ACC_N, hipMemcpyDeviceToDevice, c->stream));
                rc = hbk_abort_poison(c, xbuf + (size_t)n);
                if (rc) return rc;
                rc = exchange();
                if (rc) return rc;
                rc = hbk_delta_unpack(c, r0, u0, xbuf);
                if (rc) return rc;
                if (last) {
                    HB_HIP(hipMemcpyAsync(c->acc, xbuf + (size_t)n, sizeof(double) * HB_ACC_N, hipMemcpyDeviceToDevice, c->stream));
                    rc = hbk_reduce_ru(c);
                    if (rc) return rc;
                }
            }
        }
        rc = hb_ctx_sweep_end(c, &so);
        if (rc != HB_ERR_ABORTED || !recover || attempt >= 3) break;
        aborts++;
        // a second time-out of the same sweep: the event-ordered per-panel kernels, which wait for nothing on the device (the switch is
        // forced past HB_PIPELINE / HB_LOOKAHEAD / HB_DOTGROUP: a tuning variable must not turn the fall-back into a third try of the
        // pipeline that just stalled twice)
        bool per_panel = guard.fell_back;
        if (attempt >= 1 && !guard.fell_back) {
            (void)hb_ctx_get_pipeline(c, &guard.geo[0], &guard.geo[1], &guard.geo[2], &guard.geo[3]);
            c->force_geometry = true;
            const int rc2 = hb_ctx_switch_geometry(c, 0, 0, 1);
            c->force_geometry = false;
            if (rc2) return rc2;
            guard.fell_back = true;
            per_panel = c->pipeline == 0;
        }
        fprintf(stderr, "hibayes_gpu: iteration %d: %s — state restored, replaying the sweep%s\n", iter + 1, hb_last_error(),
                per_panel ? " on the per-panel kernels" : "");
        rc = hb_ctx_restore(c);
        if (rc) return rc;
        // (the replay waits 3 s: a wait that was merely slow — a shared or profiled GPU — then gets through)
        c->timeout_ms = std::max(c->timeout_ms, 3000);
        // a run that keeps aborting is on a slow device, not in a stall (those come once in a few thousand sweeps): wait longer from now on
        if (iter - abort_win_start > 64) { abort_win_start = iter; abort_win_count = 0; }
        if (++abort_win_count >= 3) slow_timeout = std::min(3000, std::max(100, slow_timeout) * 4);
    }
    if (rc) return rc;
    events_sum += so.n_events;
    last_events_pp = so.n_events / ((double)std::max(1, world) * std::max(1, c->npanels));
    miss_sum += so.n_cache_miss;
    redo_sum += so.n_redo;
    sum_r = so.sum_r;
    sum_r2 = so.sum_r2;
    if (rowmode) { // the three n-long reductions over all shards (the sweep's own ones saw this shard's rows only)
        if (c->row_failed) return hb_fail(HB_ERR_COMM, "row-sharded mode: an all-reduce inside the sweep failed");
        rc = row_sums(&sum_r, &sum_r2, &so.var_u);
        if (rc) return rc;
    }
    if (nc + nr) { // the block state arrived with the sweep's fetch
        const double *h = c->h_blk;
        std::copy(h, h + nc, beta.begin());
        std::copy(h + nc, h + nc + n_levels, estR.begin());
        std::copy(h + nc + n_levels, h + nc + n_levels + nr, vrtmp.begin());
        std::copy(h + nc + n_levels + nr, h + nc + n_levels + 2 * nr, vr.begin());
    }

    // hyper-parameters after the sweep
    auto draw_pi = [&]() { // rdirichlet_sample, src/stats.cpp:69-76
        std::vector<double> xn(n_fold);
        for (int j = 0; j < n_fold; j++) xn[j] = hs.gamma(fold_snp_num[j] + 1, 1.0);
        const double sx = arma_sum(xn.data(), xn.size());
        for (int j = 0; j < n_fold; j++) Pi[j] = xn[j] / sx;
    };
    switch (model_index) {
    case 1: // :603
        varg = (so.sum_g2 + s2varg_ * dfvara_) / hs.chisq(dfvara_ + (double)m_global - nvar0);
        break;
    case 2: break;
    case 3: // :666-669
        fold_snp_num[1] = so.class_count[1];
        fold_snp_num[0] = (double)m_global - nvar0 - fold_snp_num[1];
        NnzSnp = (long long)fold_snp_num[1];
        if (!fixpi) draw_pi();
        break;
    case 4: // :710-716
        fold_snp_num[1] = so.class_count[1];
        fold_snp_num[0] = (double)m_global - nvar0 - fold_snp_num[1];
        NnzSnp = (long long)fold_snp_num[1];
        varg = (so.sum_g2 + s2varg_ * dfvara_) / hs.chisq(dfvara_ + (double)NnzSnp);
        if (!fixpi) draw_pi();
        break;
    case 5: { // :738-741
        const double shape = shape0 + (double)m_global - nvar0;
        const double rate = rate0 + so.sum_vargL / 2;
        lambda2 = hs.gamma(shape, 1 / rate);
        lambda = std::sqrt(lambda2);
        break;
    }
    case 6: { // :803-814
        double nz = 0;
        for (int j = 0; j < n_fold; j++) fold_snp_num[j] = so.class_count[j];
        for (int j = 1; j < n_fold; j++) nz += fold_snp_num[j];
        NnzSnp = (long long)nz;
        varg = (so.sum_g2 + s2varg_ * dfvara_) / hs.chisq(dfvara_ + (double)NnzSnp);
        for (int j = 0; j < n_fold; j++) vara_fold[j] = varg * fold_[j];
        if (!fixpi) draw_pi(); // class_count[0] already excludes the nvar0 monomorphic markers (:813)
        break;
    }
    }
    vara_ = so.var_u;                                                       // :819
    vare_ = (sum_r2 + s2vare_ * dfvare_) / hs.chisq(ng + dfvare_);  // :823

    if (iter >= nburn) { // :826-845 (the per-marker counters themselves live on the device)
        nzct++;
    }

    // thinned store, :848-882
    if (in.store && count < n_records) {
        s_mu[count] = mu;
        mu_sum += mu;
        if (!fixpi)
            for (int j = 0; j < n_fold; j++) {
                s_pi[(size_t)count * n_fold + j] = Pi[j];
                pi_sum[j] += Pi[j];
            }
        s_Vg[count] = vara_;
        s_Ve[count] = vare_;
        vara_sum += vara_;
        vare_sum += vare_;
        if (a.store_alpha) {
            rc = hb_ctx_get_effects(c, s_alpha.data() + (size_t)count * m, nullptr, nullptr);
            if (rc) return rc;
        }
        for (int i = 0; i < nc; i++) {
            s_beta[(size_t)count * nc + i] = beta[i];
            beta_sum[i] += beta[i];
        }
        double vt = vara_ + vare_;
        for (int t = 0; t < nr; t++) {
            vt += vr[t];
            s_Vr[(size_t)count * nr + t] = vr[t];
            vr_sum[t] += vr[t];
        }
        for (int q = 0; q < n_levels; q++) {
            s_r[(size_t)count * n_levels + q] = estR[q];
            estR_sum[q] += estR[q];
        }
        s_h2[count] = vara_ / vt;
        hsq_sum += vara_ / vt;
        count++;
    }
    loop_seconds += std::chrono::duration<double>(clk::now() - t0).count();

    if (a.verbose && a.outfreq > 0 && (iter + 1) % a.outfreq == 0) { // :884-914
        const int tt = (int)std::floor(loop_seconds / (iter + 1) * (niter - iter));
        double vt = vara_ + vare_;
        for (int t = 0; t < nr; t++) vt += vr[t];
        char pis[256] = {0};
        size_t off = 0;
        std::vector<double> pc(n_fold);
        for (int j = 0; j < n_fold; j++) pc[cls_of[j]] = Pi[j]; // (the caller's class order)
        for (int j = 0; j < n_fold && off < sizeof(pis) - 16; j++) off += snprintf(pis + off, sizeof(pis) - off, "%.4f ", pc[j]);
        char lam[32] = {0};
        if (model == "BayesL") snprintf(lam, sizeof(lam), "%.4f ", lambda);
        line(" %d %lld %s%s%.4f %.4f %.4f %02dh%02dm%02ds", iter + 1, NnzSnp, pis, lam, vara_, vare_, vara_ / vt, tt / 3600,
             tt % 3600 / 60, tt % 3600 % 60);
    }
    iter++;
    if (count == n_records || iter >= niter) done = true; // :916
    return HB_OK;
}

// posterior assembly, src/Bayes.cpp:919-1040
int hb_run::finish(hb_bayes_out *o)
{
    int rc;
    HB_HIP(hipSetDevice(c->device));
    const double Rn = (double)n_records;
    o->n_records = n_records;
    o->n_levels = n_levels;
    o->nw = nw;
    o->Vg = vara_sum / Rn;
    o->Ve = vare_sum / Rn;
    o->h2 = hsq_sum / Rn;
    const double Mu = mu_sum / Rn;
    o->mu = Mu;
    std::vector<double> e(n), nz(m), asum(m), asq(m);
    for (int i = 0; i < n; i++) e[i] = y[i] - Mu;
    for (int i = 0; i < nc; i++) {
        const double b = beta_sum[i] / Rn;
        if (o->beta) o->beta[i] = b;
        const double *ci = Cmat.data() + (size_t)i * n;
        for (int k = 0; k < n; k++) e[k] -= b * ci[k];
    }
    rc = hb_ctx_get_counters(c, nz.data(), asum.data(), asq.data());
    if (rc) return rc;
    for (int i = 0; i < m; i++) {
        const double mean = asum[i] / Rn;
        if (o->alpha_sd) o->alpha_sd[i] = n_records > 1 ? std::sqrt(std::max(0.0, (asq[i] - Rn * mean * mean) / (Rn - 1))) : 0.0;
        asum[i] = mean;
    }
    if (o->alpha) std::memcpy(o->alpha, asum.data(), sizeof(double) * m);
    { // e -= X * alpha (:971): one device mat-vec; shards sum their partial products
        std::vector<double> xa(n);
        rc = hb_ctx_matvec(c, asum.data(), xa.data());
        if (rc) return rc;
        if (sharded) {
            rc = allreduce_host(xa.data(), n);
            if (rc) return rc;
        }
        for (int k = 0; k < n; k++) e[k] -= xa[k];
    }
    { // the state after the last iteration (hb_bayes_out.last), before Pi becomes its posterior mean below
        o->last = hb_warm_state{};
        o->last.mu = mu;
        o->last.vare = vare_;
        o->last.varg = varg;
        o->last.lambda2 = lambda2;
        for (int j = 0; j < n_fold; j++) o->last.pi[cls_of[j]] = Pi[j];
        o->last.vargL = o->vargL_last;
        if (o->g_last || o->vargL_last) {
            rc = hb_ctx_get_effects(c, o->g_last, nullptr, o->vargL_last);
            if (rc) return rc;
        }
    }
    if (!fixpi) {
        for (int j = 0; j < n_fold; j++) Pi[j] = pi_sum[j] / Rn;
    } else { // :979-983
        for (int r = 0; r < n_records; r++) {
            s_pi[(size_t)r * n_fold + 0] = Pi[0];
            s_pi[(size_t)r * n_fold + 1] = Pi[1];
        }
    }
    if (o->pi) for (int j = 0; j < n_pi; j++) o->pi[cls_of[j]] = Pi[j]; // (the caller's class order)
    if (nr) {
        for (int t = 0; t < nr; t++) {
            if (o->Vr) o->Vr[t] = vr_sum[t] / Rn;
            if (o->r_term_nlevels) o->r_term_nlevels[t] = nlev[t];
        }
        std::vector<double> est(n_levels);
        for (int q = 0; q < n_levels; q++) est[q] = estR_sum[q] / Rn;
        for (int t = 0; t < nr; t++)
            for (int k = 0; k < n; k++) e[k] -= est[lev_first[t] + zid[(size_t)t * n + k]];
        if (o->r_est) std::memcpy(o->r_est, est.data(), sizeof(double) * n_levels);
    }
    if (o->g) { rc = hb_ctx_get_residual(c, nullptr, o->g); if (rc) return rc; } // :1023, final-iteration u
    if (o->e) std::memcpy(o->e, e.data(), sizeof(double) * n);
    if (o->pip) {
        if (always_in) for (int i = 0; i < m; i++) o->pip[i] = 1.0; // :1026-1027
        else for (int i = 0; i < m; i++) {
            double p = nz[i] / nzct;
            if (p == 1) p = (nzct - 1) / (double)nzct; // :1030
            o->pip[i] = p;
        }
    }
    if (nw && o->gwas) { // :1034-1038; windows must not straddle shards (hibayes_amd/dist.py checks)
        std::vector<double> w(nw);
        rc = hb_ctx_get_windows(c, w.data());
        if (rc) return rc;
        if (sharded) { // (chunked: the exchange buffer holds n + 16 values)
            for (int k0 = 0; k0 < nw; k0 += (int)xcount) {
                rc = allreduce_host(w.data() + k0, std::min((int)xcount, nw - k0));
                if (rc) return rc;
            }
        }
        for (int k = 0; k < nw; k++) {
            double p = w[k] / nzct;
            if (p == 1) p = (nzct - 1) / (double)nzct;
            o->gwas[k] = p;
        }
    }
    o->nzct = nzct;
    auto cp = [](double *dst, const std::vector<double> &src) { if (dst && !src.empty()) std::memcpy(dst, src.data(), sizeof(double) * src.size()); };
    cp(o->s_mu, s_mu); cp(o->s_Vg, s_Vg); cp(o->s_Ve, s_Ve); cp(o->s_h2, s_h2);
    if (o->s_pi)
        for (int r = 0; r < n_records; r++)
            for (int j = 0; j < n_fold; j++) o->s_pi[(size_t)r * n_fold + cls_of[j]] = s_pi[(size_t)r * n_fold + j];
    cp(o->s_beta, s_beta); cp(o->s_Vr, s_Vr); cp(o->s_r, s_r);
    if (a.store_alpha) cp(o->s_alpha, s_alpha);
    o->setup_seconds = setup_seconds;
    o->loop_seconds = loop_seconds;
    o->iters_done = iter;
    o->mean_events = iter > 0 ? events_sum / iter : 0;
    o->sweeps_replayed = aborts;
    o->resident_bits = c ? c->layout : 0;
    line("Posterior parameters:");
    line("    Mu %f", Mu);
    line("    Genetic var %f", o->Vg);
    line("    Residual var %f", o->Ve);
    line("    Estimated h2 %f", o->h2);
    line("Finished: set-up %.2fs (Gram %.2fs), MCMC %.2fs, %.1f sweeps/s", setup_seconds, gram_seconds, loop_seconds,
         loop_seconds > 0 ? iter / loop_seconds : 0.0);
    return HB_OK;
}

extern "C" {

int hb_run_create(const hb_bayes_args *args, hb_run **out)
{
    if (!args || !out) return hb_fail(HB_ERR_INVALID, "hb_run_create: null argument");
    *out = nullptr;
    hb_run *r = new hb_run();
    const int rc = r->setup(args);
    if (rc) {
        delete r;
        return rc;
    }
    *out = r;
    return HB_OK;
}

int hb_run_step(hb_run *r, int32_t nsteps, int32_t *finished)
{
    if (!r) return hb_fail(HB_ERR_INVALID, "hb_run_step: null run");
    for (int s = 0; s < nsteps && !r->done; s++) {
        const int rc = r->step();
        if (rc) return rc;
    }
    if (finished) *finished = r->done ? 1 : 0;
    return HB_OK;
}

int hb_run_state(hb_run *r, hb_run_info *info)
{
    if (!r || !info) return hb_fail(HB_ERR_INVALID, "hb_run_state: null argument");
    info->iter = r->iter;
    info->records = r->count;
    info->nnz = (double)r->NnzSnp;
    info->vara = r->vara_;
    info->vare = r->vare_;
    info->varg = r->varg;
    info->mu = r->mu;
    for (int j = 0; j < HB_MAX_FOLD; j++) info->pi[j] = 0.0;
    for (int j = 0; j < r->n_fold; j++) info->pi[r->cls_of[j]] = r->Pi[j];
    info->mean_events = r->iter > 0 ? r->events_sum / r->iter : 0.0;
    info->mean_misses = r->iter > 0 ? r->miss_sum / r->iter : 0.0;
    info->mean_redo = r->iter > 0 ? r->redo_sum / r->iter : 0.0;
    info->sweeps_replayed = r->aborts;
    info->resident_bits = r->c ? r->c->layout : 0;
    info->lambda2 = r->lambda2;
    info->loop_seconds = r->loop_seconds;
    info->setup_seconds = r->setup_seconds;
    info->gram_seconds = r->gram_seconds;
    return HB_OK;
}

hb_ctx *hb_run_ctx(hb_run *r) { return r ? r->c : nullptr; }

int hb_run_finish(hb_run *r, hb_bayes_out *out)
{
    if (!r || !out) return hb_fail(HB_ERR_INVALID, "hb_run_finish: null argument");
    return r->finish(out);
}

void hb_run_destroy(hb_run *r) { delete r; }

int hb_bayes_run(const hb_bayes_args *args, hb_bayes_out *out)
{
    if (!args || !out) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: null argument");
    hb_run *r = nullptr;
    int rc = hb_run_create(args, &r);
    if (rc) return rc;
    while (!r->done) {
        rc = r->step();
        if (rc) break;
    }
    if (!rc) rc = r->finish(out);
    delete r;
    return rc;
}

} // extern "C"
