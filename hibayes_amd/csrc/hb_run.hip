// hb_run.hip — hb_bayes_run(): the host side of Bayes() (reference src/Bayes.cpp:60-1094).
//
// What stays on the host, as in the reference: argument validation (:92-117, :293, :325, :357), prior
// defaults (:319-374), the outer MCMC loop (:477), the intercept / covariate / random-effect draws
// (:479-516), the hyper-parameter draws after the marker sweep (:603, :666-669, :710-716, :738-741,
// :803-814, :819-823), the thinned store (:848-882) and the posterior assembly (:919-1040).
// What runs on the device: everything that touches an n- or m-long vector (hb_kernels.hip).
// BSLMM (nk) and the single-step epsilon block (ne) are refused with HB_ERR_UNSUPPORTED.
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
int hbk_xalpha(hb_ctx *c, const double *dev_alpha, double *dev_out);

namespace {

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

struct logger {
    const hb_bayes_args *a;
    void line(const char *fmt, ...) const
    {
        if (!a->verbose) return;
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        if (a->log) a->log(buf, a->log_user);
        else { fputs(buf, stdout); fputc('\n', stdout); }
    }
};

struct ctx_guard {
    hb_ctx *c = nullptr;
    double *r0 = nullptr, *u0 = nullptr, *xbuf = nullptr, *dalpha = nullptr, *xa = nullptr;
    ~ctx_guard()
    {
        if (r0) (void)hipFree(r0);
        if (u0) (void)hipFree(u0);
        if (xbuf) (void)hipFree(xbuf);
        if (dalpha) (void)hipFree(dalpha);
        if (xa) (void)hipFree(xa);
        if (c) hb_ctx_destroy(c);
    }
};

} // namespace

extern "C" int hb_bayes_run(const hb_bayes_args *a, hb_bayes_out *o)
{
    if (!a || !o) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: null argument");
    const auto t_setup0 = std::chrono::steady_clock::now();
    const int n = a->n, m = a->m;
    const logger lg{a};
    if (n < 2 || m < 1 || !a->y) return hb_fail(HB_ERR_INVALID, "Number of individuals not equals.");
    if (!a->model) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: model is NULL");
    const std::string model = a->model;

    // ---- validation, same order and texts as src/Bayes.cpp:92-117 ----
    for (int i = 0; i < n; i++)
        if (std::isnan(a->y[i])) return hb_fail(HB_ERR_INVALID, "NAs are not allowed in y.");
    if ((a->X_f64 == nullptr) == (a->X_i8 == nullptr))
        return hb_fail(HB_ERR_INVALID, "hb_bayes_run: exactly one of X_f64 / X_i8 must be given");
    if ((a->X_f64 && a->ld_f64 < n) || (a->X_i8 && a->ld_i8 < n))
        return hb_fail(HB_ERR_INVALID, "Number of individuals not equals.");
    const int model_index = model == "BayesRR" ? 1 : model == "BayesA" ? 2 : (model == "BayesB" || model == "BayesBpi") ? 3
                          : (model == "BayesC" || model == "BayesCpi" || model == "BSLMM") ? 4 : model == "BayesL" ? 5 : 6;
    bool fixpi = (model == "BayesB" || model == "BayesC");
    if (a->n_pi < 2 || !a->Pi) return hb_fail(HB_ERR_INVALID, "Pi should be a vector.");
    if (a->n_pi > HB_MAX_FOLD) return hb_fail(HB_ERR_UNSUPPORTED, "more mixture classes than HB_MAX_FOLD");
    std::vector<double> Pi(a->Pi, a->Pi + a->n_pi);
    const int n_pi = a->n_pi;
    if (arma_sum(Pi.data(), Pi.size()) != 1) return hb_fail(HB_ERR_INVALID, "sum of Pi should be 1.");
    if (Pi[0] == 1) return hb_fail(HB_ERR_INVALID, "all markers have no effect size.");
    for (double p : Pi)
        if (p < 0 || p > 1) return hb_fail(HB_ERR_INVALID, "elements of Pi should be at the range of [0, 1]");
    std::vector<double> fold_;
    if (a->fold) fold_.assign(a->fold, a->fold + a->n_fold);
    else {
        if (model == "BayesR") return hb_fail(HB_ERR_INVALID, "'fold' should be provided for BayesR model.");
        fold_.assign(2, 0.0);
    }
    if ((int)fold_.size() != n_pi) return hb_fail(HB_ERR_INVALID, "length of Pi and fold not equals.");
    const int n_fold = (int)fold_.size();
    if (a->Ki || a->Kival || model == "BSLMM")
        return hb_fail(HB_ERR_UNSUPPORTED, "BSLMM (Ki/Kival) is not part of the GPU path");
    if (a->epsl_index || a->epsl_Gi || a->epsl_y_J)
        return hb_fail(HB_ERR_UNSUPPORTED, "the single-step epsilon block is not part of the GPU path");

    const int world = a->world > 1 ? a->world : 1;
    const int64_t m_global = world > 1 ? a->m_global : m;
    if (world > 1 && (!a->allreduce || m_global < m)) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: sharded run needs allreduce and m_global");

    // ---- sizes, :119-124 ----
    const double vary = var_n1(a->y, n);
    const double h2 = 0.5;
    const int niter = a->niter, nburn = a->nburn, thin = a->thin;
    if (thin < 1) return hb_fail(HB_ERR_INVALID, "hb_bayes_run: thin must be >= 1");
    const int n_records = (niter - nburn) / thin;
    o->n_records = n_records;

    // ---- covariates, :126-147 ----
    const int nc = a->C ? a->nc : 0;
    std::vector<double> beta(nc, 0.0), cpc(nc, 0.0), beta_sum(nc, 0.0);
    if (nc) {
        for (size_t i = 0; i < (size_t)n * nc; i++)
            if (std::isnan(a->C[i]))
                return hb_fail(HB_ERR_INVALID, "Individuals with phenotypic value should not have missing covariates.");
        for (int i = 0; i < nc; i++) {
            const double *ci = a->C + (size_t)i * n;
            double s = 0;
            for (int k = 0; k < n; k++) s += ci[k] * ci[k];
            cpc[i] = s;
        }
    }

    // ---- environmental random effects, :149-201 with makeZ :29-57 ----
    const int nr = a->R ? a->nr : 0;
    const double dfr = a->has_dfvr ? a->dfvr : -1;
    const double s2r = a->has_s2vr ? a->s2vr : 0;
    std::vector<double> vr(nr, 0.0), vrtmp(nr, vary * (1 - h2) / (nr + 1)), vr_sum(nr, 0.0);
    std::vector<int32_t> zid((size_t)n * nr), nlev(nr), lev_first(nr);
    std::vector<double> zz;
    int n_levels = 0;
    for (int t = 0; t < nr; t++) {
        std::vector<std::string> vals(n);
        for (int k = 0; k < n; k++) {
            const char *s = a->R[(size_t)t * n + k];
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
        if (o->r_term_nlevels) o->r_term_nlevels[t] = nlev[t];
    }
    o->n_levels = n_levels;
    std::vector<double> estR(n_levels, 0.0), estR_sum(n_levels, 0.0), estR_new, r_RHS, lev_delta;

    // ---- :288-296 ----
    const bool always_in = (model_index == 1 || model_index == 2 || model_index == 5);
    if (always_in) {
        Pi[0] = 0; Pi[1] = 1;
        fixpi = true;
    } else if (model != "BayesR" && n_pi != 2) {
        return hb_fail(HB_ERR_INVALID, "length of Pi should be 2, the first value is the proportion of non-effect markers.");
    }
    // :319-326
    const double dfvara_ = a->has_dfvg ? a->dfvg : 4;
    if (dfvara_ <= 2) return hb_fail(HB_ERR_INVALID, "dfvg should not be less than 2.");
    if (niter < nburn) return hb_fail(HB_ERR_INVALID, "Number of total iteration ('niter') shold be larger than burn-in ('nburn').");
    if (model_index == 6)
        for (int k = 2; k < n_fold; k++)
            if (!(fold_[k] > fold_[k - 1]))
                return hb_fail(HB_ERR_UNSUPPORTED, "BayesR on the GPU path needs 'fold' in strictly increasing order");

    // =========================== device set-up ===========================
    ctx_guard G;
    hb_ctx_params cp{};
    cp.device = a->device;
    cp.n = n;
    cp.m = m;
    cp.panel = a->panel;
    cp.precise = a->precise;
    cp.m_offset = world > 1 ? a->m_offset : 0;
    cp.seed = a->seed;
    int rc = hb_ctx_create(&cp, &G.c);
    if (rc) return rc;
    hb_ctx *c = G.c;
    if (a->X_i8) rc = hb_ctx_upload_genotype_i8(c, a->X_i8, a->ld_i8, 0, m);
    else rc = hb_ctx_upload_genotype_f64(c, a->X_f64, a->ld_f64, 0, m);
    if (rc) return rc;

    // exchange buffer for the sharded run
    const size_t xcount = hb_exchange_count(n);
    double *xbuf = nullptr;
    if (world > 1) {
        if (a->exchange_buf) xbuf = static_cast<double *>(a->exchange_buf);
        else {
            HB_HIP(hipMalloc(reinterpret_cast<void **>(&G.xbuf), sizeof(double) * xcount));
            xbuf = G.xbuf;
        }
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&G.r0), sizeof(double) * n));
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&G.u0), sizeof(double) * n));
    }
    auto allreduce_host = [&](double *vals, int cnt) -> int { // small host vectors through the device buffer
        if (world == 1) return HB_OK;
        HB_HIP(hipMemsetAsync(xbuf, 0, sizeof(double) * xcount, c->stream));
        HB_HIP(hipMemcpyAsync(xbuf, vals, sizeof(double) * cnt, hipMemcpyHostToDevice, c->stream));
        HB_HIP(hipStreamSynchronize(c->stream));
        if (a->allreduce(xbuf, xcount, a->allreduce_user)) return hb_fail(HB_ERR_COMM, "all-reduce callback failed");
        HB_HIP(hipMemcpy(vals, xbuf, sizeof(double) * cnt, hipMemcpyDeviceToHost));
        return HB_OK;
    };

    // ---- marker statistics, :310-317 ----
    double sumvx = 0;
    int nvar0 = 0;
    rc = hb_ctx_marker_stats(c, nullptr, nullptr, &sumvx, &nvar0);
    if (rc) return rc;
    {
        double sv[2] = {sumvx, (double)nvar0};
        rc = allreduce_host(sv, 2);
        if (rc) return rc;
        sumvx = sv[0];
        nvar0 = (int)sv[1];
    }
    double gram_s = 0;
    rc = hb_ctx_build_gram(c, &gram_s);
    if (rc) return rc;

    // ---- prior defaults, :327-374 ----
    double vara_ = a->has_vg ? a->vg : ((dfvara_ - 2) / dfvara_) * vary * h2;
    double vare_ = a->has_ve ? a->ve : vary * (1 - h2) / (nr + 1);
    const double dfvare_ = a->has_dfve ? a->dfve : -2;
    const double s2vara_ = a->has_s2vg ? a->s2vg : vara_ * (dfvara_ - 2) / dfvara_;
    double varg = vara_ / ((1 - Pi[0]) * sumvx);
    const double s2varg_ = s2vara_ / ((1 - Pi[0]) * sumvx);
    const double s2vare_ = a->has_s2ve ? a->s2ve : 0;
    const double R2 = (dfvara_ - 2) / dfvara_;
    double lambda2 = 2 * (1 - R2) / (R2)*sumvx;
    double lambda = std::sqrt(lambda2);
    const double shape0 = 1.1;
    const double rate0 = (shape0 - 1) / lambda2;
    std::vector<double> vara_fold(n_fold), fold_snp_num(n_fold, 0.0);
    for (int j = 0; j < n_fold; j++) vara_fold[j] = (vara_ / ((1 - Pi[0]) * sumvx)) * fold_[j];
    if (model_index == 5) {
        std::vector<double> vl(m, varg);
        rc = hb_ctx_set_effects(c, nullptr, nullptr, vl.data());
        if (rc) return rc;
    }
    int nw = 0;
    if (a->windindx) {
        for (int i = 0; i < m; i++) nw = std::max(nw, (int)a->windindx[i]);
        double w = nw;
        if (world > 1) { // windows are global ids: every rank needs the same nw
            std::vector<double> tmp(1, w);
            // max via sum is wrong; exchange rank-wise maxima instead
            std::vector<double> slots(world, 0.0);
            slots[a->rank] = w;
            rc = allreduce_host(slots.data(), world);
            if (rc) return rc;
            for (double s : slots) nw = std::max(nw, (int)s);
        }
        rc = hb_ctx_set_windows(c, a->windindx, nw);
        if (rc) return rc;
    }
    o->nw = nw;
    if (nc) { rc = hb_ctx_set_covariates(c, a->C, nc); if (rc) return rc; }
    if (nr) { rc = hb_ctx_set_levels(c, zid.data(), nr, nlev.data()); if (rc) return rc; }

    // ---- console, :393-461 ----
    lg.line("Prior parameters:");
    lg.line("    Model fitted at [%s]", model == "BayesRR" ? "Bayes Ridge Regression" : model.c_str());
    lg.line("    Number of observations %d", n);
    lg.line("    Number of covariates %d", nc + 1);
    lg.line("    Number of envir-random effects %d", nr);
    lg.line("    Number of markers %lld", (long long)m_global);
    lg.line("    Total number of iteration %d", niter);
    lg.line("    Total number of burn-in %d", nburn);
    lg.line("    Frequency of collecting %d", thin);
    lg.line("    Phenotypic var %f", vary);
    lg.line("    Genetic var %f", vara_);
    lg.line("    Inv-Chisq gpar %f %f", dfvara_, s2vara_);
    lg.line("    Residual var %f", vare_);
    lg.line("    Inv-Chisq epar %f %f", dfvare_, s2vare_);
    lg.line("    Marker var %f", varg);
    lg.line("    Inv-Chisq alpar %f %f", dfvara_, s2varg_);
    if (nw) lg.line("    Number of windows for GWAS analysis %d", nw);
    lg.line("MCMC started: ");
    lg.line(" Iter  NumNZSnp  pi  %sVg  Ve  h2  Timeleft", model == "BayesL" ? "Lambda  " : "");

    // ---- :469-472 ----
    double mu = arma_sum(a->y, n) / n, mu_;
    {
        std::vector<double> yadj(n), zero(n, 0.0);
        for (int i = 0; i < n; i++) yadj[i] = a->y[i] - mu;
        rc = hb_ctx_set_residual(c, yadj.data(), zero.data());
        if (rc) return rc;
    }
    double sum_r = 0, sum_r2 = 0;
    rc = hb_ctx_residual_sums(c, &sum_r, &sum_r2);
    if (rc) return rc;
    o->setup_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup0).count();

    int count = 0, nzct = 0;
    long long NnzSnp = always_in ? m_global : 0;
    double mu_sum = 0, vara_sum = 0, vare_sum = 0, hsq_sum = 0, events_sum = 0;
    std::vector<double> pi_sum(n_fold, 0.0), gtmp;
    if (a->store_alpha && o->s_alpha) gtmp.resize(m);
    const auto t_loop0 = std::chrono::steady_clock::now();
    int iter;

    // =============================== MCMC, :477-917 ===============================
    for (iter = 0; iter < niter; iter++) {
        if (a->interrupt && a->interrupt(a->interrupt_user)) return hb_fail(HB_ERR_INTERRUPT, "interrupted");
        hb_stream hs(a->seed, hb_sub(HB_PURPOSE_HOST, (uint64_t)iter), 0);

        // sample intercept, :479-482
        mu_ = -(sum_r / n + std::sqrt(vare_ / n) * hs.norm());
        mu -= mu_;
        rc = hb_ctx_residual_shift(c, mu_);
        if (rc) return rc;

        // covariates, :484-494
        for (int i = 0; i < nc; i++) {
            const double oldgi = beta[i], v = cpc[i];
            double rhs;
            rc = hb_ctx_cov_dot(c, i, &rhs);
            if (rc) return rc;
            rhs += v * oldgi;
            const double gi = rhs / v + std::sqrt(vare_ / v) * hs.norm();
            rc = hb_ctx_cov_axpy(c, i, oldgi - gi);
            if (rc) return rc;
            beta[i] = gi;
        }

        // environmental random effects, :496-516
        for (int t = 0; t < nr; t++) {
            const int q0 = lev_first[t], qr = nlev[t];
            r_RHS.assign(qr, 0.0);
            estR_new.assign(qr, 0.0);
            lev_delta.assign(qr, 0.0);
            rc = hb_ctx_level_sums(c, t, r_RHS.data());
            if (rc) return rc;
            for (int q = 0; q < qr; q++) r_RHS[q] += zz[q0 + q] * estR[q0 + q];
            for (int q = 0; q < qr; q++) {
                const double l = zz[q0 + q] + vare_ / vrtmp[t];
                estR_new[q] = r_RHS[q] / l + std::sqrt(vare_ / l) * hs.norm();
                lev_delta[q] = estR[q0 + q] - estR_new[q];
            }
            rc = hb_ctx_level_axpy(c, t, lev_delta.data());
            if (rc) return rc;
            double ss = 0;
            for (int q = 0; q < qr; q++) ss += estR_new[q] * estR_new[q];
            vrtmp[t] = (ss + s2r * dfr) / hs.chisq(qr + dfr);
            vr[t] = var_n1(estR_new.data(), qr);
            for (int q = 0; q < qr; q++) estR[q0 + q] = estR_new[q];
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
        if (world > 1) {
            HB_HIP(hipMemcpyAsync(G.r0, c->r, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
            HB_HIP(hipMemcpyAsync(G.u0, c->u, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
        }
        hb_sweep_out so{};
        rc = hb_ctx_sweep(c, &in, &so);
        if (rc) return rc;
        if (world > 1) {
            // once per sweep: sum the shards' residual deltas and scalar sums (SURVEY §8 e)
            rc = hbk_delta_pack(c, G.r0, G.u0, xbuf);
            if (rc) return rc;
            HB_HIP(hipMemcpyAsync(xbuf + 2 * (size_t)n, c->acc, sizeof(double) * HB_ACC_N, hipMemcpyDeviceToDevice, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
            if (a->allreduce(xbuf, xcount, a->allreduce_user)) return hb_fail(HB_ERR_COMM, "all-reduce callback failed");
            rc = hbk_delta_unpack(c, G.r0, G.u0, xbuf);
            if (rc) return rc;
            HB_HIP(hipMemcpyAsync(c->acc, xbuf + 2 * (size_t)n, sizeof(double) * HB_ACC_N, hipMemcpyDeviceToDevice, c->stream));
            rc = hbk_reduce_ru(c);
            if (rc) return rc;
            HB_HIP(hipMemcpyAsync(c->h_acc, c->acc, sizeof(double) * HB_ACC_N, hipMemcpyDeviceToHost, c->stream));
            HB_HIP(hipStreamSynchronize(c->stream));
            so.sum_g2 = c->h_acc[HB_ACC_SUMG2];
            for (int k = 0; k < HB_MAX_FOLD; k++) so.class_count[k] = c->h_acc[HB_ACC_COUNT0 + k];
            so.sum_vargL = c->h_acc[HB_ACC_SUMVARGL];
            so.n_events = c->h_acc[HB_ACC_EVENTS];
            so.sum_r = c->h_acc[HB_ACC_SUMR];
            so.sum_r2 = c->h_acc[HB_ACC_SUMR2];
            so.var_u = c->h_acc[HB_ACC_VARU];
        }
        events_sum += so.n_events;
        sum_r = so.sum_r;
        sum_r2 = so.sum_r2;

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
        vara_ = so.var_u;                                                            // :819
        vare_ = (sum_r2 + s2vare_ * dfvare_) / hs.chisq((double)n + dfvare_);       // :823

        if (iter >= nburn) nzct++; // :826-845 (the counters themselves live on the device)

        // thinned store, :848-882
        if (in.store) {
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
            if (a->store_alpha && o->s_alpha) {
                rc = hb_ctx_get_effects(c, o->s_alpha + (size_t)count * m, nullptr, nullptr);
                if (rc) return rc;
            }
            for (int i = 0; i < nc; i++) {
                if (o->s_beta) o->s_beta[(size_t)count * nc + i] = beta[i];
                beta_sum[i] += beta[i];
            }
            double vt = vara_ + vare_;
            for (int t = 0; t < nr; t++) {
                vt += vr[t];
                if (o->s_Vr) o->s_Vr[(size_t)count * nr + t] = vr[t];
                vr_sum[t] += vr[t];
            }
            for (int q = 0; q < n_levels; q++) {
                if (o->s_r) o->s_r[(size_t)count * n_levels + q] = estR[q];
                estR_sum[q] += estR[q];
            }
            if (o->s_h2) o->s_h2[count] = vara_ / vt;
            hsq_sum += vara_ / vt;
            count++;
        }

        if (a->verbose && a->outfreq > 0 && (iter + 1) % a->outfreq == 0) { // :884-914
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count();
            const int tt = (int)std::floor(el / (iter + 1) * (niter - iter));
            double vt = vara_ + vare_;
            for (int t = 0; t < nr; t++) vt += vr[t];
            char pis[256] = {0};
            size_t off = 0;
            for (int j = 0; j < n_fold && off < sizeof(pis) - 16; j++) off += snprintf(pis + off, sizeof(pis) - off, "%.4f ", Pi[j]);
            char lam[32] = {0};
            if (model == "BayesL") snprintf(lam, sizeof(lam), "%.4f ", lambda);
            lg.line(" %d %lld %s%s%.4f %.4f %.4f %02dh%02dm%02ds", iter + 1, NnzSnp, pis, lam, vara_, vare_, vara_ / vt,
                    tt / 3600, tt % 3600 / 60, tt % 3600 % 60);
        }
        if (count == n_records) { iter++; break; } // :916
    }
    o->loop_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop0).count();
    o->iters_done = iter;
    o->mean_events = iter > 0 ? events_sum / iter : 0;

    // ============================ posterior assembly, :919-1040 ============================
    const double Rn = (double)n_records;
    o->Vg = vara_sum / Rn;
    o->Ve = vare_sum / Rn;
    o->h2 = hsq_sum / Rn;
    const double Mu = mu_sum / Rn;
    o->mu = Mu;
    std::vector<double> e(n), nz(m), asum(m), asq(m);
    for (int i = 0; i < n; i++) e[i] = a->y[i] - Mu;
    for (int i = 0; i < nc; i++) {
        const double b = beta_sum[i] / Rn;
        if (o->beta) o->beta[i] = b;
        const double *ci = a->C + (size_t)i * n;
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
    { // e -= X * alpha (:971), one device mat-vec; shards sum their partial products
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&G.dalpha), sizeof(double) * (size_t)c->m_pad));
        HB_HIP(hipMemset(G.dalpha, 0, sizeof(double) * (size_t)c->m_pad));
        HB_HIP(hipMemcpy(G.dalpha, asum.data(), sizeof(double) * m, hipMemcpyHostToDevice));
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&G.xa), sizeof(double) * (size_t)c->ld));
        rc = hbk_xalpha(c, G.dalpha, G.xa);
        if (rc) return rc;
        std::vector<double> xa(n);
        HB_HIP(hipStreamSynchronize(c->stream));
        HB_HIP(hipMemcpy(xa.data(), G.xa, sizeof(double) * n, hipMemcpyDeviceToHost));
        if (world > 1) {
            HB_HIP(hipMemset(xbuf, 0, sizeof(double) * xcount));
            HB_HIP(hipMemcpy(xbuf, xa.data(), sizeof(double) * n, hipMemcpyHostToDevice));
            if (a->allreduce(xbuf, xcount, a->allreduce_user)) return hb_fail(HB_ERR_COMM, "all-reduce callback failed");
            HB_HIP(hipMemcpy(xa.data(), xbuf, sizeof(double) * n, hipMemcpyDeviceToHost));
        }
        for (int k = 0; k < n; k++) e[k] -= xa[k];
    }
    if (!fixpi) {
        for (int j = 0; j < n_fold; j++) Pi[j] = pi_sum[j] / Rn;
    } else if (o->s_pi) { // :979-983
        for (int r = 0; r < n_records; r++) {
            o->s_pi[(size_t)r * n_fold + 0] = Pi[0];
            o->s_pi[(size_t)r * n_fold + 1] = Pi[1];
            for (int j = 2; j < n_fold; j++) o->s_pi[(size_t)r * n_fold + j] = 0.0;
        }
    }
    if (o->pi) for (int j = 0; j < n_pi; j++) o->pi[j] = Pi[j];
    if (nr) {
        for (int t = 0; t < nr; t++) if (o->Vr) o->Vr[t] = vr_sum[t] / Rn;
        for (int q = 0; q < n_levels; q++) estR_sum[q] /= Rn;
        for (int t = 0; t < nr; t++)
            for (int k = 0; k < n; k++) e[k] -= estR_sum[lev_first[t] + zid[(size_t)t * n + k]];
        if (o->r_est) std::memcpy(o->r_est, estR_sum.data(), sizeof(double) * n_levels);
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
    if (nw && o->gwas) { // :1034-1038
        std::vector<double> w(nw);
        rc = hb_ctx_get_windows(c, w.data());
        if (rc) return rc;
        if (world > 1) { // a window is hit if any shard saw it; shards hold disjoint markers but may share windows
            // counts were accumulated per shard per iteration; exact any() across shards needs per-iteration
            // exchange, so windows must not straddle shards (checked by the Python driver)
        }
        for (int k = 0; k < nw; k++) {
            double p = w[k] / nzct;
            if (p == 1) p = (nzct - 1) / (double)nzct;
            o->gwas[k] = p;
        }
    }
    o->nzct = nzct;
    lg.line("Posterior parameters:");
    lg.line("    Mu %f", Mu);
    lg.line("    Genetic var %f", o->Vg);
    lg.line("    Residual var %f", o->Ve);
    lg.line("    Estimated h2 %f", o->h2);
    lg.line("Finished: set-up %.2fs (Gram %.2fs), MCMC %.2fs, %.1f sweeps/s", o->setup_seconds, gram_s, o->loop_seconds,
            o->loop_seconds > 0 ? iter / o->loop_seconds : 0.0);
    return HB_OK;
}
