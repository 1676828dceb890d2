// hb_sbayes.hip — host side of hb_sbayes_run(): SBayesD() of the reference (src/SBayesD.cpp:5-609) as validation, prior defaults,
// the outer MCMC loop with its hyper-parameter draws and the posterior assembly; every m-long operation runs on the device
// (hb_sbayes.hpp). SURVEY §8 f4.
#include "hb_internal.hpp"
#include "hb_rng.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

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

struct sb_run {
    hb_sb_dev d;
    std::vector<void *> bufs;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    double *h_acc = nullptr;
    hb_sweep_in *h_in = nullptr;
    ~sb_run()
    {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
        for (void *p : bufs)
            if (p) (void)hipFree(p);
        if (h_acc) (void)hipHostFree(h_acc);
        if (h_in) (void)hipHostFree(h_in);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    template <typename T>
    int alloc(T **p, size_t count)
    {
        HB_HIP(hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T)));
        bufs.push_back(*p);
        HB_HIP(hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), d.stream));
        return HB_OK;
    }
};
} // namespace

extern "C" int hb_sbayes_run(const hb_sbayes_args *args, hb_sbayes_out *o)
{
    if (!args || !o) return hb_fail(HB_ERR_INVALID, "hb_sbayes_run: null argument");
    const auto t_setup = clk::now();
    const hb_sbayes_args &a = *args;
    const int m = a.m;
    if (m < 1 || !a.sumstat || !a.ldm || a.ld_sumstat < m || a.ld_ldm < m) return hb_fail(HB_ERR_INVALID, "Number of SNPs not equals."); // :29-31
    if (!a.model) return hb_fail(HB_ERR_INVALID, "hb_sbayes_run: model is NULL");
    const std::string model = a.model;
    auto line = [&](const char *fmt, ...) {
        if (!a.verbose) return;
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        if (a.log) a.log(buf, a.log_user);
        else { fputs(buf, stdout); fputc('\n', stdout); fflush(stdout); }
    };
    // ---- validation and sizes, :28-74, same order and texts ----
    const int model_index = model == "BayesRR" ? 1 : model == "BayesA" ? 2 : (model == "BayesB" || model == "BayesBpi") ? 3
                          : (model == "BayesC" || model == "BayesCpi") ? 4 : model == "BayesL" ? 5 : 6;
    const double *ss = a.sumstat;
    const int64_t lds = a.ld_sumstat;
    int n;
    {
        double s = 0;
        int c = 0;
        for (int k = 0; k < m; k++)
            if (std::isfinite(ss[3 * lds + k])) { s += ss[3 * lds + k]; c++; }
        n = (int)(s / std::max(1, c)); // :33-34 int n = mean(finite N)
    }
    bool fixpi = (model == "BayesB" || model == "BayesC");
    if (a.n_pi < 2 || !a.Pi) return hb_fail(HB_ERR_INVALID, "Pi should be a vector.");
    if (a.n_pi > HB_MAX_FOLD) return hb_fail(HB_ERR_UNSUPPORTED, "more mixture classes than HB_MAX_FOLD");
    std::vector<double> Pi(a.Pi, a.Pi + a.n_pi);
    const int n_fold = a.n_pi;
    if (arma_sum(Pi.data(), Pi.size()) != 1) return hb_fail(HB_ERR_INVALID, "sum of Pi should be 1.");
    if (Pi[0] == 1) return hb_fail(HB_ERR_INVALID, "all markers have no effect size.");
    for (double p : Pi)
        if (p < 0 || p > 1) return hb_fail(HB_ERR_INVALID, "elements of Pi should be at the range of [0, 1]");
    std::vector<double> fold_(n_fold, 0.0);
    if (a.fold) {
        if (a.n_fold != n_fold) return hb_fail(HB_ERR_INVALID, "length of Pi and fold not equals.");
        fold_.assign(a.fold, a.fold + n_fold);
    } else {
        if (model == "BayesR") return hb_fail(HB_ERR_INVALID, "'fold' should be provided for BayesR model.");
        if (n_fold != 2) return hb_fail(HB_ERR_INVALID, "length of Pi and fold not equals.");
    }
    const int niter = a.niter, nburn = a.nburn, thin = a.thin;
    if (thin < 1) return hb_fail(HB_ERR_INVALID, "hb_sbayes_run: thin must be >= 1");
    const int n_records = std::max(0, (niter - nburn) / thin);
    const bool always_in = (model_index == 1 || model_index == 2 || model_index == 5);
    long long NnzSnp = 0;
    if (always_in) {
        NnzSnp = m;
        Pi[0] = 0;
        Pi[1] = 1;
        fixpi = true;
    } else if (model != "BayesR" && n_fold != 2) {
        return hb_fail(HB_ERR_INVALID, "length of Pi should be 2, the first value is the proportion of non-effect markers.");
    }
    // BayesR with `fold` in any order: the run is the chain of the classes sorted by fold, reported in the caller's order (see hb_run.hip)
    std::vector<int> cls_of(n_fold);
    for (int k = 0; k < n_fold; k++) cls_of[k] = k;
    if (model_index == 6) {
        std::stable_sort(cls_of.begin() + 1, cls_of.end(), [&](int x, int z) { return fold_[x] < fold_[z]; });
        std::vector<double> f2(n_fold), p2(n_fold);
        for (int k = 0; k < n_fold; k++) { f2[k] = fold_[cls_of[k]]; p2[k] = Pi[cls_of[k]]; }
        fold_ = f2;
        Pi = p2;
        for (int k = 2; k < n_fold; k++)
            if (!(fold_[k] > fold_[k - 1]))
                return hb_fail(HB_ERR_UNSUPPORTED, "BayesR on the GPU path needs distinct 'fold' values for the non-null classes");
    }
    // ---- :95-115 ----
    std::vector<double> vx(m), xpx(m), xy(m, 0.0), yyi(m, 0.0), ifest(m, 1.0);
    for (int i = 0; i < m; i++) {
        vx[i] = a.ldm[(size_t)i * a.ld_ldm + i];
        xpx[i] = vx[i] * n;
    }
    int count_y = 0, nvar0 = 0;
    for (int k = 0; k < m; k++) {
        const double b = ss[1 * lds + k], se = ss[2 * lds + k], N = ss[3 * lds + k];
        if (std::isnan(b) || std::isnan(se) || std::isnan(N)) {
            ifest[k] = 0.0;
            nvar0++;
        } else {
            xy[k] = xpx[k] * b;
            yyi[k] = xpx[k] * (b * b + (N - 2) * se * se);
            count_y++;
        }
    }
    if (count_y == 0) return hb_fail(HB_ERR_INVALID, "Lack of SE.");
    const double yy = arma_sum(yyi.data(), m) / count_y;
    const double vary = yy / (n - 1);
    const double h2 = 0.5;
    // ---- priors, :117-170 ----
    const double dfvara_ = a.has_dfvg ? a.dfvg : 4;
    if (dfvara_ <= 2) return hb_fail(HB_ERR_INVALID, "dfvg should not be less than 2.");
    double vara_ = a.has_vg ? a.vg : ((dfvara_ - 2) / dfvara_) * vary * h2;
    double vare_ = a.has_ve ? a.ve : vary * (1 - h2);
    const double dfvare_ = a.has_dfve ? a.dfve : -2;
    const double s2vara_ = a.has_s2vg ? a.s2vg : vara_ * (dfvara_ - 2) / dfvara_;
    const double sumvx = arma_sum(vx.data(), m);
    double varg = vara_ / ((1 - Pi[0]) * sumvx);
    const double s2varg_ = s2vara_ / ((1 - Pi[0]) * sumvx);
    const double s2vare_ = a.has_s2ve ? a.s2ve : 0;
    if (niter < nburn) return hb_fail(HB_ERR_INVALID, "Number of total iteration ('niter') shold be larger than burn-in ('nburn').");
    const double R2 = (dfvara_ - 2) / dfvara_;
    double lambda2 = 2 * (1 - R2) / (R2)*sumvx, lambda = std::sqrt(lambda2);
    const double shape0 = 1.1, rate0 = (shape0 - 1) / lambda2;
    std::vector<double> vara_fold(n_fold), fold_snp_num(n_fold, 0.0), pi_sum(n_fold, 0.0);
    for (int j = 0; j < n_fold; j++) vara_fold[j] = (vara_ / ((1 - Pi[0]) * sumvx)) * fold_[j];
    int nw = 0;
    if (a.windindx)
        for (int i = 0; i < m; i++) {
            if (a.windindx[i] < 1) return hb_fail(HB_ERR_INVALID, "hb_sbayes_run: window ids are 1-based");
            nw = std::max(nw, (int)a.windindx[i]);
        }

    // ---- device ----
    if (hb_device_count() <= 0) return hb_fail(HB_ERR_NO_DEVICE, "no HIP device available: the hibayes GPU engine has no CPU fallback");
    HB_HIP(hipSetDevice(a.device));
    sb_run R;
    hb_sb_dev &d = R.d;
    HB_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    d.m = m;
    d.m_pad = (m + 255) / 256 * 256;
    d.n = n;
    d.seed = a.seed;
    d.nw = nw;
    int rc;
#define TRYA(x) do { rc = (x); if (rc) return rc; } while (0)
    TRYA(R.alloc(&d.ldm, (size_t)m * m));
    TRYA(R.alloc(&d.r_hat, d.m_pad));
    TRYA(R.alloc(&d.xy, d.m_pad));
    TRYA(R.alloc(&d.g, d.m_pad));
    TRYA(R.alloc(&d.xpx, d.m_pad));
    TRYA(R.alloc(&d.vx, d.m_pad));
    TRYA(R.alloc(&d.vargL, d.m_pad));
    TRYA(R.alloc(&d.thr, (size_t)d.m_pad * (HB_MAX_FOLD - 1)));
    TRYA(R.alloc(&d.invv, (size_t)d.m_pad * (HB_MAX_FOLD - 1)));
    TRYA(R.alloc(&d.sdz, (size_t)d.m_pad * (HB_MAX_FOLD - 1)));
    TRYA(R.alloc(&d.acc, HB_ACC_N));
    TRYA(R.alloc(&d.ev_gi, 512)); // SB_GS (hb_sbayes.hpp): the moves of one group
    TRYA(R.alloc(&d.ev_col, 512));
    TRYA(R.alloc(&d.ev_n, 1));
    TRYA(R.alloc(&d.tracker, d.m_pad));
    TRYA(R.alloc(&d.nzrate, d.m_pad));
    TRYA(R.alloc(&d.d_in, 1));
    if (nw) {
        TRYA(R.alloc(&d.wind, d.m_pad));
        TRYA(R.alloc(&d.wflag, nw));
        TRYA(R.alloc(&d.wppa, nw));
        HB_HIP(hipMemcpyAsync(d.wind, a.windindx, sizeof(uint32_t) * m, hipMemcpyHostToDevice, d.stream));
    }
    HB_HIP(hipHostMalloc(reinterpret_cast<void **>(&R.h_acc), sizeof(double) * HB_ACC_N));
    HB_HIP(hipHostMalloc(reinterpret_cast<void **>(&R.h_in), sizeof(hb_sweep_in)));
    HB_HIP(hipMemcpy2DAsync(d.ldm, sizeof(double) * m, a.ldm, sizeof(double) * a.ld_ldm, sizeof(double) * m, m, hipMemcpyHostToDevice, d.stream));
    HB_HIP(hipMemcpyAsync(d.xy, xy.data(), sizeof(double) * m, hipMemcpyHostToDevice, d.stream));
    HB_HIP(hipMemcpyAsync(d.r_hat, xy.data(), sizeof(double) * m, hipMemcpyHostToDevice, d.stream)); // :108 r_hat = xy
    HB_HIP(hipMemcpyAsync(d.xpx, xpx.data(), sizeof(double) * m, hipMemcpyHostToDevice, d.stream));
    HB_HIP(hipMemcpyAsync(d.vx, ifest.data(), sizeof(double) * m, hipMemcpyHostToDevice, d.stream)); // (the kernels' "is this marker sampled" word)
    {
        std::vector<double> vl(m, varg); // :165-168 vargL.fill(varg)
        HB_HIP(hipMemcpyAsync(d.vargL, vl.data(), sizeof(double) * m, hipMemcpyHostToDevice, d.stream));
        HB_HIP(hipStreamSynchronize(d.stream));
    }
    o->n = n;
    o->count_y = count_y;
    o->nw = nw;
    o->n_records = n_records;
    const double setup_seconds = std::chrono::duration<double>(clk::now() - t_setup).count();

    // ---- console, :190-246 ----
    line("Prior parameters:");
    line("    Model fitted at [%s]", model == "BayesRR" ? "Bayes Ridge Regression" : model.c_str());
    line("    Population size %d", n);
    line("    Number of markers %d", m);
    line("    Number of markers used for analysis %d", count_y);
    line("    Total number of iteration %d", niter);
    line("    Total number of burn-in %d", nburn);
    line("    Phenotypic var %f", vary);
    line("    Genetic var %f", vara_);
    line("    Inv-Chisq gpar %f %f", dfvara_, s2vara_);
    line("    Residual var %f", vare_);
    line("    Inv-Chisq epar %f %f", dfvare_, s2vare_);
    line("    Marker var %f", varg);
    line("    Inv-Chisq alpar %f %f", dfvara_, s2varg_);
    if (nw) line("    Number of windows for GWAS analysis %d", nw);
    line("MCMC started: ");
    line(" Iter  NumNZSnp  pi  %sVg  Ve  h2  Timeleft", model == "BayesL" ? "Lambda  " : "");

    std::vector<double> s_alpha, g_sum(m, 0.0), g_host(m);
    if (a.store_alpha) s_alpha.assign((size_t)n_records * m, 0.0);
    double vara_sum = 0, vare_sum = 0, hsq_sum = 0, events_sum = 0;
    int count = 0, nzct = 0, iter = 0;
    const auto t_loop = clk::now();
    for (iter = 0; iter < niter; iter++) {
        if (a.interrupt && a.interrupt(a.interrupt_user)) return hb_fail(HB_ERR_INTERRUPT, "interrupted");
        hb_stream hs(a.seed, hb_sub(HB_PURPOSE_HOST, (uint64_t)iter), 0);
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
        in.store = 0;
        *R.h_in = in;
        HB_HIP(hipMemcpyAsync(d.d_in, R.h_in, sizeof(hb_sweep_in), hipMemcpyHostToDevice, d.stream));
        if (!R.gexec) { // one sweep = 2 ceil(m / 512) + 3 launches: captured once, replayed every iteration
            HB_HIP(hipStreamSynchronize(d.stream));
            HB_HIP(hipStreamBeginCapture(d.stream, hipStreamCaptureModeRelaxed));
            rc = hbk_sb_enqueue_sweep(&d, model_index, n_fold);
            hipError_t e = hipStreamEndCapture(d.stream, &R.graph);
            if (rc) return rc;
            if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            HB_HIP(hipGraphInstantiate(&R.gexec, R.graph, nullptr, nullptr, 0));
        }
        HB_HIP(hipGraphLaunch(R.gexec, d.stream));
        if (in.count_pip && nw) TRYA(hbk_sb_windows(&d));
        HB_HIP(hipMemcpyAsync(R.h_acc, d.acc, sizeof(double) * HB_ACC_N, hipMemcpyDeviceToHost, d.stream));
        HB_HIP(hipStreamSynchronize(d.stream));
        const double *acc = R.h_acc;
        events_sum += acc[HB_ACC_EVENTS];
        auto draw_pi = [&]() { // rdirichlet_sample, src/stats.cpp:69-76
            std::vector<double> xn(n_fold);
            for (int j = 0; j < n_fold; j++) xn[j] = hs.gamma(fold_snp_num[j] + 1, 1.0);
            const double sx = arma_sum(xn.data(), xn.size());
            for (int j = 0; j < n_fold; j++) Pi[j] = xn[j] / sx;
        };
        switch (model_index) {
        case 1: varg = (acc[HB_ACC_SUMG2] + s2varg_ * dfvara_) / hs.chisq(dfvara_ + count_y); break; // :269
        case 2: break;
        case 3: // :321-324
            fold_snp_num[1] = acc[HB_ACC_COUNT0 + 1];
            fold_snp_num[0] = (double)m - nvar0 - fold_snp_num[1];
            NnzSnp = (long long)fold_snp_num[1];
            if (!fixpi) draw_pi();
            break;
        case 4: // :360-365
            fold_snp_num[1] = acc[HB_ACC_COUNT0 + 1];
            fold_snp_num[0] = (double)m - nvar0 - fold_snp_num[1];
            NnzSnp = (long long)fold_snp_num[1];
            varg = (acc[HB_ACC_SUMG2] + s2varg_ * dfvara_) / hs.chisq(dfvara_ + (double)NnzSnp);
            if (!fixpi) draw_pi();
            break;
        case 5: { // :386-389
            const double shape = shape0 + count_y, rate = rate0 + acc[HB_ACC_SUMVARGL] / 2;
            lambda2 = hs.gamma(shape, 1 / rate);
            lambda = std::sqrt(lambda2);
            break;
        }
        case 6: { // :448-460 (class 0 of the device counts already excludes the markers without statistics)
            double nz = 0;
            for (int j = 0; j < n_fold; j++) fold_snp_num[j] = acc[HB_ACC_COUNT0 + j];
            for (int j = 1; j < n_fold; j++) nz += fold_snp_num[j];
            NnzSnp = (long long)nz;
            varg = (acc[HB_ACC_SUMG2] + s2varg_ * dfvara_) / hs.chisq(dfvara_ + (double)NnzSnp);
            for (int j = 0; j < n_fold; j++) vara_fold[j] = varg * fold_[j];
            if (!fixpi) draw_pi();
            break;
        }
        }
        vara_ = (acc[HB_ACC_SUMR] + s2vara_ * dfvara_) / hs.chisq(n + dfvara_);        // :468
        vare_ = (yy - acc[HB_ACC_SUMR2] + s2vare_ * dfvare_) / hs.chisq(n + dfvare_);  // :473
        if (vare_ < 0) vare_ = vara_ * 0.5;                                            // :474
        if (iter >= nburn) nzct++;
        if (iter >= nburn && (iter + 1 - nburn) % thin == 0 && count < n_records) { // :499-512
            if (!fixpi)
                for (int j = 0; j < n_fold; j++) {
                    if (o->s_pi) o->s_pi[(size_t)count * n_fold + cls_of[j]] = Pi[j];
                    pi_sum[j] += Pi[j];
                }
            if (o->s_Vg) o->s_Vg[count] = vara_;
            if (o->s_Ve) o->s_Ve[count] = vare_;
            if (o->s_h2) o->s_h2[count] = vara_ / (vara_ + vare_);
            vara_sum += vara_;
            vare_sum += vare_;
            hsq_sum += vara_ / (vara_ + vare_);
            HB_HIP(hipMemcpy(g_host.data(), d.g, sizeof(double) * m, hipMemcpyDeviceToHost));
            for (int i = 0; i < m; i++) g_sum[i] += g_host[i];
            if (a.store_alpha) std::memcpy(s_alpha.data() + (size_t)count * m, g_host.data(), sizeof(double) * m);
            count++;
        }
        if (a.verbose && a.outfreq > 0 && (iter + 1) % a.outfreq == 0) { // :514-537
            const double el = std::chrono::duration<double>(clk::now() - t_loop).count();
            const int tt = (int)std::floor(el / (iter + 1) * (niter - iter));
            char pis[256] = {0};
            size_t off = 0;
            std::vector<double> pc(n_fold);
            for (int j = 0; j < n_fold; j++) pc[cls_of[j]] = Pi[j];
            for (int j = 0; j < n_fold && off < sizeof(pis) - 16; j++) off += snprintf(pis + off, sizeof(pis) - off, "%.4f ", pc[j]);
            char lam[32] = {0};
            if (model == "BayesL") snprintf(lam, sizeof(lam), "%.4f ", lambda);
            line(" %d %lld %s%s%.4f %.4f %.4f %02dh%02dm%02ds", iter + 1, NnzSnp, pis, lam, vara_, vare_, vara_ / (vara_ + vare_), tt / 3600,
                 tt % 3600 / 60, tt % 3600 % 60);
        }
        if (count == n_records) {
            iter++;
            break;
        }
    }
    const double loop_seconds = std::chrono::duration<double>(clk::now() - t_loop).count();
    // ---- posterior assembly, :541-580 ----
    const double Rn = (double)n_records;
    o->Vg = vara_sum / Rn;
    o->Ve = vare_sum / Rn;
    o->h2 = hsq_sum / Rn;
    if (o->alpha)
        for (int i = 0; i < m; i++) o->alpha[i] = g_sum[i] / Rn;
    if (a.store_alpha && o->s_alpha) std::memcpy(o->s_alpha, s_alpha.data(), sizeof(double) * s_alpha.size());
    if (!fixpi) {
        for (int j = 0; j < n_fold; j++) Pi[j] = pi_sum[j] / Rn;
    } else if (o->s_pi) {
        for (int r = 0; r < n_records; r++) {
            o->s_pi[(size_t)r * n_fold + 0] = Pi[0];
            o->s_pi[(size_t)r * n_fold + 1] = Pi[1];
        }
    }
    if (o->pi)
        for (int j = 0; j < n_fold; j++) o->pi[cls_of[j]] = Pi[j];
    if (o->pip) {
        if (always_in) for (int i = 0; i < m; i++) o->pip[i] = 1.0; // :571
        else {
            std::vector<uint32_t> nz(m);
            HB_HIP(hipMemcpy(nz.data(), d.nzrate, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
            for (int i = 0; i < m; i++) {
                double p = (double)nz[i] / nzct;
                if (p == 1) p = (nzct - 1) / (double)nzct; // :574
                o->pip[i] = p;
            }
        }
    }
    if (nw && o->gwas) {
        std::vector<double> w(nw);
        HB_HIP(hipMemcpy(w.data(), d.wppa, sizeof(double) * nw, hipMemcpyDeviceToHost));
        for (int k = 0; k < nw; k++) {
            double p = w[k] / nzct;
            if (p == 1) p = (nzct - 1) / (double)nzct;
            o->gwas[k] = p;
        }
    }
    if (o->r_hat) HB_HIP(hipMemcpy(o->r_hat, d.r_hat, sizeof(double) * m, hipMemcpyDeviceToHost));
    if (o->g_last) HB_HIP(hipMemcpy(o->g_last, d.g, sizeof(double) * m, hipMemcpyDeviceToHost));
    o->nzct = nzct;
    o->setup_seconds = setup_seconds;
    o->loop_seconds = loop_seconds;
    o->iters_done = iter;
    o->mean_events = iter > 0 ? events_sum / iter : 0;
    line("Posterior parameters:");
    line("    Genetic var %f", o->Vg);
    line("    Residual var %f", o->Ve);
    line("    Estimated h2 %f", o->h2);
    line("Finished: set-up %.2fs, MCMC %.2fs, %.1f sweeps/s", setup_seconds, loop_seconds, loop_seconds > 0 ? iter / loop_seconds : 0.0);
    return HB_OK;
#undef TRYA
}
