// shim/Bayes_gpu.cpp — the reference-side binding: a drop-in for the body of hibayes' exported Bayes()
// (reference src/Bayes.cpp:59-88; its generated glue _hibayes_Bayes, src/RcppExports.cpp:16-50, stays as it is) that forwards the
// 27 arguments to hb_bayes_run() of libhibayes_gpu.so and rebuilds the same Rcpp::List (src/Bayes.cpp:919-1040).
//
// UNCOMPILED HERE: R, Rcpp and RcppArmadillo do not exist in the build image. The same argument mapping is exercised by
// the ctypes binding (hibayes_amd/bayes.py) and by the plain-C caller examples/ibrm_demo.c, which the tests compile and run.
// In hibayes: put this file into src/ in place of Bayes.cpp's Bayes(), and in src/Makevars
//     PKG_CPPFLAGS += -I<repo>/include
//     PKG_LIBS     += -L<repo>/hibayes_amd -lhibayes_gpu -Wl,-rpath,<repo>/hibayes_amd
#include <RcppArmadillo.h>
#include <algorithm>
#include <string>
#include <vector>
#include "hibayes_gpu.h"

using namespace Rcpp;

// the level names of each random-effect column, in the order makeZ() numbers them (src/Bayes.cpp:36-37: sorted unique strings)
static CharacterVector levels_of(const CharacterMatrix &R, const std::vector<int32_t> &nlev)
{
    CharacterVector out;
    for (int j = 0; j < R.ncol(); j++) {
        std::vector<std::string> v;
        for (int i = 0; i < R.nrow(); i++) v.push_back(as<std::string>(R(i, j)));
        std::stable_sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        if ((int)v.size() != nlev[j]) stop("levels_of: level count differs from the library's");
        for (auto &s : v) out.push_back(s);
    }
    return out;
}

// [[Rcpp::export]]
Rcpp::List Bayes(arma::vec &y, arma::mat &X, std::string model, arma::vec Pi,
                 const Nullable<arma::vec> Kival = R_NilValue, const Nullable<arma::mat> Ki = R_NilValue,
                 const Nullable<arma::mat> C = R_NilValue, const Nullable<CharacterMatrix> R = R_NilValue,
                 const Nullable<arma::vec> fold = R_NilValue, const int niter = 50000, const int nburn = 20000,
                 const int thin = 5, const Nullable<arma::vec> epsl_y_J = R_NilValue,
                 const Nullable<arma::sp_mat> epsl_Gi = R_NilValue, const Nullable<arma::uvec> epsl_index = R_NilValue,
                 const Nullable<double> dfvr = R_NilValue, const Nullable<double> s2vr = R_NilValue,
                 const Nullable<double> vg = R_NilValue, const Nullable<double> dfvg = R_NilValue,
                 const Nullable<double> s2vg = R_NilValue, const Nullable<double> ve = R_NilValue,
                 const Nullable<double> dfve = R_NilValue, const Nullable<double> s2ve = R_NilValue,
                 const Nullable<arma::uvec> windindx = R_NilValue, const int outfreq = 100,
                 const int threads = 0, const bool verbose = true)
{
    hb_bayes_args a = {};
    a.n = y.n_elem;  a.m = X.n_cols;  a.y = y.memptr();
    a.X_f64 = X.memptr();  a.ld_f64 = X.n_rows;          // reference layout; integrality is checked on upload.
    // (fast path: hand the bigmemory .bin mapping over as a.X_i8 / a.ld_i8 and skip as.matrix(), R/bayes.r:284)
    a.model = model.c_str();
    a.Pi = Pi.memptr();  a.n_pi = Pi.n_elem;
    arma::vec fold_;  if (fold.isNotNull())  { fold_ = as<arma::vec>(fold);  a.fold = fold_.memptr(); a.n_fold = fold_.n_elem; }
    arma::mat C_;     if (C.isNotNull())     { C_ = as<arma::mat>(C);        a.C = C_.memptr();       a.nc = C_.n_cols; }
    std::vector<std::string> rs;  std::vector<const char*> rp;
    CharacterMatrix R_;
    if (R.isNotNull()) { R_ = as<CharacterMatrix>(R);                        // column-major n x nr
        for (int j = 0; j < R_.ncol(); j++) for (int i = 0; i < R_.nrow(); i++) rs.push_back(as<std::string>(R_(i, j)));
        for (auto &s : rs) rp.push_back(s.c_str());
        a.R = rp.data();  a.nr = R_.ncol(); }
    // BSLMM and the single-step epsilon block are outside the GPU path: ANY non-NULL value reaches the library, which answers
    // HB_ERR_UNSUPPORTED with a text (-> an R error), never a silent fit without the block
    arma::vec Kival_; if (Kival.isNotNull()) { Kival_ = as<arma::vec>(Kival); a.Kival = Kival_.memptr(); }
    arma::mat Ki_;    if (Ki.isNotNull())    { Ki_ = as<arma::mat>(Ki);       a.Ki = Ki_.memptr(); }
    arma::vec ey_;    if (epsl_y_J.isNotNull())   { ey_ = as<arma::vec>(epsl_y_J); a.epsl_y_J = ey_.memptr(); }
    arma::sp_mat eg_; if (epsl_Gi.isNotNull())    { eg_ = as<arma::sp_mat>(epsl_Gi); a.epsl_Gi = (const void *)&eg_; }
    arma::uvec ei_;   if (epsl_index.isNotNull()) { ei_ = as<arma::uvec>(epsl_index); a.epsl_index = (const uint32_t *)ei_.memptr(); }
    a.niter = niter;  a.nburn = nburn;  a.thin = thin;
    #define OPT(name) if (name.isNotNull()) { a.has_##name = 1; a.name = as<double>(name); }
    OPT(dfvr) OPT(s2vr) OPT(vg) OPT(dfvg) OPT(s2vg) OPT(ve) OPT(dfve) OPT(s2ve)
    #undef OPT
    std::vector<uint32_t> w;
    if (windindx.isNotNull()) { arma::uvec w_ = as<arma::uvec>(windindx); w.assign(w_.begin(), w_.end()); a.windindx = w.data(); }
    a.outfreq = outfreq;  a.threads = threads;  a.verbose = verbose;
    a.seed = (uint64_t)(unif_rand() * 4294967296.0);     // one draw from R's stream: set.seed() in ibrm() (R/bayes.r:151) still governs the run
    a.store_alpha = 1;
    a.precise = 2;                                       // exact fixed-point panel mat-vec (fp64-grade)
    a.genotype_bits = 0;                                 // auto: 2 bits per genotype resident where exact and faster (codes 0..3, BayesB / C), int8 otherwise; same chain
    a.interrupt = [](void*) -> int { try { Rcpp::checkUserInterrupt(); return 0; } catch (...) { return 1; } };
    a.log = [](const char *line, void*) { Rcpp::Rcout << line << std::endl; };

    const int n_records = (niter - nburn) / thin, nc = a.nc, nr = a.nr, K = a.n_pi;
    const int nw = a.windindx ? (int)*std::max_element(w.begin(), w.end()) : 0;
    arma::vec alpha(a.m), pi(K), g(a.n), e(a.n), pip(a.m), beta(nc), Vr(nr), r_est((size_t)a.n * std::max(nr, 1)), gwas(nw);
    arma::mat s_alpha(a.m, n_records), s_pi(K, n_records), s_beta(nc, n_records), s_Vr(nr, n_records),
              s_r((size_t)a.n * std::max(nr, 1), n_records);
    arma::rowvec s_Vg(n_records), s_Ve(n_records), s_h2(n_records), s_mu(n_records);
    std::vector<int32_t> nlev(nr);
    hb_bayes_out o = {};
    o.alpha = alpha.memptr(); o.pi = pi.memptr(); o.g = g.memptr(); o.e = e.memptr(); o.pip = pip.memptr();
    o.beta = beta.memptr(); o.Vr = Vr.memptr(); o.r_est = r_est.memptr(); o.r_term_nlevels = nlev.data(); o.gwas = gwas.memptr();
    o.s_alpha = s_alpha.memptr(); o.s_pi = s_pi.memptr(); o.s_beta = s_beta.memptr(); o.s_Vr = s_Vr.memptr(); o.s_r = s_r.memptr();
    o.s_Vg = s_Vg.memptr(); o.s_Ve = s_Ve.memptr(); o.s_h2 = s_h2.memptr(); o.s_mu = s_mu.memptr();

    if (hb_bayes_run(&a, &o) != HB_OK) throw Rcpp::exception(hb_last_error());   // same texts as src/Bayes.cpp:92-117

    List results, MCMCsample;                                    // same names/shapes as src/Bayes.cpp:919-1040
    if (nr) { results["Vr"] = Vr; MCMCsample["Vr"] = s_Vr; }
    results["Vg"] = o.Vg; results["Ve"] = o.Ve; results["h2"] = o.h2; results["mu"] = o.mu;
    MCMCsample["Vg"] = s_Vg; MCMCsample["Ve"] = s_Ve; MCMCsample["h2"] = s_h2; MCMCsample["mu"] = s_mu;
    if (nc) { results["beta"] = beta; MCMCsample["beta"] = s_beta; }
    results["alpha"] = alpha; MCMCsample["alpha"] = s_alpha;
    results["pi"] = pi;       MCMCsample["pi"] = s_pi;
    if (nr) { // Levels = sorted unique strings of each R column (makeZ, :36-37), Estimation = r_est[0 : o.n_levels)
        results["r"] = DataFrame::create(Named("Levels") = levels_of(R_, nlev), Named("Estimation") = arma::vec(r_est.head(o.n_levels)));
        MCMCsample["r"] = arma::mat(s_r.head_rows(o.n_levels)); }
    results["g"] = g; results["e"] = e; results["pip"] = pip;
    if (nw) results["gwas"] = gwas;
    results["MCMCsamples"] = MCMCsample;
    return results;
}
