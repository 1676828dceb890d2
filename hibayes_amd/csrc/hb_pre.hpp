// hb_pre.hpp — k_pre: everything per marker that does not depend on the running right-hand side (deviates, 1/v, sd*z, the inclusion test as thresholds on rhs^2).
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

// ---------------------------------------------------------------------------------------------
// k_pre: everything per marker that does not depend on the running rhs.
// Conditional posteriors restated as thresholds on q = rhs^2:
//   B/C (src/Bayes.cpp:640-645 / :683-688): included  <=>  U >= 1/(1+exp(s1-s0))
//        <=>  s1-s0 >= log((1-U)/U)  <=>  q >= 2 v vare (log((1-U)/U) + ldV/2 - logpi1 + logpi0)
//   R   (:759-781): class > c  <=>  U >= P(class <= c | q); with fold ascending that cumulative
//        probability decreases in q, so the K-1 boundaries are thresholds thr_0 <= thr_1 <= ...
//        found here by safeguarded Newton on  log B(q) - log A(q) = log((1-U)/U).
// ---------------------------------------------------------------------------------------------
struct pre_view {
    int m, m_pad;
    int64_t m_offset;
    uint64_t seed;
    const double *xpx, *vx, *g, *vargL;
    double *thr, *invv, *sdz;
    int kpad; // thresholds written per marker (1, 3 or 7)
};

__device__ double bayesr_threshold(int K, int c, const double *a, const double *b, double logT)
{
    // h(q) = logsumexp_{i>c}(a_i + b_i q) - logsumexp_{i<=c}(a_i + b_i q) - logT, increasing in q
    // (exp(0) = 1 and log(1) = 0 exactly: the term that IS its group's maximum needs no exp, a one-term group no log — the
    // same numbers with about half of the transcendental calls; this function is most of k_pre's millisecond for BayesR)
    auto h = [&](double q, double &dh) {
        double mA = -HB_INF, mB = -HB_INF;
        for (int i = 0; i < K; i++) {
            const double s = a[i] + b[i] * q;
            if (i <= c) mA = fmax(mA, s); else mB = fmax(mB, s);
        }
        double sA = 0, sB = 0, dA = 0, dB = 0;
        for (int i = 0; i < K; i++) {
            const double s = a[i] + b[i] * q;
            if (i <= c) { const double w = s == mA ? 1.0 : exp(s - mA); sA += w; dA += b[i] * w; }
            else        { const double w = s == mB ? 1.0 : exp(s - mB); sB += w; dB += b[i] * w; }
        }
        dh = dB / sB - dA / sA;
        return (mB + (sB == 1.0 ? 0.0 : log(sB))) - (mA + (sA == 1.0 ? 0.0 : log(sA))) - logT;
    };
    double dh;
    double h0 = h(0.0, dh);
    if (!(h0 < 0.0)) return 0.0;         // already above the boundary at q = 0
    if (!(dh > 0.0)) return HB_INF;      // flat: the boundary is never crossed
    // bracket
    double lo = 0.0, hi = -h0 / dh;
    if (!(hi > 0.0)) hi = 1.0;
    double hh = h(hi, dh);
    int guard = 0;
    while (hh < 0.0 && guard++ < 200) {
        lo = hi;
        hi *= 2.0;
        hh = h(hi, dh);
    }
    if (hh < 0.0) return HB_INF;
    double q = hi;
    for (int it = 0; it < 100; it++) {
        double d;
        const double hv = h(q, d);
        if (hv < 0.0) lo = q; else hi = q;
        // Round 6: a Newton step below the last bits of q means q IS the root — stop. The test used to come only after the safeguard, and the
        // safeguard rejects a step of zero (qn == q == hi is not inside (lo, hi)): a search that had converged in four steps then bisected
        // its whole bracket again, fifty evaluations of four exp and two log each — one boundary in ten, i.e. nearly every wave: k_pre was
        // 1.09 ms of a BayesR sweep at m = 500 000 (a tenth of a converged sweep, nothing beside it) and is 0.2 now; at most eleven evaluations
        // instead of sixty-two, the same roots to 2e-13 (measured offline over 18 000 boundaries, tools/r6_newton_sim.py).
        const double step = hv / d;
        if (fabs(step) <= 4e-16 * fabs(q)) break;
        double qn = q - step;
        if (!(qn > lo && qn < hi)) qn = 0.5 * (lo + hi);
        if (fabs(qn - q) <= 4e-16 * fabs(qn) || hi - lo <= 4e-16 * hi) { q = qn; break; }
        q = qn;
    }
    return q;
}

__global__ __launch_bounds__(256) void k_pre(const hb_sweep_in *__restrict__ pin, pre_view v)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.m_pad) return;
    const int kp = v.kpad;
    const bool active = (j < v.m) && (v.vx[j] != 0.0);
    if (!active) {
        for (int c = 0; c < kp; c++) {
            v.thr[(int64_t)c * v.m_pad + j] = HB_INF;
            v.invv[(int64_t)c * v.m_pad + j] = 0.0;
            v.sdz[(int64_t)c * v.m_pad + j] = 0.0;
        }
        return;
    }
    const int model = pin->model_index;
    const double vare = pin->vare;
    const uint64_t sub = hb_sub(HB_PURPOSE_MARKER, (uint64_t)pin->iter);
    const uint64_t base = (uint64_t)(v.m_offset + j) * HB_BLK_PER_MARKER;
    const double xx = v.xpx[j];
    const double gold = v.g[j];
    const double z = hb_normal_blk(v.seed, sub, base + 1);

    if (model == 6) {
        const int K = pin->n_fold;
        const double U = hb_uniform_blk(v.seed, sub, base + 0);
        const double logT = log((1.0 - U) / U);
        double a[HB_MAX_FOLD], b[HB_MAX_FOLD];
        a[0] = pin->logpi[0];
        b[0] = 0.0;
        const double lhs = xx / vare;
        for (int c = 1; c < K; c++) {
            const double vf = pin->vara_fold[c];
            const double vv = xx + vare / vf; // :761, :784
            a[c] = -0.5 * log(vf * lhs + 1.0) + pin->logpi[c];
            b[c] = 0.5 / (vv * vare);
            v.invv[(int64_t)(c - 1) * v.m_pad + j] = 1.0 / vv;
            v.sdz[(int64_t)(c - 1) * v.m_pad + j] = sqrt(vare / vv) * z;
        }
        double prev = 0.0;
        for (int c = 0; c < K - 1; c++) { // boundaries are nested: thr_0 <= thr_1 <= ...
            prev = fmax(prev, bayesr_threshold(K, c, a, b, logT));
            v.thr[(int64_t)c * v.m_pad + j] = prev;
        }
        for (int c = K - 1; c < kp; c++) {
            v.thr[(int64_t)c * v.m_pad + j] = HB_INF;
            v.invv[(int64_t)c * v.m_pad + j] = 0.0;
            v.sdz[(int64_t)c * v.m_pad + j] = 0.0;
        }
        return;
    }

    double varg = pin->varg;
    if (model == 2 || model == 3) { // per-marker variance, :613 / :636 — drawn from g of the previous sweep
        hb_stream st(v.seed, sub, base + 4);
        varg = (gold * gold + pin->s2varg_df) / st.chisq(pin->dfvara + 1.0);
    }
    double vv;
    if (model == 5) vv = xx + 1.0 / v.vargL[j]; // :726
    else vv = xx + vare / varg;                 // :595, :617, :648, :691
    double thr = -HB_INF;
    if (model == 3 || model == 4) {
        const double U = hb_uniform_blk(v.seed, sub, base + 0);
        const double logdetV = log(varg * (xx / vare) + 1.0);
        thr = 2.0 * vv * vare * (log((1.0 - U) / U) + 0.5 * logdetV - pin->logpi[1] + pin->logpi[0]);
        if (thr != thr) thr = HB_INF; // inf - inf when both log(pi) are -inf: never include
    }
    v.thr[j] = thr;
    v.invv[j] = 1.0 / vv;
    v.sdz[j] = sqrt(vare / vv) * z;
    for (int c = 1; c < kp; c++) {
        v.thr[(int64_t)c * v.m_pad + j] = HB_INF;
        v.invv[(int64_t)c * v.m_pad + j] = 0.0;
        v.sdz[(int64_t)c * v.m_pad + j] = 0.0;
    }
}

