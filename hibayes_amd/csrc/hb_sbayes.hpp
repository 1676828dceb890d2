// hb_sbayes.hpp — device side of the summary-level sampler on a dense LD matrix (reference src/SBayesD.cpp:251-470).
// Included by hb_kernels.hip (it reuses k_pre — the same six conditionals restated as thresholds on q = rhs^2 — k_bayesl_post,
// the wave helpers and the Philox layer); the host loop is hb_sbayes.hip.
//
// The reference keeps the right-hand sides of ALL markers in Gram space: r_hat = xy - n ldm g, and after marker i moves
// r_hat += n (g_old - g_new) ldm[:, i] (an m-long daxpy, :262-266) before marker i + 1 reads r_hat[i + 1]. Here a sweep runs in
// groups of SB_GS = 512 consecutive markers, two kernels per group:
//   k_sb_group   ONE workgroup, thread = marker of the group, the scheme of k_chain_group with the LD matrix as its Gram matrix:
//                rounds of up to 64 CANDIDATES (in the model, or q over the entry threshold — every marker for RR / A / L) in marker
//                order; their mutual LD entries gathered into LDS; the exact serial chain on wave 0 (one candidate per lane, lane
//                k's change broadcast, the later lanes take n (g_old - g_new) ldm[lane][k]); the round's moves folded onto every
//                later marker of the group (one coalesced column segment per move); a marker the round passed over that is pushed
//                over its threshold joins the candidates and the round is repeated — the result is always the sequential chain.
//   k_sb_update  all compute units: r_hat[j] += sum over the group's moves of n (g_old - g_new)_k ldm[j][k] for EVERY j (the
//                group's own markers included: k_sb_group leaves r_hat alone), the columns read coalesced down the rows.
// The kernel boundary is the grid barrier; a sweep is 2 ceil(m / 512) + 3 launches replayed from one captured graph (one wave and
// 64 markers per launch, the first version, was latency-bound: 35 us per 64 markers, slower than the CPU oracle. Measured and
// dropped: the update of the next group's rows first and the other rows on a second stream beside the next chain launch — the
// cross-stream edges of the graph cost more than the overlap gives, 2.44 against 2.10 ms per BayesCpi sweep at m = 20000).
// Same chain as the reference in exact arithmetic; in floating point the corrections inside a group are summed in another order
// than the reference's daxpy sequence (last bits), like every blocked path of this library.
#pragma once

#define SB_GS 512 // markers per k_sb_group launch (= its workgroup size)
#define SB_CH 32  // moves whose column segments a thread requests together when a round is folded onto the later markers

struct sb_view {
    int m, m_pad, n;
    const double *ldm; // m x m column-major
    double *r_hat;
    const double *xy;
    double *g;
    const double *xpx, *vx; // vx[i] != 0 <=> marker i has summary statistics (ifest, SBayesD.cpp:99-112)
    const double *thr, *invv, *sdz;
    uint8_t *tracker;
    uint32_t *nzrate;
    int *ev_n;      // [1] moves of the current group
    int *ev_col;    // [SB_GS] their columns
    double *ev_gi;  // [SB_GS] n (g_old - g_new)
    const uint32_t *wind;
    uint8_t *wflag;
    double *acc;    // HB_ACC_N sums
};

template <int K1>
__global__ __launch_bounds__(SB_GS) void k_sb_group(const hb_sweep_in *__restrict__ pin, sb_view v, int g0)
{
    __shared__ double cs_d[(2 + 3 * K1) * 64]; // the round's candidates by rank: rhs, g_old, thr[K1], 1/v [K1], sd z [K1]
    __shared__ double cg[64 * 64];             // cg[k][c] = ldm[c][k] for k < c (candidate ranks), zero elsewhere
    __shared__ double res_g[64], ev_del[64], red[SB_GS / 64];
    __shared__ int cs_pos[64], res_c[64], ev_pos[64], wcnt[SB_GS / 64], misc[4];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, i = g0 + t;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool in = i < v.m;
    const int ic = min(i, v.m - 1); // (loads of a thread past the end go to an address that exists)
    const int model = pin->model_index;
    // (every load of the opening is unconditional, on a clamped index: one round trip, the selects afterwards)
    const double vxi = v.vx[ic], gi0 = v.g[ic], xx = v.xpx[ic];
    double thr[K1], invv[K1], sdz[K1];
#pragma unroll
    for (int c = 0; c < K1; c++) {
        thr[c] = v.thr[(size_t)c * v.m_pad + ic];
        invv[c] = v.invv[(size_t)c * v.m_pad + ic];
        sdz[c] = v.sdz[(size_t)c * v.m_pad + ic];
    }
    double r0 = v.r_hat[ic]; // the marker's right-hand side with every move BEFORE the current round applied (without xx g_old)
    const bool active = in && vxi != 0.0;
    const double gold = in ? gi0 : 0.0;
#pragma unroll
    for (int c = 0; c < K1; c++) thr[c] = active ? thr[c] : HB_INF;
    const int gend = min(SB_GS, v.m - g0);
    const double nn = (double)v.n;
    const double *blk = v.ldm + (size_t)g0 * v.m + g0; // the group's diagonal block: blk[k m + c] = ldm[g0 + c][g0 + k]
    int pos_lo = 0, nev_total = 0;
    bool forced = false, decided = false;
    int my_cls = 0;
    double my_gn = 0.0;
    for (;;) {
        // ---- the round's candidates, ranked in marker order ----
        const bool isc = active && t >= pos_lo && (gold != 0.0 || forced || r0 * r0 >= thr[0]);
        const unsigned long long cm = __ballot(isc);
        if (lane == 0) wcnt[wave] = __popcll(cm);
        if (t == 0) misc[1] = gend;
        __syncthreads(); // B1
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < SB_GS / 64; w++) {
            const int c = wcnt[w];
            before += (w < wave) ? c : 0;
            total += c;
        }
        if (total == 0) break; // nobody (left) in the group can move
        const int rank = before + __popcll(cm & lt), ncr = min(total, 64);
        if (isc && rank == 64) misc[1] = t; // the round ends before the 65th candidate
        if (isc && rank < 64) {
            cs_d[rank] = (gold != 0.0) ? fma(xx, gold, r0) : r0; // :257 / :298 / :333 ...: rhs = r_hat[i] (+ xx g when g != 0)
            cs_d[64 + rank] = gold;
#pragma unroll
            for (int c = 0; c < K1; c++) {
                cs_d[(2 + c) * 64 + rank] = thr[c];
                cs_d[(2 + K1 + c) * 64 + rank] = invv[c];
                cs_d[(2 + 2 * K1 + c) * 64 + rank] = sdz[c];
            }
            cs_pos[rank] = t;
        }
        __syncthreads(); // B2
        const int pos_hi = misc[1];
        // ---- LD entries among the round's candidates (all of a thread's entries requested before the first is stored) ----
        {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = u * SB_GS + t, k = idx >> 6, c = idx & 63;
                const bool need = k < c && c < ncr;
                const int pk = cs_pos[min(k, ncr - 1)], pc = cs_pos[min(c, ncr - 1)];
                x[u] = blk[(size_t)pk * v.m + pc]; // (an entry that is not needed re-reads one that is: no branch around the load)
                x[u] = need ? x[u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) cg[u * SB_GS + t] = x[u];
        }
        __syncthreads(); // B3
        // ---- the exact serial chain over the round's candidates: wave 0, one candidate per lane, in marker order ----
        if (wave == 0) {
            const bool lv = lane < ncr;
            double crhs = lv ? cs_d[lane] : 0.0;
            const double cgold = lv ? cs_d[64 + lane] : 0.0;
            double cthr[K1], cinvv[K1], csdz[K1];
#pragma unroll
            for (int c = 0; c < K1; c++) {
                cthr[c] = lv ? cs_d[(2 + c) * 64 + lane] : HB_INF;
                cinvv[c] = lv ? cs_d[(2 + K1 + c) * 64 + lane] : 0.0;
                csdz[c] = lv ? cs_d[(2 + 2 * K1 + c) * 64 + lane] : 0.0;
            }
            const int cp = lv ? cs_pos[lane] : 0;
            auto decide = [&](double rhsv, int &cls, double &gn) {
                const double q = rhsv * rhsv;
                double gsel = fma(rhsv, cinvv[0], csdz[0]);
                cls = q >= cthr[0] ? 1 : 0;
#pragma unroll
                for (int c = 1; c < K1; c++) {
                    const bool ge = q >= cthr[c];
                    cls += ge ? 1 : 0;
                    gsel = ge ? fma(rhsv, cinvv[c], csdz[c]) : gsel;
                }
                gn = (q >= cthr[0]) ? gsel : 0.0;
                if (K1 == 1 && model == 5 && fabs(gn) < 1e-6) gn = 1e-6; // :376
            };
            double rnext = cg[lane]; // row k of cg, one step ahead
            for (int k = 0; k < ncr; k++) {
                const double rcur = rnext;
                rnext = cg[min(k + 1, ncr - 1) * 64 + lane];
                int cls;
                double gn;
                decide(crhs, cls, gn);
                const double gk = readlane_f64((cgold - gn) * nn, k); // :262 gi_ = (g[i] - gi) * n
                crhs = fma(gk, rcur, crhs); // r_hat[lane] += gi_ ldm[lane][k] (row k is zero at and before lane k)
            }
            int cls;
            double gn;
            decide(crhs, cls, gn); // lane k's rhs was not touched after its own step
            const double gi_ = lv ? (cgold - gn) * nn : 0.0;
            const unsigned long long moved = __ballot(gi_ != 0.0);
            if (gi_ != 0.0) {
                const int pos = __popcll(moved & lt);
                ev_pos[pos] = cp;
                ev_del[pos] = gi_;
            }
            res_c[lane] = cls;
            res_g[lane] = gn;
            if (lane == 0) misc[0] = __popcll(moved);
        }
        __syncthreads(); // B4
        const int nmoves = misc[0];
        // ---- the round's moves onto the later markers of the group: one coalesced column segment per move, SB_CH in flight ----
        double rnew = r0;
        for (int e0 = 0; e0 < nmoves; e0 += SB_CH) {
            double x[SB_CH];
            int pe[SB_CH];
#pragma unroll
            for (int q = 0; q < SB_CH; q++) {
                pe[q] = __builtin_amdgcn_readfirstlane(ev_pos[min(e0 + q, nmoves - 1)]);
                x[q] = blk[(size_t)pe[q] * v.m + min(t, gend - 1)];
            }
#pragma unroll
            for (int q = 0; q < SB_CH; q++)
                if (e0 + q < nmoves && t > pe[q]) rnew = fma(ev_del[min(e0 + q, 63)], x[q], rnew); // in move order, as the reference's daxpy sequence
        }
        // ---- did every marker the round passed over really stay below its threshold? ----
        const bool viol = active && !isc && t >= pos_lo && t < pos_hi && rnew * rnew >= thr[0];
        if (__syncthreads_or(viol ? 1 : 0)) { // B5: roll the round back, the markers that crossed join the candidates
            forced = forced || viol;
            continue;
        }
        // ---- commit the round ----
        r0 = rnew;
        if (isc && rank < 64) {
            decided = true;
            my_cls = res_c[rank];
            my_gn = res_g[rank];
        }
        if (wave == 0 && lane < nmoves) {
            v.ev_col[nev_total + lane] = g0 + ev_pos[lane];
            v.ev_gi[nev_total + lane] = ev_del[lane];
        }
        nev_total += nmoves;
        pos_lo = pos_hi;
        if (pos_lo >= gend) break;
    }
    // ---- the group's outcome ----
    if (!active || !decided) { my_cls = 0; my_gn = 0.0; }
    if (in) {
        if (my_gn != gold) v.g[i] = my_gn;
        v.tracker[i] = (uint8_t)my_cls;
        if (pin->count_pip && my_cls != 0) {
            v.nzrate[i] += 1u;
            if (v.wind) v.wflag[v.wind[i] - 1u] = 1;
        }
    }
    if (t == 0) *v.ev_n = nev_total;
    // sums the hyper-parameter draws need: g.g (RR :269), sum g^2 of the included (C :349), sum g^2 / fold[class] (R :443), class counts
    double w = 0.0;
    if (active && my_cls > 0) w = (model == 6) ? my_gn * my_gn / pin->fold[my_cls] : my_gn * my_gn;
    const double ws = block_sum(w, red);
    if (t == 0 && ws != 0.0) v.acc[HB_ACC_SUMG2] += ws; // (one chain kernel at a time: no atomics needed)
#pragma unroll
    for (int c = 0; c <= K1; c++) {
        const double cnt = block_sum((active && my_cls == c) ? 1.0 : 0.0, red);
        if (t == 0 && cnt != 0.0 && c < HB_MAX_FOLD) v.acc[HB_ACC_COUNT0 + c] += cnt;
    }
    if (t == 0) v.acc[HB_ACC_EVENTS] += (double)nev_total;
}

// r_hat[j] += sum_e gi_e ldm[j][col_e] for every j; thread = row j, the columns are wave-uniform
__global__ __launch_bounds__(64) void k_sb_update(sb_view v)
{
    __shared__ int s_col[64];
    __shared__ double s_gi[64];
    const int nev = *v.ev_n;
    if (nev == 0) return; // (uniform)
    const int t = threadIdx.x, j = blockIdx.x * 64 + t, jc = min(j, v.m - 1);
    double a = v.r_hat[jc];
    for (int c0 = 0; c0 < nev; c0 += 64) {
        const int nc = min(64, nev - c0);
        __syncthreads();
        if (t < nc) {
            s_col[t] = v.ev_col[c0 + t];
            s_gi[t] = v.ev_gi[c0 + t];
        }
        __syncthreads();
        for (int e0 = 0; e0 < nc; e0 += 16) {
            double x[16];
#pragma unroll
            for (int q = 0; q < 16; q++) x[q] = v.ldm[(size_t)s_col[min(e0 + q, nc - 1)] * v.m + jc];
#pragma unroll
            for (int q = 0; q < 16; q++) a = (e0 + q < nc) ? fma(s_gi[e0 + q], x[q], a) : a; // in move order, as the reference's daxpy sequence
        }
    }
    if (j < v.m) v.r_hat[j] = a;
}

// end of sweep: g . (xy - r_hat) and g . (xy + r_hat) (:466-474), sum of vargL (BayesL :388). One workgroup.
__global__ __launch_bounds__(1024) void k_sb_reduce(sb_view v, const double *__restrict__ vargL, int want_vargl)
{
    __shared__ double red[16];
    double s1 = 0, s2 = 0, s3 = 0;
    for (int i = threadIdx.x; i < v.m; i += blockDim.x) {
        const double gi = v.g[i], x = v.xy[i], r = v.r_hat[i];
        s1 = fma(gi, x - r, s1);
        s2 = fma(gi, x + r, s2);
        if (want_vargl) s3 += vargL[i];
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    s3 = block_sum(s3, red);
    if (threadIdx.x == 0) {
        v.acc[HB_ACC_SUMR] = s1;
        v.acc[HB_ACC_SUMR2] = s2;
        v.acc[HB_ACC_SUMVARGL] = s3;
    }
}

// ---- launchers (hb_sbayes.hip owns the buffers: struct hb_sb_dev, hb_internal.hpp) ----
int hbk_sb_enqueue_sweep(hb_sb_dev *d, int model, int n_fold)
{
    const int kp = kpad_for(model, n_fold);
    sb_view v{d->m, d->m_pad, d->n, d->ldm, d->r_hat, d->xy, d->g, d->xpx, d->vx, d->thr, d->invv, d->sdz, d->tracker, d->nzrate,
              d->ev_n, d->ev_col, d->ev_gi, d->wind, d->wflag, d->acc};
    HB_HIP(hipMemsetAsync(d->acc, 0, sizeof(double) * HB_ACC_N, d->stream));
    pre_view pv{d->m, d->m_pad, 0, d->seed, d->xpx, d->vx, d->g, d->vargL, d->thr, d->invv, d->sdz, kp};
    hipLaunchKernelGGL(k_pre, dim3((d->m_pad + 255) / 256), dim3(256), 0, d->stream, d->d_in, pv);
    const int upd_blocks = (d->m + 63) / 64;
    for (int g0 = 0; g0 < d->m; g0 += SB_GS) {
        if (kp == 1) hipLaunchKernelGGL(k_sb_group<1>, dim3(1), dim3(SB_GS), 0, d->stream, d->d_in, v, g0);
        else if (kp == 3) hipLaunchKernelGGL(k_sb_group<3>, dim3(1), dim3(SB_GS), 0, d->stream, d->d_in, v, g0);
        else hipLaunchKernelGGL(k_sb_group<7>, dim3(1), dim3(SB_GS), 0, d->stream, d->d_in, v, g0);
        hipLaunchKernelGGL(k_sb_update, dim3(upd_blocks), dim3(64), 0, d->stream, v);
    }
    if (model == 5) // vargL_i <- 1 / InvGauss(sqrt(vare) lambda / |g_i|, lambda^2), :377-378 (the marker's own stream: order-free)
        hipLaunchKernelGGL(k_bayesl_post, dim3((d->m + 255) / 256), dim3(256), 0, d->stream, d->d_in, d->m, (int64_t)0, d->seed, d->vx, d->g, d->vargL, 1);
    hipLaunchKernelGGL(k_sb_reduce, dim3(1), dim3(1024), 0, d->stream, v, d->vargL, model == 5 ? 1 : 0);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_sb_windows(hb_sb_dev *d)
{
    if (d->nw) hipLaunchKernelGGL(k_windows, dim3((d->nw + 255) / 256), dim3(256), 0, d->stream, d->wflag, d->wppa, d->nw);
    HB_HIP(hipGetLastError());
    return HB_OK;
}
