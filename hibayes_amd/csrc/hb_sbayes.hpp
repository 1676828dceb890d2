// hb_sbayes.hpp — device side of the summary-level sampler on a dense LD matrix (reference src/SBayesD.cpp:251-470).
// Included by hb_kernels.hip (it reuses k_pre — the same six conditionals restated as thresholds on q = rhs^2 — k_bayesl_post,
// the wave helpers and the Philox layer); the host loop is hb_sbayes.hip.
//
// The reference keeps the right-hand sides of ALL markers in Gram space: r_hat = xy - n ldm g, and after marker i moves
// r_hat += n (g_old - g_new) ldm[:, i] (an m-long daxpy, :262-266) before marker i + 1 reads r_hat[i + 1]. Here a sweep runs in
// blocks of 64 consecutive markers, two kernels per block:
//   k_sb_chain   ONE wave, lane = marker of the block: the exact serial chain over the block with the 64 x 64 LD sub-block in
//                LDS (step k: every lane evaluates its draw from its running rhs, lane k's change is broadcast, the later lanes
//                take n (g_old - g_new) ldm[lane][k]) — the chain kernel of the individual-level path with the LD block as its Gram matrix;
//   k_sb_update  all compute units: r_hat[j] += sum over the block's moves of n (g_old - g_new)_k ldm[j][k] for EVERY j (the
//                block's own markers included: k_sb_chain leaves r_hat alone), a column slab read once, coalesced down the rows.
// The kernel boundary is the grid barrier; a sweep is 2 ceil(m / 64) + 3 launches replayed from one captured graph.
// Same chain as the reference in exact arithmetic; in floating point the corrections inside a block are summed in another order
// than the reference's daxpy sequence (last bits), like every blocked path of this library.
#pragma once

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
    int *ev_n;      // [1] moves of the current block
    int *ev_col;    // [64] their columns
    double *ev_gi;  // [64] n (g_old - g_new)
    const uint32_t *wind;
    uint8_t *wflag;
    double *acc;    // HB_ACC_N sums
};

template <int K1>
__global__ __launch_bounds__(64) void k_sb_chain(const hb_sweep_in *__restrict__ pin, sb_view v, int b0)
{
    __shared__ double L[64][65]; // L[k][lane] = ldm[b0 + lane][b0 + k] (row k of the steps: what lane takes when k moves)
    const int lane = threadIdx.x, i = b0 + lane;
    const bool in = i < v.m;
    const int model = pin->model_index;
    // the LD sub-block: column b0 + k of ldm, rows b0 .. b0 + 63 — one coalesced 512-byte read per k, all in flight
#pragma unroll 8
    for (int k = 0; k < 64; k++) L[k][lane] = (in && b0 + k < v.m) ? v.ldm[(size_t)(b0 + k) * v.m + i] : 0.0;
    const bool active = in && v.vx[i] != 0.0;
    const double gold = in ? v.g[i] : 0.0, xx = in ? v.xpx[i] : 0.0;
    double thr[K1], invv[K1], sdz[K1];
#pragma unroll
    for (int c = 0; c < K1; c++) {
        thr[c] = active ? v.thr[(size_t)c * v.m_pad + i] : HB_INF;
        invv[c] = active ? v.invv[(size_t)c * v.m_pad + i] : 0.0;
        sdz[c] = active ? v.sdz[(size_t)c * v.m_pad + i] : 0.0;
    }
    double rhs = in ? v.r_hat[i] : 0.0;
    if (gold != 0.0) rhs = fma(xx, gold, rhs); // :257 / :298 / :333 ...: rhs = r_hat[i] (+ xx g when g != 0)
    auto decide = [&](double rhsv, int &cls, double &gn) {
        const double q = rhsv * rhsv;
        double gsel = fma(rhsv, invv[0], sdz[0]);
        cls = q >= thr[0] ? 1 : 0;
#pragma unroll
        for (int c = 1; c < K1; c++) {
            const bool ge = q >= thr[c];
            cls += ge ? 1 : 0;
            gsel = ge ? fma(rhsv, invv[c], sdz[c]) : gsel;
        }
        gn = (q >= thr[0]) ? gsel : 0.0;
        if (K1 == 1 && model == 5 && fabs(gn) < 1e-6) gn = 1e-6; // :376
    };
    const double nn = (double)v.n;
    __syncthreads();
    for (int k = 0; k < 64; k++) {
        int cls;
        double gn;
        decide(rhs, cls, gn);
        const double gi_ = active ? (gold - gn) * nn : 0.0; // :262 gi_ = (g[i] - gi) * n
        const double gk = readlane_f64(gi_, k);
        if (gk != 0.0 && lane > k) rhs = fma(gk, L[k][lane], rhs); // (uniform branch) r_hat[lane] += gi_ ldm[lane][k]
    }
    int cls;
    double gn;
    decide(rhs, cls, gn); // lane k's rhs was not touched after its own step
    if (!active) { cls = 0; gn = 0.0; }
    const double gi_ = active ? (gold - gn) * nn : 0.0;
    const unsigned long long moved = __ballot(gi_ != 0.0);
    if (gi_ != 0.0) {
        const int pos = __popcll(moved & ((1ull << lane) - 1ull));
        v.ev_col[pos] = i;
        v.ev_gi[pos] = gi_;
    }
    if (lane == 0) *v.ev_n = __popcll(moved);
    if (in) {
        if (gn != gold) v.g[i] = gn;
        v.tracker[i] = (uint8_t)cls;
        if (pin->count_pip && cls != 0) {
            v.nzrate[i] += 1u;
            if (v.wind) v.wflag[v.wind[i] - 1u] = 1;
        }
    }
    // sums the hyper-parameter draws need: g.g (RR :269), sum g^2 of the included (C :349), sum g^2 / fold[class] (R :443), class counts
    double w = 0.0;
    if (active && cls > 0) w = (model == 6) ? gn * gn / pin->fold[cls] : gn * gn;
    const double ws = wave_sum(w);
    if (lane == 0 && ws != 0.0) v.acc[HB_ACC_SUMG2] += ws; // (one chain kernel at a time: no atomics needed)
#pragma unroll
    for (int c = 0; c <= K1; c++) {
        const int cnt = __popcll(__ballot(active && cls == c));
        if (lane == 0 && cnt && c < HB_MAX_FOLD) v.acc[HB_ACC_COUNT0 + c] += (double)cnt;
    }
    if (lane == 0) v.acc[HB_ACC_EVENTS] += (double)__popcll(moved);
}

// r_hat[j] += sum_e gi_e ldm[j][col_e] for every j; thread = row j, the columns are wave-uniform
__global__ __launch_bounds__(256) void k_sb_update(sb_view v)
{
    __shared__ int s_col[64];
    __shared__ double s_gi[64];
    const int nev = *v.ev_n;
    if (nev == 0) return; // (uniform)
    if (threadIdx.x < 64 && (int)threadIdx.x < nev) {
        s_col[threadIdx.x] = v.ev_col[threadIdx.x];
        s_gi[threadIdx.x] = v.ev_gi[threadIdx.x];
    }
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= v.m) return;
    double a = v.r_hat[j];
    for (int e0 = 0; e0 < nev; e0 += 8) {
        double x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = v.ldm[(size_t)s_col[min(e0 + q, nev - 1)] * v.m + j];
#pragma unroll
        for (int q = 0; q < 8; q++) a = (e0 + q < nev) ? fma(s_gi[e0 + q], x[q], a) : a; // in move order, as the reference's daxpy sequence
    }
    v.r_hat[j] = a;
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
    const int upd_blocks = (d->m + 255) / 256;
    for (int b0 = 0; b0 < d->m; b0 += 64) {
        if (kp == 1) hipLaunchKernelGGL(k_sb_chain<1>, dim3(1), dim3(64), 0, d->stream, d->d_in, v, b0);
        else if (kp == 3) hipLaunchKernelGGL(k_sb_chain<3>, dim3(1), dim3(64), 0, d->stream, d->d_in, v, b0);
        else hipLaunchKernelGGL(k_sb_chain<7>, dim3(1), dim3(64), 0, d->stream, d->d_in, v, b0);
        hipLaunchKernelGGL(k_sb_update, dim3(upd_blocks), dim3(256), 0, d->stream, v);
    }
    if (model == 5) // vargL_i <- 1 / InvGauss(sqrt(vare) lambda / |g_i|, lambda^2), :377-378 (the marker's own stream: order-free)
        hipLaunchKernelGGL(k_bayesl_post, dim3((d->m + 255) / 256), dim3(256), 0, d->stream, d->d_in, d->m, (int64_t)0, d->seed, d->vx, d->g, d->vargL);
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
