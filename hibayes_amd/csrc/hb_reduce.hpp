// hb_reduce.hpp — end-of-sweep reductions (src/Bayes.cpp:819, :823), BayesL's per-marker variances, GWAS windows.
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

// ---------------------------------------------------------------------------------------------
// end-of-sweep reductions behind src/Bayes.cpp:819 (var(u), N-1, two-pass like arma::var) and
// :823 (yadj.yadj); also sum(yadj) for the next intercept draw (:480). One workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_reduce_ru(const double *__restrict__ r, const double *__restrict__ u,
                                                    int n, double *__restrict__ acc)
{
    __shared__ double red[16];
    // (eight loads of each array in flight per thread — the plain loop waited for every one of its 49 — and the same additions in the same
    // order: the sums are bit for bit what they were. It stayed at 25 us at n = 50 000 — one compute unit draws ~45 GB/s from HBM whatever it keeps in
    // flight; the same per-thread sums spread over sixteen workgroups and one final tree would be the same numbers at a tenth of the time)
    double sr = 0, sr2 = 0, su = 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * (int)blockDim.x) {
        double a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + k * (int)blockDim.x;
            a[k] = i < n ? r[i] : 0.0;
            b[k] = i < n ? u[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + k * (int)blockDim.x < n) {
                sr += a[k];
                sr2 = fma(a[k], a[k], sr2);
                su += b[k];
            }
        }
    }
    sr = block_sum(sr, red);
    sr2 = block_sum(sr2, red);
    su = block_sum(su, red);
    const double mean = su / n;
    double a2 = 0, a3 = 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * (int)blockDim.x) {
        double b[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + k * (int)blockDim.x;
            b[k] = i < n ? u[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + k * (int)blockDim.x < n) {
                const double d = mean - b[k];
                a2 = fma(d, d, a2);
                a3 += d;
            }
        }
    }
    a2 = block_sum(a2, red);
    a3 = block_sum(a3, red);
    if (threadIdx.x == 0) {
        acc[HB_ACC_SUMR] = sr;
        acc[HB_ACC_SUMR2] = sr2;
        acc[HB_ACC_VARU] = n > 1 ? (a2 - a3 * a3 / n) / (n - 1) : 0.0;
    }
}

// BayesL: vargL_j <- 1 / InvGauss(sqrt(vare) lambda / |g_j|, lambda^2), src/Bayes.cpp:729-730
__global__ __launch_bounds__(256) void k_bayesl_post(const hb_sweep_in *__restrict__ pin, int m, int64_t m_offset,
                                                     uint64_t seed, const double *__restrict__ vx,
                                                     const double *__restrict__ g, double *__restrict__ vargL, int strict)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m || vx[j] == 0.0) return;
    const uint64_t sub = hb_sub(HB_PURPOSE_MARKER, (uint64_t)pin->iter);
    hb_stream st(seed, sub, (uint64_t)(m_offset + j) * HB_BLK_PER_MARKER + 2);
    const double vargi = 1.0 / st.invgauss(sqrt(pin->vare) * pin->lambda / fabs(g[j]), pin->lambda2);
    // (src/Bayes.cpp:730 keeps vargi >= 0, src/SBayesD.cpp:377 only vargi > 0: `strict` is the summary-level rule)
    if (strict ? vargi > 0.0 : vargi >= 0.0) vargL[j] = vargi;
}

__global__ __launch_bounds__(1024) void k_sum_vec(const double *__restrict__ x, int n, double *__restrict__ out)
{
    __shared__ double red[16];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s;
}

__global__ void k_windows(uint8_t *__restrict__ wflag, double *__restrict__ wppa, int nw)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    wppa[w] += (double)wflag[w];
    wflag[w] = 0;
}

