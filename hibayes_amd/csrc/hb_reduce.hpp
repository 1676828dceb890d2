// hb_reduce.hpp — end-of-sweep reductions (src/Bayes.cpp:819, :823), BayesL's per-marker variances, GWAS windows.
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

// ---------------------------------------------------------------------------------------------
// end-of-sweep reductions behind src/Bayes.cpp:819 (var(u), N-1, two-pass like arma::var) and
// :823 (yadj.yadj); also sum(yadj) for the next intercept draw (:480). One workgroup.
// ---------------------------------------------------------------------------------------------
// Sixteen workgroups of one wave instead of one workgroup of sixteen (round 5: one compute unit draws ~45 GB/s from HBM, and the sums run at the end of
// every sweep with nothing beside them: 25 us at n = 50 000, 14 now — the rest is the hand-overs below), with the SAME numbers bit for bit: thread (b, lane) adds the elements b * 64 + lane,
// + 1024, + 2048 ... in that order, as thread b * 64 + lane of the one workgroup did; a wave's 64 partial sums are folded by the same shuffles;
// the sixteen wave sums are added in wave order by the workgroup that arrives last (a ticket), which also publishes the mean for the second
// pass — the others wait for it (sixteen waves are always resident together) — and, after the second pass, resets the tickets.
// Round 6 (advisor finding): that wait is BOUNDED like every other hand-off of the library (HB_TIMEOUT_TICKS; on expiry the abort flag is
// raised, fetch_acc reports HB_ERR_ABORTED and hb_run_step restores and replays), the mean and its ticket are published by memory-side
// exchanges and polled by memory-side fetch-ors (the point all XCDs share: neither a writer's nor a reader's L2 can hold them back,
// DESIGN 9.1), and the tickets are reset by the last workgroup whether or not the sweep was aborted (hb_ctx_restore clears them too).
// ws: [0..15] sum r, [16..31] sum r^2, [32..47] sum u (then: sum d^2), [48..63] sum d, [64] mean; counters at ws + 80 (three unsigned, zero at rest).
__global__ __launch_bounds__(64) void k_reduce_ru(const double *__restrict__ r, const double *__restrict__ u, int n, double *__restrict__ acc,
                                                  double *__restrict__ ws, unsigned *__restrict__ flags)
{
    const int lane = threadIdx.x, b = blockIdx.x, nb = gridDim.x, stride = nb * 64;
    unsigned *cnt = reinterpret_cast<unsigned *>(ws + 80);
    double sr = 0, sr2 = 0, su = 0;
    // (a pass is a chain of dependent memory round trips, ~1 us each: 28 elements of both arrays in flight per thread make it two at n = 50 000)
    constexpr int B1 = 28, B2 = 56;
    for (int i0 = b * 64 + lane; i0 < n; i0 += B1 * stride) {
        double av[B1], bv[B1];
#pragma unroll
        for (int k = 0; k < B1; k++) {
            const int i = i0 + k * stride;
            av[k] = i < n ? r[i] : 0.0;
            bv[k] = i < n ? u[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < B1; k++) {
            if (i0 + k * stride < n) {
                sr += av[k];
                sr2 = fma(av[k], av[k], sr2);
                su += bv[k];
            }
        }
    }
    sr = wave_sum(sr);
    sr2 = wave_sum(sr2);
    su = wave_sum(su);
    __shared__ int s_last;
    if (lane == 0) {
        (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(&ws[b]), (unsigned long long)__double_as_longlong(sr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(&ws[16 + b]), (unsigned long long)__double_as_longlong(sr2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(&ws[32 + b]), (unsigned long long)__double_as_longlong(su), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(&cnt[0], 1u) == (unsigned)nb - 1u;
    }
    __syncthreads();
    if (s_last) { // (the whole wave: lane i fetches wave i's sums — one round trip, not forty-eight —, lane 0 adds them in wave order)
        __threadfence();
        const int li = min(lane, nb - 1);
        const double v0 = ld_fresh(&ws[li]);
        const double v1 = ld_fresh(&ws[16 + li]);
        const double v2 = ld_fresh(&ws[32 + li]);
        double t0 = 0, t1 = 0, t2 = 0;
        for (int i = 0; i < nb; i++) {
            t0 += readlane_f64(v0, i);
            t1 += readlane_f64(v1, i);
            t2 += readlane_f64(v2, i);
        }
        if (lane == 0) {
            acc[HB_ACC_SUMR] = t0;
            acc[HB_ACC_SUMR2] = t1;
            (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(&ws[64]), (unsigned long long)__double_as_longlong(t2 / n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            (void)__hip_atomic_exchange(&cnt[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lane == 0 && !s_last) {
        const unsigned long long t0 = wall_clock64();
        unsigned looks = 0;
        while (__hip_atomic_fetch_or(&cnt[1], 0u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(2);
            if ((++looks & 63u) == 0u && (ld_flag(flags + HB_FLAG_ABORT) != 0u || wall_clock64() - t0 > HB_TIMEOUT_TICKS)) {
                if (ld_flag(flags + HB_FLAG_ABORT) == 0u) hb_abort_log(flags, HB_LOG_WAIT_GE, true, 0xA11u, (unsigned)b, 0ull);
                st_flag(flags + HB_FLAG_ABORT, 1u); // (the sums of an aborted sweep are never used: fetch_acc fails first)
                break;
            }
        }
    }
    __syncthreads();
    const double mean = ld_fresh(&ws[64]);
    double a2 = 0, a3 = 0;
    for (int i0 = b * 64 + lane; i0 < n; i0 += B2 * stride) {
        double bv[B2];
#pragma unroll
        for (int k = 0; k < B2; k++) {
            const int i = i0 + k * stride;
            bv[k] = i < n ? u[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < B2; k++) {
            if (i0 + k * stride < n) {
                const double d = mean - bv[k];
                a2 = fma(d, d, a2);
                a3 += d;
            }
        }
    }
    a2 = wave_sum(a2);
    a3 = wave_sum(a3);
    __syncthreads(); // (s_last is rewritten: everybody has read it)
    if (lane == 0) {
        (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(&ws[32 + b]), (unsigned long long)__double_as_longlong(a2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(&ws[48 + b]), (unsigned long long)__double_as_longlong(a3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(&cnt[2], 1u) == (unsigned)nb - 1u; // (everybody has read the mean and left its sums: the last one closes)
    }
    __syncthreads();
    if (s_last) {
        __threadfence();
        const int li = min(lane, nb - 1);
        const double v2 = ld_fresh(&ws[32 + li]);
        const double v3 = ld_fresh(&ws[48 + li]);
        double t2 = 0, t3 = 0;
        for (int i = 0; i < nb; i++) {
            t2 += readlane_f64(v2, i);
            t3 += readlane_f64(v3, i);
        }
        if (lane == 0) {
            acc[HB_ACC_VARU] = n > 1 ? (t2 - t3 * t3 / n) / (n - 1) : 0.0;
            cnt[0] = 0u;
            cnt[1] = 0u;
            cnt[2] = 0u;
        }
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

