// hb_kernels.hip — gfx950 kernels of the block-Gibbs marker sweep.
//
// One sweep (reference src/Bayes.cpp:586-816) is executed panel by panel; a panel is P
// consecutive markers.  For each panel:
//   k_dot      d = X_p' yadj                  bandwidth-bound int8 mat-vec (the dominant kernel)
//   k_chain    the serial conditional updates of the panel's markers, made exact by the panel
//              Gram matrix G = X_p' X_p:  after marker k moves by D_k, rhs_j -= G[k][j] D_k
//              for every later marker j of the panel  (== what the reference gets by updating
//              yadj with daxpy before the next ddot)
//   k_update   yadj -= X_p[:, changed] D,  u += X_p[:, changed] D
// Per-marker quantities that do not depend on the running rhs (the uniform and normal deviates,
// 1/v, sd*z, and the inclusion test rewritten as thresholds on rhs^2) are produced once per
// sweep by k_pre, so the serial part is a handful of fp64 operations per marker.
#include "hb_internal.hpp"
#include "hb_rng.hpp"
#include <type_traits>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

#define HB_INF __builtin_huge_val()

// ---- device-side flags of the persistent pipeline ----
// Every shared word is accessed with relaxed agent-scope atomics (sc1); payloads are written with 4/8-byte
// agent-scope atomic stores (write-through) and drained with s_waitcnt vmcnt(0) before the flag moves, so no
// release fence is needed; consumers read the payload with agent-scope atomic loads (sc1), so no acquire
// fence either (cdna_hip_programming.md §6 Guideline 16, forms R1 / "sc1 both sides").
#define HB_FLAG_CHAIN_DONE 0
#define HB_FLAG_ABORT 1
#define HB_FLAG_XCC 2               /* 1 + the XCD the chain workgroup runs on (k_warm) */
#define HB_NFLAGS 72                /* words in the flag block that every sweep clears */
// How long a wait inside the pipeline may last before it gives up and aborts the sweep, in ticks of wall_clock64() (100 MHz). A device
// global, set per sweep from hb_ctx.timeout_ms (hbk_set_timeout): 100 ms by default — a healthy hand-off takes microseconds, the
// device's own occasional pauses ~1 ms (§9.0), and an aborted sweep is replayed by hb_run_step, so giving up early is cheap; the
// replay of a sweep runs with 3 s, and a run that aborts repeatedly (a shared or profiled GPU) raises its own default.
__device__ unsigned long long hb_timeout_ticks = 10000000ull;
#define HB_TIMEOUT_TICKS hb_timeout_ticks
// Abort log (diagnostics of a pipeline time-out, read by fetch_acc in hb_ctx.hip): whoever leaves a wait because the sweep is
// being aborted appends one record of 8 words — what it was waiting for, whether the time-out was its own, the clock, the value
// it last saw. flags[HB_FLAG_LOGN] counts the records, they start at flags + HB_LOG_BASE (the flag block has 4096 words).
#define HB_FLAG_LOGN 64
#define HB_LOG_BASE 128
#define HB_LOG_CAP 480
#define HB_LOG_CHAIN_DOT 1    /* k_chain_dense: a = marker index into dsum[], b = panel */
#define HB_LOG_CHAIN_FCORR 2  /* ... into fcorr[] */
#define HB_LOG_CHAIN_FC2 3    /* ... into fcorr2[] */
#define HB_LOG_FOLD_DD 4      /* k_fold_dense: a = index into dd[], b = target panel | step << 16 */
#define HB_LOG_UPD_DENSE 5    /* update_rows_dense: a = first panel of the group, b = block */
#define HB_LOG_WAIT_GE 6      /* wait_ge: a = word, b = value wanted */
#define HB_LOG_GROUP 7        /* k_chain_group / k_fwd / k_chain_persist: a = code, b = panel or group */

__device__ __forceinline__ unsigned ld_flag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(unsigned *p, unsigned v)
{
#if defined(HB_PUBLISH_ATOMIC) && HB_PUBLISH_ATOMIC
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_sc1(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// HB_PUBLISH_ATOMIC (an A/B for the dense stall): publish with a no-return atomic exchange — performed at the memory side,
// the point all XCDs share — instead of a write-through store that the writer's L2 forwards
#ifndef HB_PUBLISH_ATOMIC
#define HB_PUBLISH_ATOMIC 0
#endif
#if HB_PUBLISH_ATOMIC
__device__ __forceinline__ void st_sc1(double *p, double v) { (void)__hip_atomic_exchange(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(int *p, int v) { (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
__device__ __forceinline__ void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

__device__ __attribute__((noinline)) void hb_abort_log(unsigned *flags, unsigned kind, bool own, unsigned a, unsigned b, unsigned long long seen)
{
    const unsigned i = atomicAdd(flags + HB_FLAG_LOGN, 1u);
    if (i >= HB_LOG_CAP) return;
    unsigned *r = flags + HB_LOG_BASE + 8 * i;
    const unsigned long long now = wall_clock64();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    r[0] = kind | (own ? 0x10000u : 0u) | ((xcc & 15u) << 20);
    r[1] = a;
    r[2] = b;
    r[3] = blockIdx.x;
    r[4] = (unsigned)now;
    r[5] = (unsigned)(now >> 32);
    r[6] = (unsigned)seen;
    r[7] = (unsigned)(seen >> 32);
}

// Polling pace. A waiter looks again after a short sleep; HB_BACKOFF builds (an A/B for the dense stall) stretch the
// sleep once a wait has lasted a few hundred looks, so that a long wait stops being continuous traffic on the memory path.
#ifndef HB_BACKOFF
#define HB_BACKOFF 0
#endif
__device__ __forceinline__ void hb_poll_pause(unsigned &looks, int base)
{
#if HB_BACKOFF
    if (looks > 4096u) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
    else if (looks > 256u) __builtin_amdgcn_s_sleep(64);
    else if (base <= 1) __builtin_amdgcn_s_sleep(1);
    else __builtin_amdgcn_s_sleep(8);
#else
    (void)looks;
    if (base <= 1) __builtin_amdgcn_s_sleep(1);
    else __builtin_amdgcn_s_sleep(8);
#endif
}

// A poll that cannot be served a stale line. The hand-offs are polled with agent-scope (sc1) loads, which the XCD's L2 may serve;
// round 4's abort log (profiles/r04_dense_stall_diagnostics.txt) shows what the dense stall of round 3 was: once in ~10^9 polled
// words a reader's L2 keeps returning the sentinel a word was pre-filled with although the producer's write-through store reached
// memory long ago (the reader asked for the line ahead of time, and its copy was never dropped) — every later look hits that copy,
// and the pipeline waits until its 3 s time-out. A returning agent-scope atomic (fetch-or with 0) is performed at the memory side,
// the one place all eight XCDs agree on: it returns what memory holds and leaves it unchanged. Every wait looks that way once in
// HB_FRESH_EVERY looks — a wait that is served at once never pays for it.
#ifndef HB_FRESH_EVERY
#define HB_FRESH_EVERY 0 /* 0: never (the default since the stall turned out to be on the WRITER's side, see hb_long_wait) */
#endif
__device__ __forceinline__ double ld_fresh(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_fetch_or(reinterpret_cast<unsigned long long *>(const_cast<double *>(p)), 0ull,
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ unsigned ld_flag_fresh(const unsigned *p)
{
    return __hip_atomic_fetch_or(const_cast<unsigned *>(p), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_fresh(const int *p)
{
    return (int)__hip_atomic_fetch_or(reinterpret_cast<unsigned *>(const_cast<int *>(p)), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (uniform) is this look of a wait a memory-side one?
__device__ __forceinline__ bool hb_fresh_look(unsigned looks)
{
#if HB_FRESH_EVERY > 0
    return (looks % HB_FRESH_EVERY) == HB_FRESH_EVERY - 1;
#else
    (void)looks;
    return false;
#endif
}

// A wait that has lasted a few hundred looks writes back the dirty lines of ITS OWN XCD's L2 (buffer_wbl2 sc1). What the launch
// stamps and the memory-side looks of round 4 showed about the dense stall (profiles/r04_dense_stall_diagnostics.txt): once in a few
// thousand sweeps the device pauses for ~1 ms (a launch starts 0.85 ms after its predecessor ended; `max_ms` of the in-situ stamps shows
// the same pauses in runs that do not stall), and afterwards ONE write-through store instruction of the chain workgroup — a sub-block's
// 64 changes of effect — is in nobody's view: every reader on every other XCD, memory-side atomics included, sees the pre-filled sentinel
// for 3 s, while the value appears in memory the moment the kernels end (their end-of-kernel release writes the L2 back). The line sits
// dirty in the WRITER's L2. The writer is by then waiting itself — for the sums that depend on that very store — so the remedy lives in
// the waits: whoever has published write-through data and then waits longer than any healthy hand-off takes flushes its L2. A healthy
// wait never gets here (hand-offs take microseconds); a stalled one is released within a fraction of a millisecond instead of 3 s.
#ifndef HB_UPD_FLAG_FIRST
#define HB_UPD_FLAG_FIRST 0
#endif
#ifndef HB_FLUSH_LOOKS
#define HB_FLUSH_LOOKS 0 /* off: measured, it does not release a stall (§9.0) — 11 sweeps in 16 000 still timed out with it */
#endif
__device__ unsigned hb_long_wait_flushes; // (diagnostics: how often a wait got that far; read by fetch_acc with HB_DEBUG_ABORT)
__device__ __forceinline__ void hb_long_wait(unsigned looks)
{
#if HB_FLUSH_LOOKS > 0
    if ((looks % HB_FLUSH_LOOKS) == HB_FLUSH_LOOKS - 1) { // (uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) atomicAdd(&hb_long_wait_flushes, 1u);
    }
#else
    (void)looks;
#endif
}

// one lane waits until *word >= want; bounded; returns false when the run is being aborted
template <int SLEEP = 8>
__device__ __forceinline__ bool wait_ge(unsigned *flags, int word, unsigned want)
{
    const unsigned long long t0 = wall_clock64();
    for (unsigned looks = 0;; looks++) {
        if ((hb_fresh_look(looks) ? ld_flag_fresh(flags + word) : ld_flag(flags + word)) >= want) return true;
        if (hb_fresh_look(looks) ? ld_flag_fresh(flags + HB_FLAG_ABORT) : ld_flag(flags + HB_FLAG_ABORT)) return false;
        hb_long_wait(looks);
        if (wall_clock64() - t0 > HB_TIMEOUT_TICKS) {
            st_flag(flags + HB_FLAG_ABORT, 1u);
            st_flag(flags + 8, want); // (diagnostics: who gave up, hb_ctx.hip fetch_acc)
            hb_abort_log(flags, HB_LOG_WAIT_GE, true, (unsigned)word, want, ld_flag(flags + word));
            return false;
        }
        __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum, result valid in every thread; red must hold blockDim.x/64 entries
template <typename T>
__device__ __forceinline__ T block_sum(T v, T *red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    T s = 0;
    for (int i = 0; i < nw; i++) s += red[i];
    return s;
}

// ---------------------------------------------------------------------------------------------
// marker statistics, reference src/Bayes.cpp:310-317 — integer-exact
// one workgroup per column
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stats(const int8_t *__restrict__ X, int64_t ld, int n, int m,
                                               double *__restrict__ xpx, double *__restrict__ vx,
                                               int *__restrict__ xinfo, double *__restrict__ s1out)
{
    __shared__ long long red[4];
    const int j = blockIdx.x;
    const int8_t *col = X + (int64_t)j * ld;
    long long s1 = 0, s2 = 0;
    int mn = 127, mx = -128;
    for (int64_t r0 = (int64_t)threadIdx.x * 16; r0 < ld; r0 += 256 * 16) {
        const int4 v = *reinterpret_cast<const int4 *>(col + r0);
        const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = (int)(int8_t)(w[q] >> (8 * b));
                if (r0 + q * 4 + b < n) {
                    s1 += x;
                    s2 += x * x;
                    mn = min(mn, x);
                    mx = max(mx, x);
                }
            }
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (j < m) {
        atomicMin(&xinfo[0], mn);
        atomicMax(&xinfo[1], mx);
    }
    if (threadIdx.x == 0) {
        if (s1out) s1out[j] = j < m ? (double)s1 : 0.0; // (row-sharded cross-check mode: the shards' integer sums are added up by the host)
        if (j < m) {
            xpx[j] = (double)s2;
            const long long num = (long long)n * s2 - s1 * s1; // n*S2 - S1^2, exact
            vx[j] = (num == 0 || n < 2) ? 0.0 : (double)num / ((double)n * (double)(n - 1));
        } else {
            xpx[j] = 0.0;
            vx[j] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// residual update: yadj -= sum_e x_e D_e for the markers that moved (shared by k_update and by the extra grid row of
// the fused mat-vec launch). thread = 4 consecutive rows; the move list is staged in LDS, 8 column loads in flight.
// ---------------------------------------------------------------------------------------------
struct upd_view {
    const int8_t *X;             // base of the genotype matrix
    int P, p0, p1;               // panels [p0, p1) whose moves are applied (p1 <= p0: nothing to do)
    const int32_t *ev_count, *ev_idx;
    const double *ev_delta;
    const double *r_in;          // residual before, and ...
    double *r, *u;               // ... after (distinct buffers under look-ahead); u updated in place
    float *r32;
    unsigned *flags;             // non-null: wait for chain_done >= p1 first (persistent pipeline)
    // fixed-point path (precise == 2): the new version is also written as HB_ND digit planes of rint(yadj * 2^E);
    // E comes from the bound on max |yadj| the chain publishes with these moves, so |q| <= 2^54 is guaranteed
    int8_t *rq;                  // digit planes of the output slot (null: other paths)
    const double *mbv;           // bound on max |yadj| after these moves
    int *vexp_out;               // exponent of the output slot
    const uint32_t *X2;          // non-null: the genotypes in the 2-bit resident layout (hb_dotq2.hpp), ld2w words per column
    int64_t ld2w;
    int dense;                   // every marker of a panel moves (BayesRR / A / L with k_chain_dense): one row per lane, 64 rows per wave (update_rows_dense)
    const double *dd;            // ... and the changes by marker, zero where nothing moved (k_chain_dense's dd[])
};

// four consecutive individuals (row0 a multiple of 4) of one column, one genotype per byte: from the int8 matrix, or expanded in
// registers from the 2-bit resident layout (individual 16 w + 4 k + b sits in bits [8 b + 2 k, 8 b + 2 k + 1] of word w)
__device__ __forceinline__ int hb_ld4(const int8_t *X, int64_t ld, const uint32_t *X2, int64_t ld2w, int64_t col, int64_t row0)
{
    if (X2) {
        const unsigned w = X2[col * ld2w + (row0 >> 4)];
        return (int)((w >> ((row0 & 12) >> 1)) & 0x03030303u);
    }
    return *reinterpret_cast<const int *>(X + col * ld + row0);
}

// exponent E with bound * 2^E < 2^54 (0 for an all-zero or non-finite bound)
__device__ __forceinline__ int hb_fix_exp(double bound)
{
    if (!(bound > 0.0) || !(bound < 1e300)) return 0;
    const int e = min(max(53 - ilogb(bound), -900), 900);
    return e;
}

// balanced base-256 digits of four fixed-point values, packed per plane (byte b = row b)
__device__ __forceinline__ void hb_store_digits(int8_t *rq, int64_t ld, int64_t row0, int E, double r0, double r1, double r2, double r3)
{
    long long q[4] = {__double2ll_rn(ldexp(r0, E)), __double2ll_rn(ldexp(r1, E)), __double2ll_rn(ldexp(r2, E)),
                      __double2ll_rn(ldexp(r3, E))};
#pragma unroll
    for (int k = 0; k < HB_ND; k++) {
        unsigned w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int d = (k == HB_ND - 1) ? (int)q[b] : (int)(int8_t)(q[b] & 0xff);
            q[b] = (q[b] - d) >> 8;
            w |= ((unsigned)d & 0xffu) << (8 * b);
        }
        *reinterpret_cast<unsigned *>(rq + (int64_t)k * ld + row0) = w;
    }
}

// rows [row0, row0 + 4) of the residual: yadj -= sum_e x_e D_e, u += the same, r32 = (float)yadj
__device__ __forceinline__ void update_rows(int64_t ld, const upd_view &q, int blk, int *s_ix,
                                            double *s_dl, int *s_ok, unsigned long long *ust = nullptr)
{
    // (ust: HB_DEBUG_ABORT diagnostics — block 64 of the launch leaves the lengths of its phases, four 16-bit counts of 100 MHz ticks:
    // poll of counts and bound | move lists | columns and sums | stores; tools/launch_roles.py prints their means)
    const unsigned long long tA = ust ? wall_clock64() : 0ull;
    unsigned long long tB = tA, tC = tA, tD = tA;
    const int64_t row0 = ((int64_t)blk * blockDim.x + threadIdx.x) * 4;
    const bool mine = row0 < ld;
    // the residual rows do not depend on the chain: fetch them before waiting for it
    double2 r01 = make_double2(0, 0), r23 = r01, u01 = r01, u23 = r01;
    if (mine) {
        r01 = *reinterpret_cast<const double2 *>(q.r_in + row0);
        r23 = *reinterpret_cast<const double2 *>(q.r_in + row0 + 2);
        u01 = *reinterpret_cast<const double2 *>(q.u + row0);
        u23 = *reinterpret_cast<const double2 *>(q.u + row0 + 2);
    }
    // Round 4: THE DATA IS THE FLAG, and the whole group takes three dependent memory round trips — (1) the panels' move counts and the
    // bound, polled directly: the sweep pre-fills both with a pattern no value has (count -1, bound ffff...), every word lands whole, so
    // a word is either that pattern (look again) or final; (2) the move lists of all panels of the group at once, their entries
    // pre-filled and validated the same way; (3) the genotype columns of up to 32 moves at a time — instead of chain_done first, then the
    // bound, then the counts, then per panel with moves its list and its columns (~10 trips of 2-3 us each beside the streaming tiles:
    // the update blocks of a BayesR launch lived 19 us against 7 for its tiles, profiles/r04_launch_roles_*).
    // The moves are applied in the same order (panel, then position in its list): the same sums bit for bit.
    int fixE = 0;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int total = 0;
    int nevs[8]; // a group has at most 8 panels
    {
        double mbv = 0.0;
        const bool poll = q.flags != nullptr;
#if HB_UPD_FLAG_FIRST
        // (A/B: one lane waits for chain_done first — ONE polled word for all update blocks — and the counts and the bound are then read
        // once, validated like below: a trip more, but the lines the chain stores its counts and bounds to are not polled)
        if (poll) {
            if (threadIdx.x == 0) *s_ok = wait_ge(q.flags, HB_FLAG_CHAIN_DONE, (unsigned)q.p1) ? 1 : 0;
            __syncthreads();
            if (!*s_ok) return;
        }
#endif
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            if (q.rq) mbv = ld_sc1(q.mbv); // (every thread the same word: one broadcast load per wave, in flight with the counts)
#pragma unroll
            for (int i = 0; i < 8; i++) nevs[i] = ld_sc1(q.ev_count + (size_t)min(q.p0 + i, q.p1 - 1) * HB_EVS);
            if (!poll) break; // (the per-panel kernels: a kernel boundary separates this from the chain)
            bool bad = q.rq && __double_as_longlong(mbv) == -1ll;
#pragma unroll
            for (int i = 0; i < 8; i++) bad |= nevs[i] < 0;
            if (!bad) break; // (wave-uniform: every lane read the same words)
            if (ld_flag(q.flags + HB_FLAG_ABORT) || wall_clock64() - t0 > HB_TIMEOUT_TICKS) {
                if (threadIdx.x == 0) {
                    st_flag(q.flags + HB_FLAG_ABORT, 1u);
                    st_flag(q.flags + 8, (unsigned)q.p1); // (diagnostics: who gave up, hb_ctx.hip fetch_acc)
                    if ((blk & 31) == 0) hb_abort_log(q.flags, HB_LOG_WAIT_GE, wall_clock64() - t0 > HB_TIMEOUT_TICKS, (unsigned)q.p0, (unsigned)q.p1, 0ull);
                }
                return;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (q.rq) { // (uniform) exponent of the new version, the same number in every workgroup
            fixE = hb_fix_exp(mbv);
            if (blk == 0 && threadIdx.x == 0) *q.vexp_out = fixE;
        }
    }
    if (ust) tB = tC = tD = wall_clock64();
    int off[9];
    off[0] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) off[i + 1] = off[i] + (q.p0 + i < q.p1 ? nevs[i] : 0);
    total = off[8];
    constexpr int CH = 448; // moves staged per pass (s_ix: 512 ints, s_dl: 512 doubles — the 64 behind the last move hold changes of zero: a batch reads past the list without a test per move)
    // (the group's columns from a scalar base + a 32-bit byte offset wherever the group's genotypes span less than 4 GB)
    const int8_t *Xg = q.X ? q.X + (int64_t)q.p0 * q.P * ld : nullptr;
    const uint32_t *X2g = q.X2 ? q.X2 + (int64_t)q.p0 * q.P * q.ld2w : nullptr;
    const bool wide_off = (uint64_t)(q.p1 - q.p0) * q.P * (uint64_t)ld < (1ull << 32);
    for (int base = 0; base < total; base += CH) {
        const int cnt = min(CH, total - base);
        __syncthreads();
        if (threadIdx.x < 64) s_dl[cnt + threadIdx.x] = 0.0;
        for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
            const int ge = base + e;
            int i = 0;
#pragma unroll
            for (int k = 1; k < 8; k++) i += (ge >= off[k]) ? 1 : 0; // panel of move ge (off[] is non-decreasing; panels past the group add nothing)
            int oi = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) oi = (i == k) ? off[k] : oi;
            const size_t src = (size_t)(q.p0 + i) * q.P + (size_t)(ge - oi);
            int ix = ld_sc1(q.ev_idx + src);
            double dl = ld_sc1(q.ev_delta + src);
            if (q.flags) { // (an entry whose count is already visible may itself still be on its way: pre-filled like the counts)
                const unsigned long long t1 = wall_clock64();
                while (ix < 0 || __double_as_longlong(dl) == -1ll) {
                    if (ld_flag(q.flags + HB_FLAG_ABORT) || wall_clock64() - t1 > HB_TIMEOUT_TICKS) { st_flag(q.flags + HB_FLAG_ABORT, 1u); ix = 0; dl = 0.0; break; }
                    __builtin_amdgcn_s_sleep(2);
                    ix = ld_sc1(q.ev_idx + src);
                    dl = ld_sc1(q.ev_delta + src);
                }
            }
            s_ix[e] = i * q.P + ix; // the move's COLUMN, counted from the group's first (a 32-bit byte offset from a scalar base then addresses it: one register per load in flight instead of two)
            s_dl[e] = dl;
        }
        __syncthreads();
        if (ust && base == 0) tC = tD = wall_clock64();
        if (!mine) continue;
        // columns in flight per thread: 32 where a panel has many moves (BayesR's ~50: two trips), 8 where a group has a handful (the
        // point-mass models in the stationary regime: padding a batch of 32 with repeats of the last column cost 100 conversions and
        // fp64 multiply-adds per row for nothing, beside tiles that keep the vector unit busy — 11 us of a block's 18, r04_launch_roles_*)
        auto batch = [&](auto UBC, auto WOC, int e) {
            constexpr int UB = decltype(UBC)::value;
            constexpr bool WO = decltype(WOC)::value; // (32-bit offsets from the group's scalar base)
            int w[UB];
            // (which layout is decided OUTSIDE the loops: a test per load made hipcc branch per load and wait for each 2-bit word before
            // the next was requested)
            if (!WO) {
#pragma unroll
                for (int k = 0; k < UB; k++) w[k] = hb_ld4(q.X, ld, q.X2, q.ld2w, (int64_t)q.p0 * q.P + s_ix[min(e + k, cnt - 1)], row0);
            } else if (q.X2) {
                const unsigned rw = ((unsigned)row0 >> 4) * 4u, sh = ((unsigned)row0 & 12u) >> 1, ldb = (unsigned)q.ld2w * 4u;
#pragma unroll
                for (int k = 0; k < UB; k++)
                    w[k] = (int)*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(X2g) + ((unsigned)s_ix[min(e + k, cnt - 1)] * ldb + rw));
#pragma unroll
                for (int k = 0; k < UB; k++) w[k] = (int)(((unsigned)w[k] >> sh) & 0x03030303u);
            } else {
#pragma unroll
                for (int k = 0; k < UB; k++)
                    w[k] = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(Xg) + ((unsigned)s_ix[min(e + k, cnt - 1)] * (unsigned)ld + (unsigned)row0));
            }
            if (UB > 32) __builtin_amdgcn_sched_barrier(0); // (all loads out before any arithmetic, and the arithmetic eight moves at a time: hipcc otherwise reads all 64 changes from LDS ahead — 231 VGPRs)
#pragma unroll
            for (int k = 0; k < UB; k++) {
                if (UB > 32 && (k & 7) == 0) __builtin_amdgcn_sched_barrier(0);
                const double d = s_dl[e + k]; // (zero past the list)
                a0 = fma((double)(int8_t)(w[k]), d, a0);
                a1 = fma((double)(int8_t)(w[k] >> 8), d, a1);
                a2 = fma((double)(int8_t)(w[k] >> 16), d, a2);
                a3 = fma((double)(int8_t)(w[k] >> 24), d, a3);
            }
        };
        int e = 0;
        if (wide_off) {
            for (; cnt - e > 32; e += 64) batch(std::integral_constant<int, 64>(), std::true_type(), e); // (BayesR's ~52 moves in ONE trip: the update rows of a launch are what the chain's next dots wait for)
            for (; cnt - e > 8; e += 32) batch(std::integral_constant<int, 32>(), std::true_type(), e);
            if (e < cnt) batch(std::integral_constant<int, 8>(), std::true_type(), e);
        } else {
            for (; cnt - e > 8; e += 32) batch(std::integral_constant<int, 32>(), std::false_type(), e);
            if (e < cnt) batch(std::integral_constant<int, 8>(), std::false_type(), e);
        }
    }
    if (ust) { asm volatile("" : "+v"(a0), "+v"(a1)); tD = wall_clock64(); }
    if (!mine || (total == 0 && q.r_in == q.r)) return;
    r01.x -= a0; r01.y -= a1; r23.x -= a2; r23.y -= a3;
    *reinterpret_cast<double2 *>(q.r + row0) = r01;
    *reinterpret_cast<double2 *>(q.r + row0 + 2) = r23;
    *reinterpret_cast<float4 *>(q.r32 + row0) = make_float4((float)r01.x, (float)r01.y, (float)r23.x, (float)r23.y);
    if (q.rq) hb_store_digits(q.rq, ld, row0, fixE, r01.x, r01.y, r23.x, r23.y);
    if (total) {
        u01.x += a0; u01.y += a1; u23.x += a2; u23.y += a3;
        *reinterpret_cast<double2 *>(q.u + row0) = u01;
        *reinterpret_cast<double2 *>(q.u + row0 + 2) = u23;
    }
    if (ust && blk == 64 && threadIdx.x == 0) {
        const unsigned long long tE = wall_clock64();
        auto c16 = [](unsigned long long d) { return d > 65535ull ? 65535ull : d; };
        *ust = c16(tB - tA) | (c16(tC - tB) << 16) | (c16(tD - tC) << 32) | (c16(tE - tD) << 48);
    }
}

// The same update where every marker of a panel moved (BayesRR / A / L with k_chain_dense: a second pass over the panel's genotypes).
// A mat-vec launch has ONE update wave per 256 rows with update_rows, and such a wave walks the panel in batches of a few loads
// per lane, one loaded memory round trip (~3 us beside the streaming tiles) per batch: measured 60-77 us per panel of 512 at
// n = 50k with 8 or 32 loads in flight, software-pipelined or not, and the same with one row per lane and 32 byte loads in
// flight (16 round trips). Here a wave owns 64 rows and brings its 64 x 512 slab of genotypes into LDS by LDS-DMA
// (global_load_lds_dwordx4: lane l = rows 16 (l & 3) .. + 15 of column 16 i + l / 4, so a piece of 1 KiB is 16 columns x 64 rows),
// in chunks of 128 columns through two 8-KB buffers: the first two chunks are requested BEFORE the wave waits for the chain (the
// genotypes do not depend on it), the others land under the arithmetic — one sign-extending LDS byte read, one convert and one
// fused multiply-add per column, lane = row. The changes come from k_chain_dense's dd[] (one double per marker, zero for a
// marker that did not move: a term x * 0 changes no sum), so no move list is read and every address is known at once.
// Blocks b and b + 8 — the same XCD under round-robin dispatch — take the two halves of the same 128-byte lines.
// Same sums in the same (marker) order as update_rows: the same residual bit for bit. Groups of at most 2 panels.
// smem: [0, 8192) the group's changes (<= 1024 doubles), [8192, 8208) flags, [HBU_SLAB, HBU_SLAB + 16384) two chunk buffers.
#define HBU_SENT(x) (__double_as_longlong(x) == -1ll)
#define HBU_SLAB 8448
#define HBU_LDS (HBU_SLAB + 16384)
__device__ __forceinline__ void update_rows_dense(int64_t ld, const upd_view &q, int blk, int nblk, char *smem)
{
    double *s_dl = reinterpret_cast<double *>(smem);
    const signed char *slab = reinterpret_cast<const signed char *>(smem + HBU_SLAB);
    const unsigned slab_lds = (unsigned)(uintptr_t)(smem + HBU_SLAB);
    const int lane = threadIdx.x;
    const int full = nblk & ~15;
    const int rc = blk < full ? (blk & ~15) + ((blk & 7) << 1) + ((blk >> 3) & 1) : blk;
    const int64_t row0 = (int64_t)rc * 64, row = row0 + lane;
    const int ncol = (q.p1 - q.p0) * q.P, nch = ncol >> 7; // (P is a multiple of 128: panel 512)
    const int8_t *xp = q.X + (int64_t)q.p0 * q.P * ld + row0 + (lane & 3) * 16 + (int64_t)(lane >> 2) * ld;
    auto issue = [&](int ch) { // columns 128 ch .. 128 ch + 127 of the group: 8 pieces
        const unsigned dst = slab_lds + (unsigned)(ch & 1) * 8192u;
        const int8_t *src = xp + (int64_t)ch * 128 * ld;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            unsigned keep; // (M0, the LDS destination base, is compiler-reserved: set and restored inside the statement)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src + (int64_t)i * 16 * ld), "s"(dst + (unsigned)i * 1024u)
                         : "memory");
        }
    };
    issue(0);
    if (nch > 1) issue(1);
    double r0 = q.r_in[row], u0 = q.u[row];
    // the group's changes and the bound the digits' exponent comes from, polled directly (both are sentinel-prefilled and written
    // once per sweep: every 8-byte value lands whole) — one round trip where waiting for chain_done first and loading them
    // afterwards is two
    double dv[16], mbv = 0.0;
    {
        const unsigned long long t0 = wall_clock64();
        int relook = 0;
        for (;;) {
#pragma unroll
            for (int i = 0; i < 16; i++) dv[i] = ld_sc1(q.dd + (size_t)q.p0 * q.P + min(64 * i + lane, ncol - 1));
            if (q.rq) mbv = ld_sc1(q.mbv); // (every lane the same word: one broadcast load)
            if (relook > 1) { // (a word that is still missing although the group's last one was seen: read it at the memory side, see ld_fresh)
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (HBU_SENT(dv[i])) dv[i] = ld_fresh(q.dd + (size_t)q.p0 * q.P + min(64 * i + lane, ncol - 1));
                if (q.rq && HBU_SENT(mbv)) mbv = ld_fresh(q.mbv);
            }
            relook++;
            bool bad = q.rq && HBU_SENT(mbv);
#pragma unroll
            for (int i = 0; i < 16; i++) bad |= HBU_SENT(dv[i]);
            if (!q.flags || !__any(bad)) break; // (no flags: the serial kernels, everything is final)
            // not there yet: wait on ONE word — the group's last change, or the bound, both written at its very end — and look at
            // everything again afterwards (784 waves polling 17 words each would be traffic the chain does not need)
            const double *last = q.rq ? q.mbv : q.dd + (size_t)q.p0 * q.P + (ncol - 1);
            bool dead = false;
            unsigned looks = 0;
            while (HBU_SENT(hb_fresh_look(looks) ? ld_fresh(last) : ld_sc1(last))) {
                if ((hb_fresh_look(looks) ? ld_flag_fresh(q.flags + HB_FLAG_ABORT) : ld_flag(q.flags + HB_FLAG_ABORT)) || wall_clock64() - t0 > HB_TIMEOUT_TICKS) { dead = true; break; }
#ifdef HB_UPD_SLEEP
                __builtin_amdgcn_s_sleep(HB_UPD_SLEEP);
                __builtin_amdgcn_s_sleep(HB_UPD_SLEEP);
#else
                hb_poll_pause(looks, 8);
#endif
                looks++;
            }
            // (the last word is there and an earlier one is not yet visible: look again, but never without the bound on the wait)
            if (!dead && (ld_flag(q.flags + HB_FLAG_ABORT) || wall_clock64() - t0 > HB_TIMEOUT_TICKS)) dead = true;
            if (dead) {
                if (lane == 0) {
                    const bool own = wall_clock64() - t0 > HB_TIMEOUT_TICKS;
                    st_flag(q.flags + HB_FLAG_ABORT, 1u);
                    st_flag(q.flags + 8, (unsigned)q.p1);
                    if ((blk & 31) == 0 || own) hb_abort_log(q.flags, HB_LOG_UPD_DENSE, own, (unsigned)q.p0, (unsigned)blk, (unsigned long long)__double_as_longlong(ld_sc1(last)));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                return;
            }
        }
    }
    int fixE = 0;
    if (q.rq) fixE = hb_fix_exp(mbv);
#pragma unroll
    for (int i = 0; i < 16; i++)
        if (64 * i + lane < ncol) s_dl[64 * i + lane] = dv[i];
    if (q.rq && blk == 0 && lane == 0) *q.vexp_out = fixE;
    __syncthreads();
    double a = 0.0;
    for (int ch = 0; ch < nch; ch++) {
        // (in flight behind chunk ch: chunk ch + 1 — 8 pieces — and nothing else: the loads above have been consumed)
        if (ch + 1 < nch) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const signed char *sl = slab + (ch & 1) * 8192 + lane;
        const double *dl = s_dl + (ch << 7);
#pragma unroll 4
        for (int e = 0; e < 128; e += 8) {
            int w[8];
#pragma unroll
            for (int k = 0; k < 8; k++) w[k] = sl[(e + k) * 64];
#pragma unroll
            for (int k = 0; k < 8; k++) a = fma((double)w[k], dl[e + k], a);
        }
        if (ch + 2 < nch) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (every read of this buffer has returned)
            issue(ch + 2);
        }
    }
    r0 -= a;
    q.r[row] = r0;
    q.r32[row] = (float)r0;
    if (q.rq) { // balanced base-256 digits of rint(yadj 2^E), one byte per plane (hb_store_digits for one row)
        long long qv = __double2ll_rn(ldexp(r0, fixE));
#pragma unroll
        for (int k = 0; k < HB_ND; k++) {
            const int d = (k == HB_ND - 1) ? (int)qv : (int)(int8_t)(qv & 0xff);
            qv = (qv - d) >> 8;
            q.rq[(int64_t)k * ld + row] = (int8_t)d;
        }
    }
    q.u[row] = u0 + a;
}

// ---------------------------------------------------------------------------------------------
// k_dot: partial[split][col] = sum over the split's rows of x[row][col] * yadj[row]
// tile = 8 columns x (256 threads x 16 rows); grid = (ncols/8, nsplit)
// ---------------------------------------------------------------------------------------------
// Pipeline hand-off of a mat-vec launch (red_ncols = 0 outside the pipeline). The split partials of a launch are
// added up by the FIRST grid row of the NEXT launch (the kernel boundary makes them visible: no per-tile atomics, no
// write-through stores and no reduction tail in the streaming workgroups). The sums are written through to dsum[],
// which the sweep pre-filled with a NaN bit pattern: the chain workgroup needs no flag to know a value has arrived,
// and reads 8 bytes per marker instead of 8 per split.
struct dot_sync {
    const double *red_partial; // [split][pstride] partials of the previous launch's columns
    double *red_dsum;          // their sums
    int red_ncols;             // 0: nothing to reduce in this launch
    int nsplit;
};

__device__ __forceinline__ void reduce_partials(const dot_sync &sy, int pstride, int blk, int tid)
{
    const int col = blk * 256 + tid;
    if (col >= sy.red_ncols) return;
    double tot = 0.0;
    for (int q = 0; q < sy.nsplit; q++) tot += sy.red_partial[(int64_t)q * pstride + col]; // split order: a fixed sum
    st_sc1(&sy.red_dsum[col], tot);
}

__global__ __launch_bounds__(256) void k_reduce_partials(dot_sync sy, int pstride) { reduce_partials(sy, pstride, blockIdx.x, threadIdx.x); }

typedef unsigned int hb_u4 __attribute__((ext_vector_type(4)));

template <bool SIGNED>
__device__ __forceinline__ float b2f(unsigned w, int b)
{
    if (SIGNED) return (float)(int)(int8_t)(w >> (8 * b));
    return (float)((w >> (8 * b)) & 0xffu);
}

template <bool PRECISE, bool SIGNED>
__global__ __launch_bounds__(256) void k_dot(const int8_t *__restrict__ X, int64_t ld,
                                             const float *__restrict__ r32,
                                             const double *__restrict__ r64, int nchunks,
                                             int chunks_per_split, double *__restrict__ partial,
                                             int pstride, dot_sync sy, upd_view uq)
{
    using acc_t = typename std::conditional<PRECISE, double, float>::type;
    __shared__ acc_t red[4][8];
    const int ct = blockIdx.x, tid = threadIdx.x;
    int sp = blockIdx.y;
    if (uq.p1 > uq.p0) {
        // Pipeline launch: the FIRST grid row carries the residual update of an earlier group (its result is the version
        // the NEXT launch reads), so there is no third stream and no cross-stream event. First, because workgroups are
        // dispatched in grid order: these few start with the launch, wait for the chain workgroup while the tiles stream,
        // and are done long before the launch ends (as the last row they only got a compute unit when the tiles were nearly
        // through, and every launch ended with their wait, event fetch and column loads: +4 us on 23).
        if (sp == 0) {
            __shared__ int s_ix[512];
            __shared__ double s_dl[512];
            __shared__ int s_ok[2];
            for (int blk = ct; (int64_t)blk * 1024 < ld; blk += gridDim.x) update_rows(ld, uq, blk, s_ix, s_dl, s_ok);
            return;
        }
        sp -= 1;
    }
    if (sy.red_ncols > 0) { // next row: add up the previous launch's partials
        if (sp == 0) {
            reduce_partials(sy, pstride, ct, tid);
            return;
        }
        sp -= 1;
    }
    const int8_t *xc = X + (int64_t)ct * 8 * ld;
    acc_t acc[8];
#pragma unroll
    for (int c = 0; c < 8; c++) acc[c] = 0;
    const int ch1 = min(nchunks, (sp + 1) * chunks_per_split);
    for (int ch = sp * chunks_per_split; ch < ch1; ++ch) {
        const int64_t row0 = ((int64_t)ch * 256 + tid) * 16;
        if (row0 < ld) {
            uint4 xv[8];
#pragma unroll
            for (int c = 0; c < 8; c++) { // streamed once: non-temporal, so that the residual stays in L2
                const hb_u4 w = __builtin_nontemporal_load(reinterpret_cast<const hb_u4 *>(xc + (int64_t)c * ld + row0));
                xv[c] = make_uint4(w.x, w.y, w.z, w.w);
            }
            acc_t rv[16];
            if (PRECISE) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double2 t = *reinterpret_cast<const double2 *>(r64 + row0 + 2 * q);
                    rv[2 * q] = (acc_t)t.x;
                    rv[2 * q + 1] = (acc_t)t.y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 t = *reinterpret_cast<const float4 *>(r32 + row0 + 4 * q);
                    rv[4 * q] = (acc_t)t.x;
                    rv[4 * q + 1] = (acc_t)t.y;
                    rv[4 * q + 2] = (acc_t)t.z;
                    rv[4 * q + 3] = (acc_t)t.w;
                }
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const unsigned w[4] = {xv[c].x, xv[c].y, xv[c].z, xv[c].w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        acc[c] = fma((acc_t)b2f<SIGNED>(w[q], b), rv[q * 4 + b], acc[c]);
                }
            }
        }
    }
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const acc_t s = wave_sum(acc[c]);
        if (lane == 0) red[wv][c] = s;
    }
    __syncthreads();
    if (tid < 8) {
        const acc_t s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        partial[(int64_t)sp * pstride + ct * 8 + tid] = (double)s;
    }
}

// ---------------------------------------------------------------------------------------------
// k_dotq: the exact fixed-point mat-vec (precise == 2).  d_j = x_j . yadj is computed as
//     sum_k 256^k (x_j . D_k) * 2^-E,   D_k = digit plane k of q = rint(yadj * 2^E)  (balanced base-256 digits, int8)
// with every x_j . D_k an exact int8 x int8 -> int32 dot product (v_dot4_i32_i8, 4 multiply-adds per lane and
// instruction: 7 instructions per 4 genotypes against 8 for the fp32 path). Integer sums are order-independent, so the
// row splits combine through 64-bit atomics and the result does not depend on the launch geometry at all; its error is
// the quantisation of yadj alone (<= 2^-55 max|yadj| per element: below the rounding error of an fp64 ddot).
// One wave = 64 columns x NS stages of 128 rows, lane = column: the genotype tile AND the stage's digit planes arrive by
// LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction, double-buffered, counted vmcnt — nothing else is in the
// vector-memory queue); a lane reads its own column with ds_read_b128 and the digits with wave-uniform (broadcast)
// ds_read_b128. No cross-lane reduction anywhere. The slot stride of 1040 bytes rotates the LDS banks between the
// eight DMA pieces of a stage.
// Block roles by index: [0, nfin) finalize the previous launch's columns into dsum[], [nfin, nfin + nupd) residual update of an
// earlier group (its digits included), then the tiles.
// ---------------------------------------------------------------------------------------------
typedef int hb_v4i __attribute__((ext_vector_type(4)));
#define HBQ_RS 128                       /* rows per stage */
#define HBQ_SLOT 1040
#define HBQ_NX 8                         /* DMA pieces per stage for the genotype tile (8 columns x 128 rows each) */
#define HBQ_XB (HBQ_NX * HBQ_SLOT)
#define HBQ_BUF (HBQ_XB + 1024)          /* + one piece for the 7 digit planes */
#define HBQ_PER (HBQ_NX + 1)
#define HBQ_LDS (2 * HBQ_BUF)

struct dq_view {
    const int8_t *X;       // first column of this launch
    int64_t ld;
    const int8_t *rq;      // digit planes of the residual slot read
    const int *vexp_in;    // their exponent ...
    int *gexp_out;         // ... recorded for this launch's finalize
    long long *accq;       // [HB_ND][accstride], at this launch's first column
    int64_t accstride;
    int nstages, NS, ncg;
    int nupd, nfin;
    const long long *fin_acc; // finalize: digit-plane sums of the earlier launch's columns
    double *fin_out;
    const int *fin_exp;
    int fin_ncols;
    const uint8_t *X2;     // 2-bit resident layout (k_dotq2): first column of this launch, ld2 bytes per column
    int64_t ld2;
    unsigned long long *stamp; // optional (hb_ctx_set_profiling bit 3): [block][2] = wall_clock64() at the block's start and end
    unsigned long long *ldiag; // optional (HB_DEBUG_ABORT): [0] start of the launch's first block, [1] latest block end, [2] blocks finished
};

// (diagnostics of a pipeline time-out: when did each mat-vec launch start and end — fetch_acc prints the launches around the stall)
__device__ __forceinline__ void hb_ldiag_note(unsigned long long *ld, unsigned long long t0)
{
    if (threadIdx.x != 0 || (blockIdx.x & 31) != 0) return; // (every 32nd block: ~50 atomics per launch on two words perturb nothing)
    if (blockIdx.x == 0) __hip_atomic_store(&ld[0], t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(&ld[1], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&ld[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool NT>
__device__ __forceinline__ void hbq_dma16(unsigned voff, const int8_t *sbase, unsigned lds_dst)
{
    unsigned keep; // M0 (the LDS destination base) is compiler-reserved: set and restored inside the statement
    if (NT) // genotypes are streamed once: non-temporal, so that the digit planes and the chain's working set stay in L2
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(sbase), "s"(lds_dst)
                     : "memory");
}

__device__ __forceinline__ void hbq_finalize(const long long *acc, int64_t stride, int col, int E, double *out)
{
    double a = 0.0;
#pragma unroll
    for (int k = HB_ND - 1; k >= 0; k--) a = fma(a, 256.0, (double)acc[(int64_t)k * stride + col]);
    st_sc1(out + col, ldexp(a, -E));
}

__global__ __launch_bounds__(64) void k_dotq_fin(const long long *__restrict__ acc, int64_t stride, int ncols,
                                                 const int *__restrict__ pexp, double *__restrict__ out)
{
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col < ncols) hbq_finalize(acc, stride, col, *pexp, out);
}

__device__ __forceinline__ void dotq_block(const dq_view &v, const upd_view &uq, char *smem)
{
    const int lane = threadIdx.x;
    int b = blockIdx.x;
    // (the finalize blocks come FIRST: the chain workgroup is waiting for their sums, and the update blocks behind them wait for
    // the chain — with one update wave per 64 rows they can fill every slot of the chip, and a finalize block queued behind them
    // would never start)
    if (b < v.nfin) {
        const int col = b * 64 + lane;
        if (col < v.fin_ncols) hbq_finalize(v.fin_acc, v.accstride, col, *v.fin_exp, v.fin_out);
        return;
    }
    b -= v.nfin;
    if (b < v.nupd) { // residual update of an earlier group: 256 rows per block (64 where every marker moves), lists staged in the (unused) tile buffers
        if (uq.dense) update_rows_dense(v.ld, uq, b, v.nupd, smem); // (launched with HBU_LDS bytes of dynamic LDS)
        else update_rows(v.ld, uq, b, reinterpret_cast<int *>(smem), reinterpret_cast<double *>(smem + 2048),
                         reinterpret_cast<int *>(smem + 2048 + 4096), v.ldiag ? v.ldiag + 3 : nullptr);
        return;
    }
    b -= v.nupd;
    const int cg = b % v.ncg, sp = b / v.ncg;
    if (b == 0 && lane == 0) *v.gexp_out = *v.vexp_in;
    const int st0 = sp * v.NS, st1 = min(v.nstages, st0 + v.NS);
    if (st0 >= st1) return;
    const int64_t ld = v.ld;
    const int8_t *xg = v.X + (int64_t)cg * 64 * ld;
    const unsigned voff = (unsigned)((lane >> 3) * ld + (lane & 7) * 16);                    // piece i: columns 8i .. 8i+7
    const unsigned doff = (unsigned)(min(lane >> 3, HB_ND - 1) * ld + (lane & 7) * 16);    // digit piece: plane lane/8
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    int acc[HB_ND];
#pragma unroll
    for (int k = 0; k < HB_ND; k++) acc[k] = 0;
    auto issue = [&](int st, int buf) {
        const int8_t *base = xg + (int64_t)st * HBQ_RS;
        const unsigned dst = lds0 + (unsigned)buf * HBQ_BUF;
#pragma unroll
        for (int i = 0; i < HBQ_NX; i++) hbq_dma16<true>(voff, base + (int64_t)(8 * i) * ld, dst + i * HBQ_SLOT);
        hbq_dma16<false>(doff, v.rq + (int64_t)st * HBQ_RS, dst + HBQ_XB);
    };
    issue(st0, 0);
    int buf = 0;
    for (int st = st0; st < st1; ++st) {
        if (st + 1 < st1) {
            issue(st + 1, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(HBQ_PER) : "memory"); // everything but the stage just requested has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char *bp = smem + buf * HBQ_BUF;
        const hb_v4i *px = reinterpret_cast<const hb_v4i *>(bp + (lane >> 3) * HBQ_SLOT + (lane & 7) * HBQ_RS);
        const char *pd = bp + HBQ_XB;
#pragma unroll
        for (int s = 0; s < HBQ_RS / 16; s++) {
            const hb_v4i x = px[s];
#pragma unroll
            for (int k = 0; k < HB_ND; k++) {
                const hb_v4i d = *reinterpret_cast<const hb_v4i *>(pd + k * HBQ_RS + s * 16);
                acc[k] = __builtin_amdgcn_sdot4(x.x, d.x, acc[k], false);
                acc[k] = __builtin_amdgcn_sdot4(x.y, d.y, acc[k], false);
                acc[k] = __builtin_amdgcn_sdot4(x.z, d.z, acc[k], false);
                acc[k] = __builtin_amdgcn_sdot4(x.w, d.w, acc[k], false);
            }
        }
        buf ^= 1;
    }
#pragma unroll
    for (int k = 0; k < HB_ND; k++)
        __hip_atomic_fetch_add(v.accq + (int64_t)k * v.accstride + cg * 64 + lane, (long long)acc[k], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// Every block's role is decided by its index (see above). With v.stamp set — the in-situ measurement of bench.py: the launches
// of a real sweep, chain and update rows beside them — each block also records the constant 100 MHz clock at its start and end;
// the launch's duration is then max(end) - min(start) over its blocks, what a kernel trace reports for it.
__global__ __launch_bounds__(64) void k_dotq(dq_view v, upd_view uq)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t0 = 0;
    if (v.stamp || v.ldiag) t0 = wall_clock64();
    dotq_block(v, uq, smem);
    if (v.stamp && threadIdx.x == 0) {
        v.stamp[2 * (size_t)blockIdx.x] = t0;
        v.stamp[2 * (size_t)blockIdx.x + 1] = wall_clock64();
    }
    if (v.ldiag) hb_ldiag_note(v.ldiag, t0);
}

#include "hb_dotq2.hpp"

// Sweep start of the fixed-point path: max |yadj| -> mb[0] and the exponent of slot 0, then slot 0's digit planes.
// One workgroup (n is a few hundred KB).
__global__ __launch_bounds__(256) void k_sweep_init(double *__restrict__ acc, unsigned *__restrict__ flags, int32_t *__restrict__ ev_count,
                                                    int np, unsigned long long *__restrict__ dsum, int m_pad, int p_lo,
                                                    unsigned long long *__restrict__ fcorr, unsigned long long *__restrict__ dd,
                                                    unsigned long long *__restrict__ mbs, int mb_lo, int mb_hi,
                                                    unsigned long long *__restrict__ fc2, int32_t *__restrict__ ev_idx,
                                                    unsigned long long *__restrict__ ev_delta, int P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (acc && i < HB_ACC_N) acc[i] = 0.0; // (null: a later range of the same sweep keeps the sums)
    // (a later range of the same sweep keeps an abort raised by an earlier one, and with it who gave up waiting for what — words 8..14
    // and the abort log's record count — which is what HB_DEBUG_ABORT prints)
    if (i < HB_NFLAGS && (acc || (i != HB_FLAG_ABORT && !(i >= 8 && i <= 14) && i != HB_FLAG_LOGN))) flags[i] = 0u;
    // move counts and list entries of the panels of this range on: "not written yet" (the update rows poll them directly; an earlier
    // range's move lists stay readable)
    for (int k = p_lo + i; k < np; k += stride) ev_count[(size_t)k * HB_EVS] = -1;
    for (size_t k = (size_t)p_lo * P + i; k < (size_t)np * P; k += stride) {
        ev_idx[k] = -1;
        ev_delta[k] = ~0ull;
    }
    for (int k = i; k < m_pad; k += stride) dsum[k] = ~0ull;
    if (fcorr)
        for (int k = i; k < m_pad; k += stride) fcorr[k] = ~0ull;
    if (dd)
        for (int k = i; k < m_pad; k += stride) dd[k] = ~0ull;
    if (fc2)
        for (int k = i; k < m_pad; k += stride) fc2[k] = ~0ull;
    if (mbs) // (the dense update rows poll the group's bound on max |yadj| together with its changes: "not written yet")
        for (int k = mb_lo + i; k < mb_hi; k += stride) mbs[(size_t)k * HB_MBS] = ~0ull;
}

__global__ __launch_bounds__(1024) void k_quant0(const double *__restrict__ r, int64_t ld, int8_t *__restrict__ rq,
                                                 double *__restrict__ mb, int *__restrict__ vexp, const double *__restrict__ force_max)
{
    __shared__ double red[16];
    __shared__ double s_max;
    double mx = force_max ? *force_max : 0.0; // (row-sharded mode: max |yadj| over ALL shards, so that every shard's digits share one exponent)
    for (int64_t i = threadIdx.x; i < ld && !force_max; i += blockDim.x) mx = fmax(mx, fabs(r[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m2 = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) m2 = fmax(m2, red[i]);
        s_max = m2;
        mb[0] = m2;
        vexp[0] = hb_fix_exp(m2);
    }
    __syncthreads();
    const int E = hb_fix_exp(s_max);
    for (int64_t row0 = (int64_t)threadIdx.x * 4; row0 < ld; row0 += (int64_t)blockDim.x * 4)
        hb_store_digits(rq, ld, row0, E, r[row0], r[row0 + 1], r[row0 + 2], r[row0 + 3]);
}

__global__ void k_sum_partials(const double *__restrict__ partial, int pstride, int nsplit, int ncols,
                               double *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    double s = 0;
    for (int sp = 0; sp < nsplit; sp++) s += partial[(int64_t)sp * pstride + j];
    out[j] = s;
}

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
        double qn = q - hv / d;
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

// ---------------------------------------------------------------------------------------------
// k_chain: one workgroup of P threads (thread = marker of the panel, wave = 64-marker sub-block).
// ---------------------------------------------------------------------------------------------
struct chain_view {
    int m_pad, P, nsplit, L, Lb; // L: version lag of the serial pipeline; Lb: Gram band blocks per panel - 1
    const double *xpx, *vx;
    double *g;
    uint8_t *tracker;
    uint32_t *nzrate;
    double *alpha_sum, *alpha_sq;
    const double *thr, *invv, *sdz;
    const int32_t *gram;
    const double *partial; // [split][m_pad] (serial pipeline)
    const double *dsum;    // [m_pad] reduced by the mat-vec itself (persistent pipeline)
    int32_t *ev_count, *ev_idx;
    double *ev_delta;
    double *acc;
    const uint32_t *wind;
    uint8_t *wflag;
    long long *dbg; // optional: 32 cycle stamps per panel (tools/chain_timeline.py)
    // fixed-point path: running bound on max |yadj| (mb[0] at sweep start, mb[1 + h] after group / panel h) — each move D of a
    // marker raises it by at most xabs * |D|; the update derives the digits' exponent from it (null: other paths)
    double *mb;
    double xabs;
};

// Cycle stamps of the chain kernels (tools/chain_timeline.py): compiled in only with -DHB_STAMPS=1 (tools/build_variant.sh) —
// thirteen "is profiling on?" branches per panel are a tenth of a quiet panel's instructions.
#ifndef HB_STAMPS
#define HB_STAMPS 0
#endif
#if HB_STAMPS
#define HB_STAMP(i) do { if (v.dbg && t == 0) v.dbg[(size_t)p * 32 + (i)] = clock64(); } while (0)
#define HB_STAMP_VAL(i, x) do { if (v.dbg && t == 0) v.dbg[(size_t)p * 32 + (i)] = (x); } while (0)
#else
#define HB_STAMP(i) do { } while (0)
#define HB_STAMP_VAL(i, x) do { } while (0)
#endif

__device__ __forceinline__ double readlane_f64(double v, int k)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

// LDS plan (dynamic, one object): [row cache: nslot x P int32][ev_del: P f64][ev_ix: P i32][slot_of: P i32]
// [red: 16 f64][cnts: 16 i32][wcnt: 16 i32].
// The row cache holds the full Gram rows G[k][0..P) of the markers that are certain to move this sweep
// (g_old != 0): both the in-wave corrections and the cross-wave ones are then LDS reads.  A marker that
// enters the model from zero (a "surprise") falls back to reading its row from global memory.
template <int K1>
__global__ __launch_bounds__(512) void k_chain(const hb_sweep_in *__restrict__ pin, chain_view v, int p, int nslot)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int P = v.P, S = P >> 6;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    int32_t *rowc = reinterpret_cast<int32_t *>(smem);
    char *base = smem + (size_t)nslot * P * 4;
    double *ev_del = reinterpret_cast<double *>(base);
    int *ev_ix = reinterpret_cast<int *>(base + (size_t)P * 8);
    int *slot_of = reinterpret_cast<int *>(base + (size_t)P * 12);
    double *red = reinterpret_cast<double *>(base + (size_t)P * 16);
    int *cnts = reinterpret_cast<int *>(base + (size_t)P * 16 + 128);
    int *wcnt = cnts + 16;

    const int j = p * P + t;
    const int32_t *gp = v.gram + (size_t)p * (v.Lb + 1) * P * P; // l = 0: this panel's own Gram block
    HB_STAMP(0);

    // ---- issue every per-marker load up front (one memory latency for all of them) ----
    const int model = pin->model_index;
    const double vxj = v.vx[j];
    const double gold = v.g[j];
    const double xx = v.xpx[j];
    double thr[K1], invv[K1], sdz[K1];
#pragma unroll
    for (int c = 0; c < K1; c++) {
        thr[c] = v.thr[(size_t)c * v.m_pad + j];
        invv[c] = v.invv[(size_t)c * v.m_pad + j];
        sdz[c] = v.sdz[(size_t)c * v.m_pad + j];
    }
    double ps[16];
    {
        const int last = v.nsplit - 1;
#pragma unroll
        for (int sp = 0; sp < 16; sp++) ps[sp] = v.partial[(size_t)min(sp, last) * v.m_pad + j]; // clamped: no branches
    }
    const bool active = vxj != 0.0;
    const bool hot = active && gold != 0.0;

    // ---- slots for the hot markers, in marker order ----
    const unsigned long long hmask = __ballot(hot);
    if (lane == 0) wcnt[wave] = __popcll(hmask);
    if (t < 16) cnts[t] = 0;
    __syncthreads();
    int sbase = 0, nhot = 0;
    for (int w = 0; w < S; w++) {
        const int c = wcnt[w];
        sbase += (w < wave) ? c : 0;
        nhot += c;
    }
    const int myslot_raw = sbase + __popcll(hmask & ((1ull << lane) - 1ull));
    const int myslot = (hot && myslot_raw < nslot) ? myslot_raw : -1; // lane-resident: slot of marker t
    slot_of[t] = myslot;
    if (myslot >= 0) ev_ix[myslot] = t; // borrowed as the slot -> marker list until the chain starts
    __syncthreads();
    // ---- stream the hot rows into LDS. One item = 256 consecutive ints of the row cache; four items per
    // wave in flight. Indices are clamped instead of predicated so that the loads stay branch-free. ----
    {
        const int ncached = min(nhot, nslot);
        if (ncached > 0) {
            const int lgP = 31 - __clz(P);
            const int total = ncached << lgP;          // ints in the cache image
            const int items = (total + 255) >> 8;
            for (int it0 = wave; it0 < items; it0 += 4 * S) {
                int4 val0, val1, val2, val3;
                int lin[4];
#pragma unroll
                for (int q = 0; q < 4; q++) lin[q] = min(((it0 + q * S) << 8) + lane * 4, total - 4);
                val0 = *reinterpret_cast<const int4 *>(gp + ((size_t)ev_ix[lin[0] >> lgP] << lgP) + (lin[0] & (P - 1)));
                val1 = *reinterpret_cast<const int4 *>(gp + ((size_t)ev_ix[lin[1] >> lgP] << lgP) + (lin[1] & (P - 1)));
                val2 = *reinterpret_cast<const int4 *>(gp + ((size_t)ev_ix[lin[2] >> lgP] << lgP) + (lin[2] & (P - 1)));
                val3 = *reinterpret_cast<const int4 *>(gp + ((size_t)ev_ix[lin[3] >> lgP] << lgP) + (lin[3] & (P - 1)));
                *reinterpret_cast<int4 *>(rowc + lin[0]) = val0; // clamped duplicates rewrite identical data
                *reinterpret_cast<int4 *>(rowc + lin[1]) = val1;
                *reinterpret_cast<int4 *>(rowc + lin[2]) = val2;
                *reinterpret_cast<int4 *>(rowc + lin[3]) = val3;
            }
        }
    }
    double rhs = 0.0;
#pragma unroll
    for (int sp = 0; sp < 16; sp++) rhs += (sp < v.nsplit) ? ps[sp] : 0.0;
    for (int sp = 16; sp < v.nsplit; sp++) rhs += v.partial[(size_t)sp * v.m_pad + j];
    // :594/:616/:725 add xx*oldgi always, :639/:682/:757 only when oldgi != 0 — identical values
    if (gold != 0.0) rhs = fma(xx, gold, rhs);
    // Look-ahead: this panel's mat-vec ran against the residual without the moves of the previous L panels.
    // Fold them in with the band Gram blocks  G_l[k][t] = x_{(p-l)P+k} . x_{pP+t}:  rhs_t -= G_l[k][t] D_k.
    for (int l = 1; l <= v.L; l++) {
        const int bp = p - l;
        if (bp < 0) break;
        const int nevp = v.ev_count[(size_t)bp * HB_EVS];
        const int32_t *gx = gp + (size_t)l * P * P;
        const int32_t *eix = v.ev_idx + (size_t)bp * P;
        const double *edl = v.ev_delta + (size_t)bp * P;
        for (int e0 = 0; e0 < nevp; e0 += 8) {
            int gv[8];
            double dl[8];
#pragma unroll
            for (int q8 = 0; q8 < 8; q8++) {
                const int e = min(e0 + q8, nevp - 1);
                gv[q8] = gx[(size_t)eix[e] * P + t];
                dl[q8] = (e0 + q8 < nevp) ? edl[e] : 0.0;
            }
#pragma unroll
            for (int q8 = 0; q8 < 8; q8++) rhs = fma(-(double)gv[q8], dl[q8], rhs);
        }
    }
    int cls_f = 0;
    double g_f = 0.0;
    __syncthreads();
    HB_STAMP(1);

    int ev_prev = 0;
    int *ev_sl = slot_of; // after the hot rows are cached, slot_of is only needed through ev_sl/myslot
    (void)ev_sl;
    for (int s = 0; s < S; s++) {
        if (wave == s) {
            int cnt = cnts[0];
            int lo = 0;
            unsigned long long hleft = hmask;                   // hot lanes not yet passed
            const unsigned long long amask = __ballot(active);  // polymorphic lanes
            for (;;) {
                // the next certain event is the next hot lane: fetch its Gram entry while deciding
                const int knext = hleft ? (__ffsll((long long)hleft) - 1) : 0;
                const int snext = __builtin_amdgcn_readlane(myslot, knext);
                int gnext = 0;
                if (hleft && snext >= 0) gnext = rowc[(size_t)snext * P + t];
                const double q = rhs * rhs;
                // a marker at zero moves only if it enters the model (q >= thr[0]); a hot one always moves
                const unsigned long long live = ~0ull << lo;
                const unsigned long long mask = ((__ballot(q >= thr[0]) & amask) | hleft) & live;
                if (mask == 0ull) break;
                const int k = __ffsll((long long)mask) - 1;
                int cls = 0;
                double iv = 0.0, sz = 0.0;
#pragma unroll
                for (int c = 0; c < K1; c++) {
                    const bool ge = q >= thr[c];
                    cls += ge ? 1 : 0;
                    iv = ge ? invv[c] : iv;
                    sz = ge ? sdz[c] : sz;
                }
                double gn = (cls > 0) ? fma(rhs, iv, sz) : 0.0;
                if (model == 5 && fabs(gn) < 1e-6) gn = 1e-6; // :728
                const double delta = gn - gold;
                if (lane == k) { cls_f = cls; g_f = gn; }
                const double dk = readlane_f64(delta, k);
                const int tk = 64 * s + k;
                if (dk != 0.0) { // (a hot marker redrawing exactly its old value would be a no-op)
                    int gv;
                    int slot = snext;
                    if (!(hleft && k == knext)) slot = __builtin_amdgcn_readlane(myslot, k);
                    if (hleft && k == knext && snext >= 0) {
                        gv = gnext;
                    } else if (slot >= 0) {
                        gv = rowc[(size_t)slot * P + t];
                    } else { // a marker entering the model from zero: its Gram row is still in global memory
                        gv = gp[(size_t)tk * P + t];
                    }
                    if (lane > k) rhs = fma(-(double)gv, dk, rhs);
                    if (lane == k) { ev_ix[cnt] = (slot << 16) | tk; ev_del[cnt] = dk; }
                    cnt++;
                }
                lo = k + 1;
                if (lo >= 64) break;
                hleft &= ~((2ull << k) - 1ull);
            }
            if (lane == 0) cnts[0] = cnt;
        }
        __syncthreads();
        const int ev_now = cnts[0];
        if (wave > s) { // later sub-blocks take the new events; event records first, Gram entries second
            for (int e0 = ev_prev; e0 < ev_now; e0 += 8) {
                int rec[8], gv[8];
                double dl[8];
#pragma unroll
                for (int q8 = 0; q8 < 8; q8++) {
                    const int e = min(e0 + q8, ev_now - 1);
                    rec[q8] = ev_ix[e];
                    dl[q8] = (e0 + q8 < ev_now) ? ev_del[e] : 0.0;
                }
#pragma unroll
                for (int q8 = 0; q8 < 8; q8++) {
                    const int slot = __builtin_amdgcn_readfirstlane(rec[q8] >> 16);
                    const int k = __builtin_amdgcn_readfirstlane(rec[q8] & 0xffff);
                    if (slot >= 0) gv[q8] = rowc[(size_t)slot * P + t];
                    else gv[q8] = gp[(size_t)k * P + t];
                }
#pragma unroll
                for (int q8 = 0; q8 < 8; q8++) rhs = fma(-(double)gv[q8], dl[q8], rhs);
            }
        }
        ev_prev = ev_now;
        if (s < 24) HB_STAMP(2 + s);
    }
    HB_STAMP(26);

    // ---- write back ----
    if (!active) { cls_f = 0; g_f = 0.0; }
    v.g[j] = g_f;
    v.tracker[j] = (uint8_t)cls_f;
    if (pin->count_pip && cls_f != 0) {
        v.nzrate[j] += 1u;
        if (v.wind) v.wflag[v.wind[j] - 1u] = 1;
    }
    if (pin->store) {
        v.alpha_sum[j] += g_f;
        v.alpha_sq[j] += g_f * g_f;
    }
    // sums the hyper-parameter draws need: :603 g.g (RR), :698 sum g^2 of included (C),
    // :791 sum g^2/fold[class] (R); class counts exclude monomorphic markers
    double w = 0.0;
    if (cls_f > 0) w = (model == 6) ? g_f * g_f / pin->fold[cls_f] : g_f * g_f;
    const int nev = cnts[0];
#pragma unroll
    for (int c = 0; c <= K1; c++) {
        const unsigned long long mk = __ballot(active && cls_f == c);
        if (lane == 0 && mk) atomicAdd(&cnts[1 + c], __popcll(mk));
    }
    const double wsum = block_sum(w, red); // two barriers: also publishes the class counts
    double absd = 0.0;
    for (int e = t; e < nev; e += P) {
        v.ev_idx[(size_t)p * P + e] = ev_ix[e] & 0xffff;
        v.ev_delta[(size_t)p * P + e] = ev_del[e];
        absd += fabs(ev_del[e]);
    }
    if (v.mb) { // (uniform)
        absd = block_sum(absd, red);
        if (t == 0) v.mb[(size_t)(1 + p) * HB_MBS] = fma(v.xabs, absd, v.mb[(size_t)p * HB_MBS]);
    }
    if (t == 0) {
        v.ev_count[(size_t)p * HB_EVS] = nev;
        atomicAdd(&v.acc[HB_ACC_SUMG2], wsum);
        atomicAdd(&v.acc[HB_ACC_EVENTS], (double)nev);
    }
    if (t <= K1 && t < HB_MAX_FOLD && cnts[1 + t]) atomicAdd(&v.acc[HB_ACC_COUNT0 + t], (double)cnts[1 + t]);
    HB_STAMP(27);
}

// ---------------------------------------------------------------------------------------------
// k_chain_persist: the same serial chain as k_chain, as ONE workgroup that lives for the whole sweep.
// It walks the panels in order; panel p starts when its reduced dots have been written (dsum[] is pre-filled with
// a NaN pattern), and a group of panels ends by publishing its moves (write-through) and chain_done = last
// panel + 1, which the update row of that group is waiting for.  Because the mat-vec of a later
// panel q may have read a residual that does not contain panel p's moves yet (q's group read version
// g(q) D - Lv - 1), each move is also folded forward into the per-marker corrections (an LDS ring, one slot per panel)
// of the next Lb panels through the band Gram blocks  G_l[q][k][t] = x_{pP+k} . x_{qP+t},  l = q - p.
// ---------------------------------------------------------------------------------------------
struct persist_view {
    int npanels, D, Lv, Lb; // Lb: panels of band the chain folds into ((Lv + 1) D - 1)
    int Lg;                 // band blocks per panel stored in gram[] minus one (>= Lb: one stored band serves every geometry up to it)
    int p0;                 // first panel of this (partial) sweep, a multiple of D; npanels is its END (hb_ctx_sweep_range)
    unsigned *flags;
    const int *slot_of, *hotpack;        // per-sweep row-cache lists from k_hotlist
    const float *thr0f;                  // ... and the opening filter
    double candf;                        // a marker at zero is a chain candidate when q >= candf * thr0 (candf <= 1)
    double *fcorr;                       // k_fwd's corrections (null: the chain folds all Lv D panels ahead itself)
};

#define HB_LBMAX 20
#ifndef HB_APPLY_PREFIX
#define HB_APPLY_PREFIX 1 /* a wave applies only the prefix of a round's moves that can touch it */
#endif
#ifndef HB_FOLD_GATHER
#define HB_FOLD_GATHER 1 /* the fold's moves gathered with one LDS pass + v_readlane */
#endif
#ifndef HB_DECIDE_PAR
#define HB_DECIDE_PAR 1
#endif
#ifndef HB_FAST1
#define HB_FAST1 1 /* single-candidate panels skip the rounds */
#endif
#ifndef HB_NPF
#define HB_NPF 1 /* candidates per panel whose band rows are requested ahead (1 or 2) */
#endif
#define HB_CROWD 8 /* candidates in a round from which their Gram entries are gathered up front */
#ifndef HB_R_EARLY
#define HB_R_EARLY 1 /* with k_fwd beside the chain: the next panel's dots and k_fwd's sums are (re-)requested right after a panel's rounds */
#endif
#ifndef HB_R_FOLDPRE
#define HB_R_FOLDPRE 1 /* ... and the band rows of its first 32 moves before the publish, used after the results */
#endif
#ifndef HB_ROW_TRI
#define HB_ROW_TRI 1 /* panel 512: the row cache keeps a row of the panel's second half as its second 1-KiB piece alone (k_hotlist) */
#endif
#ifndef HB_RING_NODOTS
#define HB_RING_NODOTS 1 /* with k_fwd beside the chain the ring does not fetch the dots three panels ahead: they are never there yet, and the line it read stayed in the XCD's L2 as the copy the early request one panel ahead then got (sentinel at 80 % of the panels; 0 % without) */
#endif
#ifndef HB_FPRE_N
#define HB_FPRE_N 64 /* band rows (moves) requested before the publish: 16, 32, 48 or 64 */
#endif
#ifndef HB_FILL_ALL
#define HB_FILL_ALL 1 /* the ring waves issue their share of the row cache's pieces too (0: the four non-ring waves alone) */
#endif
#ifndef HB_APPLY_PROG
#define HB_APPLY_PROG 0 /* (A/B, off) ... and while the serial pass is still running: every verified block of it publishes its moves' records and the waves at the barrier apply them. Measured: apply + violation barrier 6 500 -> 950 cycles, but the serial pass 7 800 -> 13 000 (the publishing, and SGPR spills in its loop at 254 VGPRs): 52.7 sweeps/s against 52.8 */
#endif
#ifndef HB_APPLY_LEAN
#define HB_APPLY_LEAN 1 /* a crowded round's moves are applied from 16-byte records read with one broadcast LDS load (0: the round-3 loop) */
#endif
#ifndef HB_SPEC_B
#define HB_SPEC_B 16 /* steps per speculated block */
#endif
#ifndef HB_R_SPEC
#define HB_R_SPEC 1 /* crowded rounds of a mixture model: the serial pass in blocks of eight steps on speculated classes */
#endif
#ifndef HB_SERIAL_BRANCHLESS
#define HB_SERIAL_BRANCHLESS 1
#endif

// Row-cache list of every panel, in marker order, capped at nslot rows: the markers that are certain to move
// (polymorphic, g_old != 0) and the markers that are LIKELY to enter the model this sweep. Entry means q >= thr0
// with thr0 already fixed by the marker's uniform draw, and a marker at zero has q ~ xx*vare*chi2_1, so
// "thr0 <= kappa * xx * vare" predicts almost every entry (history does not: re-entry is at chance level).
// A predicted marker only gets its Gram row prefetched; whether it moves is still decided by the chain.
// Produced once per sweep, off the chain's critical path. One workgroup per panel.
#define HB_HS 256 /* ints per panel in the packed hot-list: [0] = rows to cache, [4 ...] = their markers (one 1-KiB DMA piece) */
__global__ __launch_bounds__(512) void k_hotlist(const hb_sweep_in *__restrict__ pin, const double *__restrict__ vx,
                                                 const double *__restrict__ g, const double *__restrict__ thr0,
                                                 const double *__restrict__ xpx, double kappa, int P, int nslot,
                                                 int *__restrict__ slot_of, int *__restrict__ hotpack, float *__restrict__ thr0f,
                                                 uint8_t *__restrict__ tracker)
{
    __shared__ int wcnt[16];
    const int p = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63, S = P >> 6;
    const int j = p * P + t;
    const bool active = vx[j] != 0.0;
    const bool hot = active && (g[j] != 0.0 || thr0[j] <= kappa * xpx[j] * pin->vare);
    // the chain only rewrites the class of markers that are or were in the model: a marker at zero is class 0 by
    // definition, whatever state the caller may have installed
    if (g[j] == 0.0) tracker[j] = 0;
    // The chain's opening filter, 4 bytes per marker (it travels to the chain's LDS by DMA): NaN = monomorphic marker
    // (skipped, src/Bayes.cpp:589), -inf = in the model (certain to move), else the entry threshold on q = rhs^2 rounded
    // DOWN to float — a superset test; whoever passes it is decided with the exact fp64 threshold.
    {
        float f;
        if (!active) f = __int_as_float(0x7fc00000);
        else if (g[j] != 0.0) f = -__int_as_float(0x7f800000);
        else {
            const double th = thr0[j];
            f = (float)th;
            if ((double)f > th) f = nextafterf(f, -__int_as_float(0x7f800000));
        }
        thr0f[j] = f;
    }
    const unsigned long long hmask = __ballot(hot);
    if (lane == 0) wcnt[wave] = __popcll(hmask);
    __syncthreads();
    int sbase = 0, tot = 0;
    for (int w = 0; w < S; w++) {
        const int c = wcnt[w];
        sbase += (w < wave) ? c : 0;
        tot += c;
    }
    const int raw = sbase + __popcll(hmask & ((1ull << lane) - 1ull));
    // Where a listed row sits in the chain's row cache, in units of 64 ints: base64 * 64 + column. Row k is only ever used at
    // columns > k (a move touches later markers), so at panel 512 — two 1-KiB pieces per row — a row of the panel's second half is
    // kept as its second piece alone: the cache holds a third more rows in the same LDS (HB_ROW_TRI). The list is in marker order,
    // so the whole rows (n2 of them) come first; a half row's base points 256 columns before its piece (shifted by one piece when
    // there is no whole row before it, so that no base is negative). [0] = rows that fit, [1] = rows listed, [2] = whole rows among
    // those that fit, [3] = that shift, in pieces.
    const bool tri = HB_ROW_TRI && P == 512;
    int n2 = tot;
    if (tri) {
        n2 = 0;
        for (int w = 0; w < S / 2; w++) n2 += wcnt[w];
    }
    const int U = max(P >> 6, 1), Uh = U >> 1, cap64 = nslot * U;
    const int shift = (tri && n2 == 0) ? Uh : 0;
    const int off64 = raw < n2 ? raw * U : n2 * U + (raw - n2) * Uh + shift;
    const int len64 = raw < n2 ? U : Uh;
    const bool fits = hot && off64 + len64 <= cap64;
    const int slot = fits ? (raw < n2 ? off64 : off64 - Uh) : -1;
    slot_of[j] = active ? slot : -2; // -2: monomorphic marker, skipped by the chain
    // (the list goes on past the rows that got a slot, up to the 252 entries a piece holds: k_warm pulls those rows into the chain's
    // L2 as well — a candidate without a slot then costs the chain an L2 hit instead of a trip to memory)
    if (hot && raw < HB_HS - 4) hotpack[(size_t)p * HB_HS + 4 + raw] = t;
    if (t == 0) {
        int count;
        if (n2 * U >= cap64) count = cap64 / U;
        else count = n2 + (tri ? min(tot - n2, max(0, (cap64 - n2 * U - shift) / Uh)) : 0);
        count = min(count, HB_HS - 4);
        hotpack[(size_t)p * HB_HS] = count;
        hotpack[(size_t)p * HB_HS + 1] = min(tot, HB_HS - 4);
        hotpack[(size_t)p * HB_HS + 2] = min(n2, count);
        hotpack[(size_t)p * HB_HS + 3] = shift ? 1 : 0;
    }
}

// Forward corrections of one batch shape: FW moves x up to LB band blocks, all loads in flight together.
// Panel q = p + l needs the correction iff its mat-vec group read a residual without panel p's moves:
// q / D <= p / D + Lv, i.e. l <= (Lv + 1) D - 1 - p mod D — a contiguous range 1..lcount, computed once per panel by the
// caller (no division here).
template <int LB, int FW>
__device__ __forceinline__ void fold_forward(double *corrL, int R, const int32_t *__restrict__ gram, int Lb, int lcount, int pslot,
                                             int P, int t, int nev, const int *ev_ix, const double *ev_del, int p)
{
    const size_t PP = (size_t)P * P, step = (size_t)(Lb + 2) * PP;
    for (int e0 = 0; e0 < nev; e0 += FW) {
        int gv[LB][FW];
        int kk[FW];
        double dl[FW];
        if (HB_FOLD_GATHER && FW >= 8) { // (the wide batches of the narrow bands: dense sweeps; two moves at a time gain nothing)
            // the batch's moves in ONE pass over LDS: lane f reads move e0 + f, every lane then takes them lane by lane
            // (v_readlane: wave-uniform row addresses without a read-and-wait per move); a lane past the list holds row 0, delta 0
            const int lane_ = t & 63, e = e0 + lane_;
            const bool have = lane_ < FW && e < nev;
            const int ixl = have ? ev_ix[e] : 0;
            const double dll = have ? ev_del[e] : 0.0;
#pragma unroll
            for (int f = 0; f < FW; f++) {
                kk[f] = __builtin_amdgcn_readlane(ixl, f) & 0xffff;
                dl[f] = readlane_f64(dll, f);
            }
        } else {
#pragma unroll
            for (int f = 0; f < FW; f++) {
                const int e = min(e0 + f, nev - 1);
                kk[f] = __builtin_amdgcn_readfirstlane(ev_ix[e] & 0xffff); // wave-uniform: the row addresses below are scalar
                dl[f] = (e0 + f < nev) ? ev_del[e] : 0.0;
            }
        }
        // block l of panel p + l starts at ((p + l)(Lb + 1) + l) P P: consecutive l are (Lb + 2) P P apart
        const int32_t *blk = gram + ((size_t)(p + 1) * (Lb + 1) + 1) * PP;
#pragma unroll
        for (int l = 1; l <= LB; l++) {
            if (l <= lcount) { // uniform
#pragma unroll
                for (int f = 0; f < FW; f++) gv[l - 1][f] = (blk + (size_t)kk[f] * P)[t];
            }
            blk += step;
        }
        int slot = pslot; // ring slot of panel p + l
#pragma unroll
        for (int l = 1; l <= LB; l++) {
            slot = (slot + 1 == R) ? 0 : slot + 1;
            if (l <= lcount) {
                double *cp = corrL + (size_t)slot * P + t; // this thread's own word: no synchronisation needed
                double acc = *cp;
#pragma unroll
                for (int f = 0; f < FW; f++) acc = fma((double)gv[l - 1][f], dl[f], acc);
                *cp = acc;
            }
        }
    }
}

// the eight per-wave words of a small LDS array in two vector reads (a panel has at most 8 waves; absent waves' words are 0)
__device__ __forceinline__ void hb_read8(const int *w, int (&o)[8])
{
    const int4 a = *reinterpret_cast<const int4 *>(w), b = *reinterpret_cast<const int4 *>(w + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// Software-pipelined version: everything panel p+1 needs that does not depend on panel p's outcome is fetched
// while panel p's serial turns run — its per-marker coefficients, its mat-vec partials (if that mat-vec has
// already finished) and the Gram rows of its hot markers (into the other half of a double-buffered LDS row
// cache, two 1-KiB pieces per wave per turn boundary).
// NPL: band blocks whose rows are requested ahead for a panel's first two candidates (== Lb, or 0: none) — a template
// parameter because the counted waits that keep those loads in flight need the count at compile time.
template <int K1, int NPL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chain_persist(const hb_sweep_in *__restrict__ pin, chain_view v, persist_view pv,
                                                       int nslot)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int P = v.P, S = P >> 6;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    // ONE row cache of nslot rows: the next panel's rows are requested (LDS-DMA) as the last thing of a panel, after the last
    // barrier of its rounds — nobody reads the cache between that barrier and the next panel's first round, which drains the
    // pieces — so the fill can go straight on top of the rows just used and the LDS a second buffer would take holds rows instead
    int32_t *rowc0 = reinterpret_cast<int32_t *>(smem);
    char *base = smem + (size_t)nslot * P * 4;
    double *ev_del = reinterpret_cast<double *>(base);
    int *ev_ix = reinterpret_cast<int *>(base + (size_t)P * 8);
    double *red = reinterpret_cast<double *>(base + (size_t)P * 12);
    int *cnts = reinterpret_cast<int *>(base + (size_t)P * 12 + 128);
    int *s_thi = cnts + 18;   // first candidate left for the next round
    int *wcnt0 = cnts + 32;   // candidates per wave: [32..47] even panels, [64..79] odd panels
    int *wviol = cnts + 48;   // wave saw a mis-speculated marker
    // staging of one round's candidates (<= 64): [field][candidate]
    double *cs_d = reinterpret_cast<double *>(base + (size_t)P * 12 + 128 + 512); // rhs, gold, thr[K1], invv[K1], sdz[K1]
    double *res_g = cs_d + (2 + 3 * K1) * 64;
    int *cs_t = reinterpret_cast<int *>(res_g + 64);
    int *cs_slot = cs_t + 64;
    int *res_c = cs_slot + 64;
    int *cg = res_c + 64; // Gram entries among one round's candidates: cg[k * 64 + c] = x_k . x_c for k < c

    const int model = pin->model_index;
    const int count_pip = pin->count_pip, store = pin->store;
    const int lgP = 31 - __clz(P);
    const int np = pv.npanels;
    // corrections still owed to the next Lb panels: ring of Lb + 1 slots of P doubles in LDS, slot = panel mod ring size;
    // every thread only ever touches its own column, so the ring needs no barrier
    const int R = pv.Lb + 1;
    double *corrL = reinterpret_cast<double *>(cg + 64 * 64);
    for (int l = 0; l < R; l++) corrL[(size_t)l * P + t] = 0.0;
    // opening ring (see below): HB_RD slots of [P reduced dots: 8 B][P filter words: 4 B][pad to 1 KiB][1 KiB packed hot-list]
    const int OSZ = ((12 * P + 1023) >> 10) << 10, OSLOT = OSZ + 1024, NPC = (OSZ >> 10) + 1; // NPC: DMA pieces per group
    char *oring = reinterpret_cast<char *>(corrL + (size_t)R * P);
    // with k_fwd beside the chain (pv.fcorr; BayesR at panel 512, one panel per group): what the panels two and more before a
    // panel owe it arrives through fcorr[] — brought into this two-slot LDS ring by LDS-DMA one panel ahead, see below — and the
    // chain itself folds a panel's moves into the NEXT panel only (half of the band rows of a dense sweep leave its compute unit)
    const bool fwd = pv.fcorr != nullptr;
    double *fcring = reinterpret_cast<double *>(oring + (size_t)4 * (((((size_t)12 * P + 1023) >> 10) << 10) + 1024));
    // one crowded round's moves as the apply reads them (HB_APPLY_LEAN): {byte offset of the row in the row cache, marker, change}
    // for the moves whose row is cached — 64 + 8 records, the list is padded with changes of zero — and {-, marker, change} for the others
    int4 *ap_rec = reinterpret_cast<int4 *>(fcring + (size_t)2 * P);
    int4 *ms_rec = ap_rec + 72;
    int pslot = -1; // p mod R
    double wacc = 0.0;
    int cacc[K1 + 1];
#pragma unroll
    for (int c = 0; c <= K1; c++) cacc[c] = 0;
    int evacc = 0, missacc = 0, redoacc = 0;
    double mbr = v.mb ? v.mb[0] : 0.0; // running bound on max |yadj| (kept by the publishing wave)
    int gcount = pv.p0 / pv.D;          // mat-vec groups published so far (absolute group index)

    // ---- the opening ring ----
    // What the opening of a panel needs — its reduced dots and one filter word per marker (k_hotlist: NaN monomorphic, -inf in
    // the model, else the entry threshold rounded down) — plus the next panel's row-cache list travel to LDS by LDS-DMA
    // (global_load_lds_dwordx4, 1 KiB per instruction), HB_RD - 1 panels ahead: with the mat-vec streaming at full rate a load
    // takes microseconds, far longer than a quiet panel lasts, and a register prefetch ring does not survive hipcc (a loaded
    // register that lives across the loop edge is copied, and the copy waits: every panel paid two loaded round trips).
    // A DMA piece has no destination register, so the only waits are the ones written here: wave 0 issues all pieces of a
    // group and, at the top of each panel, lets at most the youngest group stay in flight (counted vmcnt; everything the
    // next panel's take needs has then landed, and the panel's one barrier publishes it to the other waves).
    // A panel without candidates touches no global memory at all. A panel with candidates fetches the exact per-marker
    // data (thresholds, conditional-mean coefficients, old effect, x'x, row-cache slot) on the spot: one round trip.
    // The reduced dots need no flag: the sweep starts with dsum[] filled with a NaN bit pattern no sum can produce,
    // every 8-byte result lands atomically, so a value is either that pattern (not there yet: re-read) or final.
    constexpr int HB_RD = 4;
    constexpr long long HB_SENT = -1ll; // memset 0xFF
    const unsigned oring_lds = (unsigned)(uintptr_t)oring;
    // group G(x) = dots and filter of panel x + row-cache list of panel x + 1, into ring slot x mod HB_RD
    // the NPC pieces of a group are dealt round-robin to the first RW waves (half of the workgroup; the other half fills the row
    // cache), so that a ring wave's memory queue holds ring pieces only — which is what makes its counted wait exact
    const int RW = S > 1 ? (S >> 1) : 1;
    // (a piece's source is "wave-uniform base + 16 bytes per lane" whenever the dots' and the filter's segments of a panel are
    // whole pieces, P >= 128: the global_load_lds form with a scalar base and a loop-invariant lane offset then needs no vector
    // arithmetic and no vector temporaries per issue — hipcc guards a reused temporary with a vmcnt wait, which would stall the
    // issue behind whatever the panel still has in flight)
    const unsigned lane16 = (unsigned)lane * 16;
    auto dma_piece_s = [&](const char *sbase_, unsigned lds_dst_, bool fresh) {
        // (values that ARE wave-uniform, but that hipcc may have computed on the vector unit when scalar registers ran short)
        const unsigned long long sb = (unsigned long long)(uintptr_t)sbase_;
        const char *sbase = reinterpret_cast<const char *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sb >> 32)) << 32) |
                                                                       (unsigned)__builtin_amdgcn_readfirstlane((int)sb)));
        const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_);
        unsigned keep;
        if (fresh)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(sbase), "s"(lds_dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(sbase), "s"(lds_dst) : "memory");
    };
    auto issue_group = [&](int x, int slot) {
        const unsigned dst = oring_lds + (unsigned)slot * OSLOT;
        const char *dsrc = reinterpret_cast<const char *>(v.dsum + (size_t)x * P);
        const char *fsrc = reinterpret_cast<const char *>(pv.thr0f + (size_t)x * P);
        const char *hsrc = reinterpret_cast<const char *>(pv.hotpack + (size_t)min(x + 1, np - 1) * HB_HS);
        const int w = __builtin_amdgcn_readfirstlane(wave);
        for (int i = w; i < NPC; i += RW) {
            if (i == NPC - 1) {
                dma_piece_s(hsrc, dst + (unsigned)OSZ, false);
            } else if (P >= 128) {
                const int off = i << 10; // whole piece inside one segment
                // (with k_fwd beside the chain the dots come with the early request one panel ahead: three panels ahead they are
                // never there yet, and the line read now would be the copy the early request then finds in this XCD's L2)
                if (HB_RING_NODOTS && HB_R_EARLY && fwd && off < 8 * P && x > pv.p0) continue;
                dma_piece_s(off < 8 * P ? dsrc + off : fsrc + (off - 8 * P), dst + (unsigned)off, true);
            } else {
                const int off = (i << 10) + lane * 16;
                if (off < 12 * P) {
                    const char *src = off < 8 * P ? dsrc + off : fsrc + (off - 8 * P);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst + ((unsigned)i << 10)))
                                 : "memory");
                }
            }
        }
    };
    int my_pieces = __builtin_amdgcn_readfirstlane(wave < RW ? (NPC - wave + RW - 1) / RW : 0); // ring pieces this wave issues per group
    if (HB_RING_NODOTS && HB_R_EARLY && fwd && P >= 128 && wave < RW) { // (without the dots' pieces)
        int c = 0;
        for (int i = __builtin_amdgcn_readfirstlane(wave); i < NPC; i += RW) c += (i == NPC - 1 || (i << 10) >= 8 * P) ? 1 : 0;
        my_pieces = c;
    }
    int my_rowp = 0; // row-cache pieces this (ring) wave issued behind its last ring group
    int n_nhot = 0;

    // ---- prologue ----
    if (t == 0) { // where this workgroup runs: k_warm's workgroups on the same XCD (= the same L2) fetch ahead of it
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st_flag(pv.flags + HB_FLAG_XCC, (xcc & 15u) + 1u);
    }
    for (int i = t; i < 128; i += P) cnts[i] = 0; // (a 64-marker panel has 64 threads; absent waves' words must read 0)
    for (int i = t; i < 72 + 64; i += P) ap_rec[i] = make_int4(0, 0x7fffffff, 0, 0); // (a record read ahead of its count must at least address LDS)
    if (wave < RW)
        for (int x = pv.p0; x < pv.p0 + HB_RD - 1 && x < np; x++) issue_group(x, x - pv.p0);
    bool ok = true;
    {   // the row cache for the first panel
        const int *hl0 = pv.hotpack + (size_t)pv.p0 * HB_HS;
        n_nhot = hl0[0];
        const int32_t *gp0 = v.gram + (size_t)pv.p0 * (pv.Lg + 1) * P * P;
        if (HB_ROW_TRI && P == 512) { // (the layout k_hotlist describes: whole rows first, then second pieces alone)
            const int n2s = hl0[2], sh = hl0[3], items = n_nhot + n2s;
            for (int it = wave; it < items; it += S) {
                const int r = it < 2 * n2s ? it >> 1 : it - n2s, pc = it < 2 * n2s ? (it & 1) << 8 : 256;
                const int k = hl0[4 + r];
                *reinterpret_cast<int4 *>(rowc0 + ((it + sh) << 8) + lane * 4) = *reinterpret_cast<const int4 *>(gp0 + ((size_t)k << lgP) + pc + lane * 4);
            }
        } else {
            const int total = n_nhot << lgP, items = (total + 255) >> 8;
            for (int it = wave; it < items; it += S) {
                const int lin = min((it << 8) + lane * 4, total - 4);
                const int k = hl0[4 + (lin >> lgP)];
                *reinterpret_cast<int4 *>(rowc0 + lin) = *reinterpret_cast<const int4 *>(gp0 + ((size_t)k << lgP) + (lin & (P - 1)));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int oslot = -1; // p mod HB_RD
    int pmodD = -1; // p mod D, without a division per panel
    for (int p = pv.p0; ok && p < np; p++) {
        pmodD = (pmodD + 1 == pv.D) ? 0 : pmodD + 1;
        pslot = (pslot + 1 == R) ? 0 : pslot + 1;
        oslot = (oslot + 1 == HB_RD) ? 0 : oslot + 1;
        const int j = p * P + t;
        const int cur = p & 1;
        int32_t *rowc = rowc0;
        int32_t *rown = rowc0;
        const int32_t *gp = v.gram + (size_t)p * (pv.Lg + 1) * P * P;
        int *wcnt = wcnt0 + (cur << 5);
        const char *oslotp = oring + (size_t)oslot * OSLOT;
        HB_STAMP(0);
        // ring waves: the group of panel p + 1 has landed once at most the youngest group (panel p + 2's) is still in flight;
        // the barrier below hands it to everybody before the next panel's take
        if (wave < RW) {
            // (what may stay in flight: the youngest ring group — none was issued behind the previous panel near the end of the range —
            // and, HB_FILL_ALL, the row-cache pieces this wave issued behind it: the counter wants an immediate, hence the ladder)
            const int keep = (S == 1) ? 0 : ((p + HB_RD - 2 < np || p == pv.p0) ? my_pieces : 0) + my_rowp;
            switch (min(keep, 31)) {
#define HB_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
                HB_VMC(0) HB_VMC(1) HB_VMC(2) HB_VMC(3) HB_VMC(4) HB_VMC(5) HB_VMC(6) HB_VMC(7) HB_VMC(8) HB_VMC(9) HB_VMC(10) HB_VMC(11)
                HB_VMC(12) HB_VMC(13) HB_VMC(14) HB_VMC(15) HB_VMC(16) HB_VMC(17) HB_VMC(18) HB_VMC(19) HB_VMC(20) HB_VMC(21) HB_VMC(22)
                HB_VMC(23) HB_VMC(24) HB_VMC(25) HB_VMC(26) HB_VMC(27) HB_VMC(28) HB_VMC(29) HB_VMC(30) HB_VMC(31)
#undef HB_VMC
            }
        }
        // ---- take over the panel: LDS only ----
        const bool use_fc = fwd && p >= pv.p0 + 2; // (the first two panels of a range have nobody two panels before them)
        // (k_fwd's sums — and, HB_R_EARLY, a second copy of the panel's dots — were brought in by the ring waves during the previous
        // panel, after its barrier — their producers need the panels before — so unlike the ring groups no earlier barrier has handed
        // them to the other waves yet: one extra barrier per panel, a few hundred cycles)
        HB_STAMP(20);
#if HB_STAMPS
        if (v.dbg && lane == 0 && wave < 8) v.dbg[(size_t)p * 32 + 22 + wave] = clock64(); // (each wave's arrival at the barrier)
#endif
        if (HB_R_EARLY ? fwd : use_fc) __syncthreads();
        HB_STAMP(21);
        double dj = reinterpret_cast<const double *>(oslotp)[t];
        const float fthr = reinterpret_cast<const float *>(oslotp + 8 * P)[t];
        double fcv = use_fc ? fcring[(size_t)(p & 1) * P + t] : 0.0;
        bool aborted = false;
        {
            bool bad = __double_as_longlong(dj) == HB_SENT, badf = use_fc && __double_as_longlong(fcv) == HB_SENT;
            HB_STAMP_VAL(11, bad ? 1 : 0);
            if (__any(bad || badf)) { // this wave's dots (or k_fwd's sums) had not been written when the ring slot was filled: re-read until they are
                const unsigned long long t0 = wall_clock64();
                unsigned looks = 0;
                for (;;) {
                    const bool fresh = hb_fresh_look(looks);
                    if (bad) {
                        dj = fresh ? ld_fresh(&v.dsum[j]) : ld_sc1(&v.dsum[j]);
                        bad = __double_as_longlong(dj) == HB_SENT;
                    }
                    if (badf) {
                        fcv = fresh ? ld_fresh(&pv.fcorr[j]) : ld_sc1(&pv.fcorr[j]);
                        badf = __double_as_longlong(fcv) == HB_SENT;
                    }
                    if (!__any(bad || badf)) break;
                    const bool own = wall_clock64() - t0 > HB_TIMEOUT_TICKS;
                    if (ld_flag(pv.flags + HB_FLAG_ABORT) || own) {
                        if (lane == 0) st_flag(pv.flags + HB_FLAG_ABORT, 1u);
                        const unsigned long long bm = __ballot(bad || badf);
                        if (bm && lane == __ffsll((long long)bm) - 1) hb_abort_log(pv.flags, HB_LOG_GROUP, own, bad ? 1u : 2u, (unsigned)p, ~0ull);
                        aborted = true;
                        break;
                    }
                    hb_poll_pause(looks, 1);
                    hb_long_wait(looks);
                    looks++;
                }
            }
        }
        double corrv;
        {
            double *cp = corrL + (size_t)pslot * P + t;
            corrv = *cp + fcv;
            *cp = 0.0; // the slot is panel p + R's from now on
        }
        const bool active = fthr == fthr;                          // not NaN: a polymorphic marker
        const bool hot = fthr == -__int_as_float(0x7f800000);      // in the model: certain to move
        const bool have_next = p + 1 < np;
        const bool group_end = have_next && pmodD == pv.D - 1;
        const int32_t *gpn = gp + (size_t)(pv.Lg + 1) * P * P;
        // who can move at all: certain movers and markers whose q reaches the (rounded-down) entry threshold. For a marker at
        // zero rhs = d - corrections; the exact test follows in the chain.
        bool cand0;
        {
            const double r0 = dj - corrv;
            cand0 = active && (hot || r0 * r0 >= pv.candf * (double)fthr);
            const unsigned long long cm0 = __ballot(cand0);
            // count | lane of the wave's first candidate << 8 | lane of its second << 14 | gave-up-waiting << 24
            const unsigned long long cm1 = cm0 & (cm0 - 1ull);
            if (lane == 0)
                wcnt[wave] = __popcll(cm0) | (cm0 ? (__ffsll((long long)cm0) - 1) << 8 : 0) | (cm1 ? (__ffsll((long long)cm1) - 1) << 14 : 0) |
                             (aborted ? 1 << 24 : 0);
        }
        HB_STAMP(1);
        __syncthreads(); // the panel's one fixed barrier: wcnt[] staged, ring group of panel p + 1 published; everybody is done with panel p-1
        int tot0 = 0, c1 = -1, c2 = -1; // candidates in the panel; its first two (thread = marker index in the panel)
        {
            int w8[8];
            hb_read8(wcnt, w8);
            const int any = w8[0] | w8[1] | w8[2] | w8[3] | w8[4] | w8[5] | w8[6] | w8[7]; // a quiet panel decodes nothing
            if (any >> 24) { ok = false; break; } // a wave gave up waiting for its dots: the sweep is aborted
            if (any & 0xff) {
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    const int cnt = w8[w] & 0xff, a = w * 64 + ((w8[w] >> 8) & 63), b = w * 64 + ((w8[w] >> 14) & 63);
                    tot0 += cnt;
                    if (cnt) {
                        if (c1 < 0) { c1 = a; c2 = cnt > 1 ? b : -1; }
                        else if (c2 < 0) c2 = a;
                    }
                }
            }
        }
        // (2) a panel with candidates: the exact per-marker data, one round trip
        double thr[K1], invv[K1], sdz[K1];
        double gold = 0.0, rhs = 0.0;
        int myslot = -1;
#pragma unroll
        for (int c = 0; c < K1; c++) { thr[c] = HB_INF; invv[c] = 0.0; sdz[c] = 0.0; }

        // ---- the serial chain, speculatively compacted ----
        // Only markers that are in the model (certain to move) or whose q is near their entry threshold can move.
        // Each round compacts the next <= 64 such candidates, in marker order, into the lanes of wave 0, which runs
        // the exact serial chain over them alone; every other marker then applies the round's moves to its own rhs
        // and checks that it really stayed below its threshold. If one did not (a move pushed a non-candidate over),
        // the round is rolled back and repeated with that marker as a candidate — the outcome is always the exact
        // sequential one, the speculation only decides how much of it runs in one wave without barriers.
        int cls_f = 0;
        double g_f = 0.0;
        int nev = 0;
#if HB_STAMPS
        int nrerun = 0, nround = 0;
        HB_STAMP_VAL(15, tot0);
#endif
        int pre[2][NPL > 0 ? NPL : 1];
        if (tot0 > 0) {
            // the exact per-marker data, for the candidates only (one CU pulls ~18 bytes per clock from memory — measured,
            // tools/rowfetch_bench.hip — and every thread's copy of six arrays was a quarter of a move-panel's traffic). A marker
            // that is not a candidate is at zero; until it becomes one it is judged with its filter word (the entry threshold
            // rounded down: a superset test) and fetches its data then.
            double xx = 0.0;
            bool have_exact = cand0;
            if (cand0) {
                gold = v.g[j];
                xx = v.xpx[j];
                myslot = pv.slot_of[j];
#pragma unroll
                for (int c = 0; c < K1; c++) {
                    thr[c] = v.thr[(size_t)c * v.m_pad + j];
                    invv[c] = v.invv[(size_t)c * v.m_pad + j];
                    sdz[c] = v.sdz[(size_t)c * v.m_pad + j];
                }
            }
            const double thr_lo = (double)fthr; // <= thr[0]; NaN for a monomorphic marker (every comparison false)
            // (3) ... and, requested right behind it, the band-Gram rows that the panel's first two candidates would fold forward if
            // they move (in the sparse regime a candidate almost always does, and a panel rarely has more than two): by the time the
            // rounds are through they have landed, and the fold at the end of the panel costs no round trip. Always 2 * NPL loads, so
            // that the counted waits below are exact; rows of panels that do not exist are read from the panel's own block.
            if (NPL > 0) {
                __builtin_amdgcn_sched_barrier(0); // (the order of issue is the point: hipcc must not move these ahead of the data above)
                const int lmax = np - 1 - p;
                const size_t PP = (size_t)P * P, step = (size_t)(pv.Lg + 2) * PP;
                const int k1 = __builtin_amdgcn_readfirstlane(c1), k2 = __builtin_amdgcn_readfirstlane(c2 < 0 ? c1 : c2);
                const int32_t *blk = v.gram + ((size_t)(p + 1) * (pv.Lg + 1) + 1) * PP;
#pragma unroll
                for (int l = 1; l <= NPL; l++) {
                    const int32_t *b = l <= lmax ? blk : gp;
                    pre[0][l - 1] = (b + (size_t)k1 * P)[t];
                    if (HB_NPF > 1) pre[1][l - 1] = (b + (size_t)k2 * P)[t];
                    blk += step;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            rhs = dj;
            if (gold != 0.0) rhs = fma(xx, gold, rhs);
            rhs -= corrv;
            HB_STAMP(7);
            int t_lo = 0, nev0 = 0;
            bool forced = false;
            bool first = true; // the first round's candidate counts were staged before the panel's opening barrier
            // ---- a panel with ONE candidate (most panels with a move in the sparse regime): no compaction, no serial pass ----
            // The candidate publishes its numbers, everybody takes the same decision from them (the chain's own arithmetic),
            // applies the move to its own rhs and checks that it stayed below its threshold: two barriers instead of four or
            // five. A marker pushed over its threshold sends the panel through the general rounds below, exactly as a
            // rolled-back round would.
            bool fast_done = false;
            if (HB_FAST1 && tot0 == 1) {
                if (t == c1) {
                    cs_d[0] = rhs;
                    cs_d[64] = gold;
#pragma unroll
                    for (int c = 0; c < K1; c++) {
                        cs_d[(2 + c) * 64] = thr[c];
                        cs_d[(2 + K1 + c) * 64] = invv[c];
                        cs_d[(2 + 2 * K1 + c) * 64] = sdz[c];
                    }
                    cs_slot[0] = myslot;
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HB_NPF * NPL) : "memory"); // (the row cache's DMA pieces, as in the rounds)
                __syncthreads();
                const double crhs = cs_d[0], cgold = cs_d[64];
                const int cslot = cs_slot[0];
                const double q = crhs * crhs;
                const double cthr0 = cs_d[2 * 64];
                double iv = cs_d[(2 + K1) * 64], sz = cs_d[(2 + 2 * K1) * 64];
                int cls = q >= cthr0 ? 1 : 0;
#pragma unroll
                for (int c = 1; c < K1; c++) {
                    const bool ge = q >= cs_d[(2 + c) * 64];
                    cls += ge ? 1 : 0;
                    iv = ge ? cs_d[(2 + K1 + c) * 64] : iv;
                    sz = ge ? cs_d[(2 + 2 * K1 + c) * 64] : sz;
                }
                double gn = (q >= cthr0) ? fma(crhs, iv, sz) : 0.0;
                if (K1 == 1 && model == 5 && fabs(gn) < 1e-6) gn = 1e-6;
                const bool sel = cgold != 0.0 || q >= cthr0;
                const int rc = sel ? cls : 0;
                const double rg = sel ? gn : 0.0;
                const double dk = rg - cgold;
                double rhs_new = rhs;
                if (dk != 0.0) { // uniform
                    int gv = rowc[(max(cslot, 0) << 6) + t];
                    if (cslot < 0) gv = gp[(size_t)c1 * P + t];
                    if (t > c1) rhs_new = fma(-(double)gv, dk, rhs);
                    if (t == c1) { ev_ix[0] = (cslot << 16) | c1; ev_del[0] = dk; }
                }
                const bool viol = t != c1 && active && rhs_new * rhs_new >= thr_lo;
                const unsigned long long vm = __ballot(viol);
                if (lane == 0) wviol[wave] = vm != 0ull;
                __syncthreads();
                bool anyv = false;
                {
                    int w8[8];
                    hb_read8(wviol, w8);
                    anyv = (w8[0] | w8[1] | w8[2] | w8[3] | w8[4] | w8[5] | w8[6] | w8[7]) != 0;
                }
                if (!anyv) {
                    fast_done = true;
                    rhs = rhs_new;
                    if (t == c1) { cls_f = rc; g_f = rg; }
                    nev = dk != 0.0 ? 1 : 0;
                    if (wave == 0) missacc += (dk != 0.0 && cslot < 0) ? 1 : 0;
                } else { // as a rolled-back round: the markers that crossed join the candidates
                    forced = viol;
                    first = false;
                    if (t == 0) redoacc++;
                }
            }
            if (!fast_done) {
            for (;;) {
                const bool undec = t >= t_lo;
                // (the first round's counts were taken with the opening filter: the same predicate must rank them)
                const bool isc = first ? cand0 : (undec && active && (hot || forced || rhs * rhs >= pv.candf * thr_lo));
                const unsigned long long cm = __ballot(isc);
                if (isc && !have_exact) { // (rare: more than 64 candidates, or a marker pushed over its threshold by a move)
                    myslot = pv.slot_of[j];
#pragma unroll
                    for (int c = 0; c < K1; c++) {
                        thr[c] = v.thr[(size_t)c * v.m_pad + j];
                        invv[c] = v.invv[(size_t)c * v.m_pad + j];
                        sdz[c] = v.sdz[(size_t)c * v.m_pad + j];
                    }
                    have_exact = true;
                }
                if (!first) {
                    if (lane == 0) wcnt[wave] = __popcll(cm);
                    __syncthreads();
                }
                first = false;
                int basec = 0, tot = 0;
                {
                    int w8[8];
                    hb_read8(wcnt, w8);
#pragma unroll
                    for (int w = 0; w < 8; w++) {
                        const int c = w8[w] & 0xff;
                        basec += (w < wave) ? c : 0;
                        tot += c;
                    }
                }
                if (tot == 0) break; // nobody left can move
#if HB_STAMPS
                nround++;
#endif
                const int rank = basec + __popcll(cm & ((1ull << lane) - 1ull));
                const bool inr = isc && rank < 64;
                const int ncr = min(tot, 64);
                if (isc && rank == 64) *s_thi = t;
                if (t == 0) { cnts[2] = 0; cnts[3] = 0; cnts[4] = 0; } // (this round's records: none yet; the barrier below orders it against the serial pass)
                if (inr) {
                    cs_d[rank] = rhs;
                    cs_d[64 + rank] = gold;
#pragma unroll
                    for (int c = 0; c < K1; c++) {
                        cs_d[(2 + c) * 64 + rank] = thr[c];
                        cs_d[(2 + K1 + c) * 64 + rank] = invv[c];
                        cs_d[(2 + 2 * K1 + c) * 64 + rank] = sdz[c];
                    }
                    cs_t[rank] = t;
                    cs_slot[rank] = myslot;
                }
                // the row cache was filled by LDS-DMA a panel ago: every wave drains its own pieces before the barrier — everything
                // older than the 2 * npl candidate rows requested above, which may stay in flight (the queue completes in order)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HB_NPF * NPL) : "memory");
                __syncthreads();
                if (t_lo == 0 && nev0 == 0 && !forced) HB_STAMP(12);
                const int t_hi = tot > 64 ? *s_thi : P;
                // a crowded round (dense models): the candidates' mutual Gram entries are gathered by everybody first (from the
                // row cache; a candidate without a slot costs one parallel global fetch here instead of a serial one inside
                // the chain)
                const bool crowded = ncr >= HB_CROWD; // uniform: below that the chain reads the row cache itself
                for (int base = t; crowded && base < ncr * 64; base += 8 * P) { // eight entries per thread in flight
                    int gval[8];
#pragma unroll
                    for (int u8 = 0; u8 < 8; u8++) {
                        const int idx = base + u8 * P, k = idx >> 6, c = idx & 63;
                        gval[u8] = 0;
                        if (idx < ncr * 64 && k < c && c < ncr) {
                            const int sk = cs_slot[k];
                            gval[u8] = rowc[(max(sk, 0) << 6) + cs_t[c]];
                            if (sk < 0) gval[u8] = gp[(size_t)cs_t[k] * P + cs_t[c]];
                        }
                    }
#pragma unroll
                    for (int u8 = 0; u8 < 8; u8++) {
                        const int idx = base + u8 * P;
                        if (idx < ncr * 64) cg[idx] = gval[u8];
                    }
                }
                if (crowded) __syncthreads(); // (uniform)
                if (t_lo == 0 && nev0 == 0 && !forced) HB_STAMP(18);
                if (wave == 0) {
                    // The exact serial chain over the round's candidates, one per lane in marker order: step k asks whether
                    // lane k moves given everything before it (certain movers always do), broadcasts its change and applies
                    // it to the later lanes with the Gram entries gathered above.
                    const bool lv = lane < ncr;
                    double crhs = cs_d[lane];
                    const double cgold = lv ? cs_d[64 + lane] : 0.0;
                    double cthr[K1], cinvv[K1], csdz[K1];
#pragma unroll
                    for (int c = 0; c < K1; c++) {
                        cthr[c] = cs_d[(2 + c) * 64 + lane];
                        cinvv[c] = cs_d[(2 + K1 + c) * 64 + lane];
                        csdz[c] = cs_d[(2 + 2 * K1 + c) * 64 + lane];
                    }
                    const int ct = lv ? cs_t[lane] : 0;
                    const int cslot = lv ? cs_slot[lane] : -1;
                    const unsigned long long vmask = __ballot(lv);
                    const unsigned long long hotm = __ballot(lv && cgold != 0.0);
                    const unsigned long long noslot = __ballot(lv && cslot < 0);
                    // what each lane draws from its current rhs: class, new effect, change (the lane's own step reads these)
                    auto decide = [&](double rhsv, int &cls, double &gn) {
                        const double q = rhsv * rhsv;
                        cls = 0;
                        // (thresholds ascend, and below thr[0] the result is zeroed anyway: class 1's coefficients need no select)
#if HB_DECIDE_PAR
                        // every class's conditional mean at once (independent fused multiply-adds), then ONE select per class on the
                        // result instead of two on its coefficients: the same number, a shorter dependent chain per serial step
                        double gsel = fma(rhsv, cinvv[0], csdz[0]);
                        cls = q >= cthr[0] ? 1 : 0;
#pragma unroll
                        for (int c = 1; c < K1; c++) {
                            const bool ge = q >= cthr[c];
                            cls += ge ? 1 : 0;
                            gsel = ge ? fma(rhsv, cinvv[c], csdz[c]) : gsel;
                        }
                        gn = (q >= cthr[0]) ? gsel : 0.0; // (class > 0 <=> q >= thr[0])
#else
                        double iv = cinvv[0], sz = csdz[0];
                        cls = q >= cthr[0] ? 1 : 0;
#pragma unroll
                        for (int c = 1; c < K1; c++) {
                            const bool ge = q >= cthr[c];
                            cls += ge ? 1 : 0;
                            iv = ge ? cinvv[c] : iv;
                            sz = ge ? csdz[c] : sz;
                        }
                        gn = (q >= cthr[0]) ? fma(rhsv, iv, sz) : 0.0; // (class > 0 <=> q >= thr[0])
#endif
                        if (K1 == 1 && model == 5 && fabs(gn) < 1e-6) gn = 1e-6; // (BayesL is a one-class model)
                    };
                    if (crowded) {
                        // Dense round: the Gram entries were gathered into cg[][] (zero on and below the diagonal, so a move of
                        // lane k leaves lanes <= k alone without a compare). The loop is software-pipelined around its only
                        // loop-carried value, crhs: row k + 1 of cg is fetched (and converted) while step k decides, a certain
                        // mover needs no ballot, and a zero change needs no branch (it adds an exact zero).
                        int r1 = cg[lane], r2 = cg[(ncr > 1 ? 64 : 0) + lane]; // rows k + 1 and k + 2 in flight (two deep: a read takes ~100 cycles)
                        double gnx = (double)r1;
                        r1 = r2;
                        if (K1 == 1 && (model == 1 || model == 2 || model == 5) && hotm == vmask) {
                            // BayesRR / A / L: every marker is in the model and stays there (thr = -inf), so a step is the
                            // conditional mean, its change, one broadcast and one fused multiply-add — no test, no class
                            for (int k = 0; k < ncr; k++) {
                                const double gcur = gnx;
                                r2 = cg[min(k + 2, ncr - 1) * 64 + lane];
                                double gn = fma(crhs, cinvv[0], csdz[0]);
                                if (model == 5 && fabs(gn) < 1e-6) gn = 1e-6;
                                const double dk = readlane_f64(gn - cgold, k);
                                crhs = fma(-gcur, dk, crhs);
                                gnx = (double)r1;
                                r1 = r2;
                            }
                        } else if (HB_R_SPEC && K1 > 1) {
                            // A mixture model (BayesR: ~60 candidates in a panel, half of them certain movers): a step of the exact
                            // loop below is ~25 dependent instructions, because the class of lane k has to be decided from the rhs
                            // the step before it left. But within a class the new effect is LINEAR in rhs, and a lane's class
                            // rarely changes over the few steps before its own. So: HB_SPEC_B steps at a time on the classes every lane
                            // has NOW (a step is then: one fused multiply-add, the change, its broadcast, one fused multiply-add
                            // per lane), then the classes of the block's lanes are read off their final rhs — nobody touches a
                            // lane's rhs after its own step — and compared with what was assumed. All equal: every step computed
                            // exactly what the exact loop computes (same operands, same operations). One differs: back to the rhs
                            // saved at the block's start and again with the classes just read — the lanes before the first
                            // mismatch were exact and stay so, the mismatching lane now has its exact class, so every repeat
                            // fixes at least one more lane (at most HB_SPEC_B repeats; 0.01 per panel measured).
                            auto classify = [&](double rhsv, int &cls, double &a, double &b) {
                                const double q = rhsv * rhsv;
                                cls = 0; a = 0.0; b = 0.0;
#pragma unroll
                                for (int c = 0; c < K1; c++) {
                                    const bool ge = q >= cthr[c];
                                    cls += ge ? 1 : 0;
                                    a = ge ? cinvv[c] : a;
                                    b = ge ? csdz[c] : b;
                                }
                            };
                            int cls_s, nsp = 0, nmp = 0; // (records published so far: cached rows, others)
                            double a_s, b_s;
                            classify(crhs, cls_s, a_s, b_s);
                            constexpr int SB = K1 > 3 ? 8 : HB_SPEC_B; // (steps per block)
                            for (int k0 = 0; k0 < ncr; k0 += SB) {
                                double grow[SB];
#pragma unroll
                                for (int u = 0; u < SB; u++) grow[u] = (double)cg[min(k0 + u, ncr - 1) * 64 + lane];
                                const double save = crhs;
                                const bool inblk = lv && lane >= k0 && lane < k0 + SB;
                                for (;;) {
#pragma unroll
                                    for (int u = 0; u < SB; u++) {
                                        if (k0 + u < ncr) { // uniform
                                            const double gn = fma(crhs, a_s, b_s); // (class 0: +0, the exact loop's 0.0)
                                            const double dk = readlane_f64(gn - cgold, k0 + u);
                                            crhs = fma(-grow[u], dk, crhs);
                                        }
                                    }
                                    int cls2;
                                    double a2, b2;
                                    classify(crhs, cls2, a2, b2);
                                    const bool mis = inblk && cls2 != cls_s;
                                    cls_s = cls2; a_s = a2; b_s = b2; // (the block's lanes become exact from the front; the later lanes get a fresher guess)
                                    if (!__any(mis)) break;
#if HB_STAMPS
                                    nrerun++;
#endif
                                    crhs = save;
                                }
                                if (HB_APPLY_LEAN && HB_APPLY_PROG) {
                                    // the block is final: its moves go out now, as the records the other waves' apply reads — they
                                    // are standing at the round's barrier otherwise (the same records, at the same places, as the
                                    // listing after the pass writes once more)
                                    const double dmb = fma(crhs, a_s, b_s) - cgold;
                                    const bool mvl = inblk && dmb != 0.0;
                                    const unsigned long long mvb = __ballot(mvl), mvs = mvb & ~noslot, mvm = mvb & noslot, below = (1ull << lane) - 1ull;
                                    if (mvl) {
                                        const long long db = __double_as_longlong(dmb);
                                        if (cslot >= 0) ap_rec[nsp + __popcll(mvs & below)] = make_int4(cslot << 8, ct, (int)db, (int)(db >> 32));
                                        else ms_rec[nmp + __popcll(mvm & below)] = make_int4(0, ct, (int)db, (int)(db >> 32));
                                    }
                                    nsp += __popcll(mvs);
                                    nmp += __popcll(mvm);
                                    if (lane == 0) {
                                        __hip_atomic_store(&cnts[3], nmp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        __hip_atomic_store(&cnts[2], nsp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    }
                                }
                            }
                        } else
                        for (int k = 0; k < ncr; k++) {
                            const double gcur = gnx;
                            r2 = cg[min(k + 2, ncr - 1) * 64 + lane];
#if HB_SERIAL_BRANCHLESS
                            // (no test for "lane k stays at zero": its change is then an exact zero, and a ballot, a scalar test and a
                            // branch per step cost more than the decide they skip)
                            {
                                int cls;
                                double gn;
                                decide(crhs, cls, gn);
                                const double dk = readlane_f64(gn - cgold, k);
                                crhs = fma(-gcur, dk, crhs);
                            }
#else
                            bool stays = false;
                            if (!((hotm >> k) & 1ull)) { // (uniform) a marker at zero moves only if it crosses its entry threshold
                                const unsigned long long mv = __ballot(crhs * crhs >= cthr[0]) & vmask;
                                stays = !((mv >> k) & 1ull);
                            }
                            if (!stays) {
                                int cls;
                                double gn;
                                decide(crhs, cls, gn);
                                const double dk = readlane_f64(gn - cgold, k);
                                crhs = fma(-gcur, dk, crhs);
                            }
#endif
                            gnx = (double)r1; // (landed an iteration ago)
                            r1 = r2;
                        }
                    } else
                    for (int k = 0; k < ncr; k++) {
                        const double q = crhs * crhs;
                        const unsigned long long mv = (__ballot(q >= cthr[0]) & vmask) | hotm;
                        if (!((mv >> k) & 1ull)) continue; // uniform: lane k stays where it is
                        int cls;
                        double gn;
                        decide(crhs, cls, gn);
                        const double dk = readlane_f64(gn - cgold, k);
                        if (dk != 0.0) {
                            // (the LDS read is unconditional on purpose: a select between an LDS and a global address becomes one
                            // flat load, and a flat load waits for every outstanding vector-memory operation)
                            int gv;
                            if (crowded) {
                                gv = cg[k * 64 + lane];
                                asm volatile("" : "+v"(gv)); // (keeps the three loads apart)
                            } else {
                                const int sk = __builtin_amdgcn_readlane(cslot, k);
                                gv = rowc[(max(sk, 0) << 6) + ct];                                                // always: LDS
                                asm volatile("" : "+v"(gv));
                                if (sk < 0) {
                                    gv = gp[(size_t)__builtin_amdgcn_readlane(ct, k) * P + ct];               // a miss: global
                                    asm volatile("" : "+v"(gv));
                                }
                            }
                            if (lane > k) crhs = fma(-(double)gv, dk, crhs);
                        }
                    }
                    // Lane k's rhs is not touched after its own step, so its outcome can be read off now, for all lanes at once:
                    // the same decision from the same number, without per-step bookkeeping. Moves are listed in lane (= marker) order.
                    {
                        int cls;
                        double gn;
                        decide(crhs, cls, gn);
                        const bool sel = lv && (cgold != 0.0 || crhs * crhs >= cthr[0]);
                        const int rc = sel ? cls : 0;
                        const double rg = sel ? gn : 0.0;
                        const double dmine = rg - cgold;
                        const unsigned long long moved = __ballot(lv && dmine != 0.0);
                        if (lv && dmine != 0.0) {
                            const int pos = nev0 + __popcll(moved & ((1ull << lane) - 1ull));
                            ev_ix[pos] = (cslot << 16) | ct;
                            ev_del[pos] = dmine;
                        }
                        missacc += __popcll(moved & noslot);
                        res_c[lane] = rc;
                        res_g[lane] = rg;
                        if (lane == 0) cnts[0] = nev0 + __popcll(moved);
                        if (HB_APPLY_LEAN && crowded) { // (the same moves once more, as the other waves' apply wants them)
                            const unsigned long long mvs = moved & ~noslot, mvm = moved & noslot, below = (1ull << lane) - 1ull;
                            const long long db = __double_as_longlong(dmine);
                            if (lv && dmine != 0.0) {
                                if (cslot >= 0) ap_rec[__popcll(mvs & below)] = make_int4(cslot << 8, ct, (int)db, (int)(db >> 32));
                                else ms_rec[__popcll(mvm & below)] = make_int4(0, ct, (int)db, (int)(db >> 32));
                            }
                            const int nsr = __popcll(mvs);
                            if (lane < 8) ap_rec[nsr + lane] = make_int4(0, 0x7fffffff, 0, 0);
                            if (lane == 0) {
                                cnts[3] = __popcll(mvm);
                                __hip_atomic_store(&cnts[2], nsr, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(&cnts[4], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); // the list is complete
                            }
                        }
                    }
                }
                // ---- the apply of a crowded round, WHILE the serial pass runs (HB_APPLY_PROG): the other waves take the records of every
                // block the pass has finished (the count is released after them) instead of standing at the barrier below until the
                // whole pass is through; wave 0 does its own non-candidates afterwards. Cached rows in marker order, whatever the
                // blocks' timing (a block's records are appended in marker order and applied in list order), the others after the
                // pass — the same sums bit for bit as the apply behind the barrier.
                double acc_prog = rhs;
                if (HB_APPLY_LEAN && HB_APPLY_PROG && crowded) {
                    const bool doap0 = undec && !inr;
                    const bool anyap = __any(doap0);
                    int lo = 0;
                    // (the wave that shares wave 0's SIMD — four SIMDs, waves dealt round-robin — stays asleep until the pass is through:
                    // every instruction it issues is an issue slot the serial pass does not get: 7 800 -> 13 000 cycles measured)
                    if (S == 8 && wave == 4)
                        while (!__hip_atomic_load(&cnts[4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) __builtin_amdgcn_s_sleep(8);
                    for (;;) {
                        const int fin = __hip_atomic_load(&cnts[4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const int hi = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cnts[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                        if (anyap && hi > lo) {
                            const int kl = lo + lane < hi ? ap_rec[lo + lane].y : 0x7fffffff; // (a round has at most 64 moves)
                            const int n_in = __popcll(__ballot(kl < (t | 63)));
                            const int n_un = __popcll(__ballot(kl < (t & ~63))) & ~7;
                            for (int e0 = lo; e0 < lo + n_un; e0 += 8) { // moves of markers before the wave's first: no select
                                int4 rc[8];
                                int gv[8];
#pragma unroll
                                for (int q = 0; q < 8; q++) rc[q] = ap_rec[e0 + q];
#pragma unroll
                                for (int q = 0; q < 8; q++) gv[q] = reinterpret_cast<const int *>(smem + rc[q].x)[t];
#pragma unroll
                                for (int q = 0; q < 8; q++)
                                    acc_prog = fma(-(double)gv[q], __longlong_as_double(((long long)rc[q].w << 32) | (unsigned)rc[q].z), acc_prog);
                            }
                            for (int e0 = lo + n_un; e0 < lo + n_in; e0 += 8) { // the wave's own stretch (records past `hi` may be half written: never used)
                                int4 rc[8];
                                int gv[8];
#pragma unroll
                                for (int q = 0; q < 8; q++) rc[q] = ap_rec[e0 + q];
#pragma unroll
                                for (int q = 0; q < 8; q++) gv[q] = reinterpret_cast<const int *>(smem + (rc[q].x & 0x3fffc))[t];
#pragma unroll
                                for (int q = 0; q < 8; q++) {
                                    const double nw = fma(-(double)gv[q], __longlong_as_double(((long long)rc[q].w << 32) | (unsigned)rc[q].z), acc_prog);
                                    acc_prog = (e0 + q < hi && rc[q].y < t) ? nw : acc_prog;
                                }
                            }
                        }
                        lo = max(lo, hi);
                        if (fin) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    const int nmr = cnts[3];
                    if (anyap) {
                        for (int e0 = 0; e0 < nmr; e0 += 8) { // moves whose row is not in the cache
                            int4 rc[8];
                            int gv[8];
#pragma unroll
                            for (int q = 0; q < 8; q++) rc[q] = ms_rec[min(e0 + q, nmr - 1)];
#pragma unroll
                            for (int q = 0; q < 8; q++) gv[q] = gp[(size_t)__builtin_amdgcn_readfirstlane(rc[q].y) * P + t];
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const double nw = fma(-(double)gv[q], __longlong_as_double(((long long)rc[q].w << 32) | (unsigned)rc[q].z), acc_prog);
                                acc_prog = (e0 + q < nmr && rc[q].y < t) ? nw : acc_prog;
                            }
                        }
                    }
                }
                if (t_lo == 0 && nev0 == 0 && !forced) HB_STAMP(13);
                __syncthreads();
                const int nev1 = cnts[0];
                // everybody still undecided applies the round's moves (those of earlier markers) to its own rhs
                double rhs_new = rhs;
                // (the round's moves are listed in marker order, and a move only touches later markers: a wave needs the moves of
                // the markers before its last one — a prefix of the list, on average half of it)
                int nap = nev1;
#if HB_APPLY_PREFIX
                {
                    const int nr = nev1 - nev0; // <= 64: one lane per move
                    const int kl = lane < nr ? (ev_ix[nev0 + lane] & 0xffff) : 0x7fffffff;
                    nap = nev0 + __popcll(__ballot(kl < ((t | 63))));
                }
#endif
                const bool doap = undec && !inr;
                if (HB_APPLY_LEAN && HB_APPLY_PROG && crowded) {
                    if (doap) rhs_new = acc_prog; // (applied before the barrier, while the serial pass ran)
                } else if (HB_APPLY_LEAN && crowded) {
                    // A crowded round (BayesR: ~50 moves): the apply used to be the longest phase of the panel — eight waves, two
                    // per SIMD, each issuing ~15 instructions per move (the move's record handed round by v_readlane, a scalar row
                    // address, the test for a row outside the cache, the select for "this marker comes later") at ~13 cycles an
                    // instruction: 10 600 cycles of 52 000 (profiles/r04_bayesr_chain_phases.txt). Here the serial pass leaves
                    // the round's moves as 16-byte records {row's byte offset in the cache, marker, change} that every lane reads
                    // with ONE broadcast LDS load; a move of a marker before the wave's first needs no select at all, so a move
                    // costs five instructions (record, address, Gram entry, conversion, fused multiply-add). The few moves whose
                    // row is not cached come afterwards, their global loads in flight together. (The moves are summed in a
                    // different order than the per-panel kernel sums them: the same chain up to the rounding of rhs, which
                    // every comparison in tests/ already allows for.)
                    const int nsr = cnts[2], nmr = cnts[3];
                    if (__any(doap)) {
                        const int kl = lane < nsr ? ap_rec[lane].y : 0x7fffffff;
                        const int nap_s = __popcll(__ballot(kl < (t | 63)));
                        const int nfull = __popcll(__ballot(kl < (t & ~63))) & ~7;
                        double acc = rhs;
                        for (int e0 = 0; e0 < nfull; e0 += 8) {
                            int4 rc[8];
                            int gv[8];
#pragma unroll
                            for (int q = 0; q < 8; q++) rc[q] = ap_rec[e0 + q];
#pragma unroll
                            for (int q = 0; q < 8; q++) gv[q] = reinterpret_cast<const int *>(smem + rc[q].x)[t];
#pragma unroll
                            for (int q = 0; q < 8; q++)
                                acc = fma(-(double)gv[q], __longlong_as_double(((long long)rc[q].w << 32) | (unsigned)rc[q].z), acc);
                        }
                        for (int e0 = nfull; e0 < nap_s; e0 += 8) { // (the wave's own stretch of the panel; the list is padded with eight changes of zero)
                            int4 rc[8];
                            int gv[8];
#pragma unroll
                            for (int q = 0; q < 8; q++) rc[q] = ap_rec[e0 + q];
#pragma unroll
                            for (int q = 0; q < 8; q++) gv[q] = reinterpret_cast<const int *>(smem + rc[q].x)[t];
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const double nw = fma(-(double)gv[q], __longlong_as_double(((long long)rc[q].w << 32) | (unsigned)rc[q].z), acc);
                                acc = rc[q].y < t ? nw : acc;
                            }
                        }
                        for (int e0 = 0; e0 < nmr; e0 += 8) { // moves whose row is not in the cache
                            int4 rc[8];
                            int gv[8];
#pragma unroll
                            for (int q = 0; q < 8; q++) rc[q] = ms_rec[min(e0 + q, nmr - 1)];
#pragma unroll
                            for (int q = 0; q < 8; q++) gv[q] = gp[(size_t)__builtin_amdgcn_readfirstlane(rc[q].y) * P + t];
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const double nw = fma(-(double)gv[q], __longlong_as_double(((long long)rc[q].w << 32) | (unsigned)rc[q].z), acc);
                                acc = (e0 + q < nmr && rc[q].y < t) ? nw : acc;
                            }
                        }
                        if (doap) rhs_new = acc;
                    }
                } else if (doap) {
                    for (int e0 = nev0; e0 < nap; e0 += 8) {
                        int rec[8], gv[8];
                        double dl[8];
#pragma unroll
                        for (int q8 = 0; q8 < 8; q8++) {
                            const int e = min(e0 + q8, nap - 1);
                            rec[q8] = ev_ix[e];
                            dl[q8] = ev_del[e];
                        }
#pragma unroll
                        for (int q8 = 0; q8 < 8; q8++) {
                            const int slot = __builtin_amdgcn_readfirstlane(rec[q8] >> 16);
                            const int k = __builtin_amdgcn_readfirstlane(rec[q8] & 0xffff);
                            gv[q8] = rowc[(max(slot, 0) << 6) + t];
                            if (slot < 0) gv[q8] = gp[(size_t)k * P + t];
                        }
#pragma unroll
                        for (int q8 = 0; q8 < 8; q8++) {
                            const bool ap = e0 + q8 < nap && (rec[q8] & 0xffff) < t;
                            rhs_new = ap ? fma(-(double)gv[q8], dl[q8], rhs_new) : rhs_new;
                        }
                    }
                }
                if (t_lo == 0 && nev0 == 0 && !forced) HB_STAMP(14);
                const bool viol = undec && !inr && t < t_hi && active && rhs_new * rhs_new >= (have_exact ? thr[0] : thr_lo);
                const unsigned long long vm = __ballot(viol);
                if (lane == 0) wviol[wave] = vm != 0ull;
                __syncthreads();
                bool anyv = false;
                {
                    int w8[8];
                    hb_read8(wviol, w8);
#pragma unroll
                    for (int w = 0; w < 8; w++) anyv |= w8[w] != 0;
                }
                if (t_lo == 0 && nev0 == 0 && !forced) HB_STAMP(19);
                if (anyv) { // roll the round back; the markers that crossed their threshold join the candidates
                    forced |= viol;
                    if (t == 0) cnts[0] = nev0;
                    if (t == 0) redoacc++;
                    continue;
                }
                rhs = rhs_new;
                if (inr) { cls_f = res_c[rank]; g_f = res_g[rank]; }
                nev0 = nev1;
                t_lo = t_hi;
                if (t_lo >= P) break;
            }
            nev = cnts[0];
            }
            // Every load of this panel is consumed HERE on every path, as far as hipcc can see: the exact data and the band rows
            // are used under conditions (a staged candidate, the prefetched mover), and a load hipcc still counts as possibly
            // outstanding at the loop's back edge makes it guard the next panel's first reuse of those registers with
            // s_waitcnt vmcnt(0) — which also drains the DMA pieces issued at the end of this panel (it cannot see them): every
            // panel, quiet ones included, then waited out a full memory round trip right after its opening barrier.
            asm volatile("" ::"v"(gold), "v"(xx), "v"(myslot));
#pragma unroll
            for (int c = 0; c < K1; c++) asm volatile("" ::"v"(thr[c]), "v"(invv[c]), "v"(sdz[c]));
#pragma unroll
            for (int l = 0; l < (NPL > 0 ? NPL : 1); l++) {
                if (NPL > 0) asm volatile("" ::"v"(pre[0][l]));
                if (NPL > 0 && HB_NPF > 1) asm volatile("" ::"v"(pre[1][l]));
            }
        }
        HB_STAMP(2);
        HB_STAMP_VAL(10, nev);
#if HB_STAMPS
        HB_STAMP_VAL(16, nround);
        HB_STAMP_VAL(17, nrerun);
#endif
        // ---- with k_fwd beside the chain: what the NEXT panel's take needs from other workgroups is requested here, a results-and-
        // fold's length ahead of that take, instead of as the last thing of the panel (a round trip the take then waited out) and
        // three panels ahead (its dots: the ring's copy predates the launch that finalizes them at every panel, and the take's
        // re-read was a second round trip). One 1-KiB piece of each per ring wave (P = 512); a word not written yet shows the
        // sentinel the sweep filled dsum[] / fcorr[] with and is polled at the take as before. The pieces are older than anything
        // the rest of the panel issues, so the counted wait at the top of the next panel covers them.
        if (HB_R_EARLY && fwd && wave < RW && have_next) {
            const unsigned wo = (unsigned)__builtin_amdgcn_readfirstlane(wave) << 10;
            const int nslot_o = (oslot + 1 == HB_RD) ? 0 : oslot + 1;
            dma_piece_s(reinterpret_cast<const char *>(v.dsum + (size_t)(p + 1) * P) + wo, oring_lds + (unsigned)nslot_o * OSLOT + wo, true);
            if (p + 1 >= pv.p0 + 2)
                dma_piece_s(reinterpret_cast<const char *>(pv.fcorr + (size_t)(p + 1) * P) + wo,
                            (unsigned)(uintptr_t)fcring + (unsigned)(((p + 1) & 1) * P * 8) + wo, true);
        }
        // (... and the band rows the panel's first 32 moves fold into the next panel: the loads fly while the moves are published and the
        // results written, instead of starting after them)
        int fgv[HB_FPRE_N];
        const bool fpre = HB_R_FOLDPRE && K1 <= 3 && fwd && nev > 0 && have_next; // (K1 = 7 has no registers to spare)
        int fixl[HB_FPRE_N / 64 + 1];
        if (fpre) {
            const int32_t *blk1 = v.gram + ((size_t)(p + 1) * (pv.Lg + 1) + 1) * ((size_t)P * P) + t;
#pragma unroll
            for (int h = 0; h < (HB_FPRE_N + 63) / 64; h++) fixl[h] = (h * 64 + lane < nev) ? (ev_ix[h * 64 + lane] & 0xffff) : 0;
#pragma unroll
            for (int f0 = 0; f0 < HB_FPRE_N; f0 += 16) {
                if (f0 == 0 || nev > f0) { // (uniform: sixteen rows at a time, as many as the panel has moves)
#pragma unroll
                    for (int f = f0; f < f0 + 16; f++) fgv[f] = blk1[(size_t)__builtin_amdgcn_readlane(fixl[f >> 6], f & 63) * P];
                }
            }
        }
        HB_STAMP(3);
        if (tot0 > 0) {
            // ---- publish the panel's moves (the update of this group waits for them). Only the last wave does it,
            // from the LDS lists: write-through stores now; the drain + chain_done flag at the next panel's take, so
            // that no wave of the chain ever waits for a store to reach memory. A quiet panel keeps the zero count
            // the sweep started with. ----
            if (wave == S - 1 && nev > 0) {
                double absd = 0.0;
                for (int e = lane; e < nev; e += 64) {
                    st_sc1(&v.ev_idx[(size_t)p * P + e], ev_ix[e] & 0xffff);
                    st_sc1(&v.ev_delta[(size_t)p * P + e], ev_del[e]);
                    absd += fabs(ev_del[e]);
                }
                if (lane == 0) st_sc1(&v.ev_count[(size_t)p * HB_EVS], nev);
                if (v.mb) {
                    mbr = fma(v.xabs, wave_sum(absd), mbr);
                    // (the group's bound goes out WITH its last panel's moves — the update rows poll it — not after the results and the
                    // forward fold at the panel's end: ~11 000 cycles earlier, profiles/r04_bayesr_chain_phases.txt)
                    if (group_end && lane == 0) st_sc1(&v.mb[(size_t)(1 + gcount) * HB_MBS], mbr);
                }
            }
            HB_STAMP(4);
            if (!active) { cls_f = 0; g_f = 0.0; }
            if (g_f != gold) v.g[j] = g_f;
            if (hot || cls_f != 0) v.tracker[j] = (uint8_t)cls_f; // a marker at zero that stays there keeps its 0
            if (count_pip && cls_f != 0) {
                __hip_atomic_fetch_add(&v.nzrate[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // no return value: nothing to wait for
                if (v.wind) v.wflag[v.wind[j] - 1u] = 1;
            }
            if (store && g_f != 0.0) {
                // one writer per marker: an atomic add gives the same sum as load-add-store, without the load's round trip
                unsafeAtomicAdd(&v.alpha_sum[j], g_f);
                unsafeAtomicAdd(&v.alpha_sq[j], g_f * g_f);
            }
            if (cls_f > 0) wacc += (model == 6) ? g_f * g_f / pin->fold[cls_f] : g_f * g_f;
#pragma unroll
            for (int c = 0; c <= K1; c++) cacc[c] += (active && cls_f == c) ? 1 : 0;
            evacc = nev + evacc;
            HB_STAMP(5);
            // ---- fold the moves forward into the corrections of the next Lb panels ----
            const int lcount = min(min(fwd ? 1 : pv.Lb, (pv.Lv + 1) * pv.D - 1 - pmodD), np - 1 - p); // panels that need the correction FROM HERE (k_fwd: the others)
            bool from_pre = NPL > 0 && nev > 0 && nev <= HB_NPF;
            int w0 = 0, w1 = 0;
            if (from_pre) { // did exactly (a subset of) the first two candidates move? Their rows are already here
                const int e0 = ev_ix[0] & 0xffff, e1 = ev_ix[nev - 1] & 0xffff;
                w0 = e0 == c1 ? 0 : (HB_NPF > 1 && e0 == c2 ? 1 : -1);
                w1 = e1 == c1 ? 0 : (HB_NPF > 1 && e1 == c2 ? 1 : -1);
                from_pre = w0 >= 0 && w1 >= 0;
            }
            if (fpre) { // (k_fwd beside the chain: the next panel only; the rows were requested before the publish; the same fused multiply-adds in the same order as fold_forward's)
                const int slot = (pslot + 1 == R) ? 0 : pslot + 1;
                double *cp = corrL + (size_t)slot * P + t;
                double acc = *cp;
                double dll[HB_FPRE_N / 64 + 1];
#pragma unroll
                for (int h = 0; h < (HB_FPRE_N + 63) / 64; h++) dll[h] = (h * 64 + lane < nev) ? ev_del[h * 64 + lane] : 0.0;
#pragma unroll
                for (int f0 = 0; f0 < HB_FPRE_N; f0 += 16) {
                    if (f0 == 0 || nev > f0) {
#pragma unroll
                        for (int f = f0; f < f0 + 16; f++) acc = fma((double)fgv[f], readlane_f64(dll[f >> 6], f & 63), acc);
                    }
                }
                *cp = acc;
                if (nev > HB_FPRE_N) fold_forward<1, 32>(corrL, R, v.gram, pv.Lg, lcount, pslot, P, t, nev - HB_FPRE_N, ev_ix + HB_FPRE_N, ev_del + HB_FPRE_N, p);
            } else if (from_pre) {
                const double d0 = ev_del[0], d1 = nev > 1 ? ev_del[1] : 0.0;
                int slot = pslot;
#pragma unroll
                for (int l = 1; l <= (NPL > 0 ? NPL : 1); l++) {
                    slot = (slot + 1 == R) ? 0 : slot + 1;
                    if (l <= lcount) { // (lcount <= Lb = NPL here; the same fused multiply-adds, in event order, as fold_forward's)
                        double *cp = corrL + (size_t)slot * P + t;
                        double acc = *cp;
                        acc = fma((double)(HB_NPF > 1 && w0 ? pre[1][l - 1] : pre[0][l - 1]), d0, acc);
                        if (HB_NPF > 1 && nev > 1) acc = fma((double)(w1 ? pre[1][l - 1] : pre[0][l - 1]), d1, acc);
                        *cp = acc;
                    }
                }
            } else if (nev > 0) { // batch shape by band width: as many loads in flight as the registers allow
                // (a kernel specialised for one band width — NPL == Lb — carries only that width's fold: the others would
                // be dead code that still costs registers in the loop every panel runs)
                if (fwd) fold_forward<1, 32>(corrL, R, v.gram, pv.Lg, lcount, pslot, P, t, nev, ev_ix, ev_del, p); // (the next panel only: 32 moves per trip)
                else if (NPL > 12) fold_forward<HB_LBMAX, 2>(corrL, R, v.gram, pv.Lg, lcount, pslot, P, t, nev, ev_ix, ev_del, p);
                else if (NPL > 0 && NPL <= 2) fold_forward<2, 16>(corrL, R, v.gram, pv.Lg, lcount, pslot, P, t, nev, ev_ix, ev_del, p);
                else if (pv.Lb <= 2) fold_forward<2, 16>(corrL, R, v.gram, pv.Lg, lcount, pslot, P, t, nev, ev_ix, ev_del, p);
                else if (pv.Lb <= 5) fold_forward<5, 8>(corrL, R, v.gram, pv.Lg, lcount, pslot, P, t, nev, ev_ix, ev_del, p);
                else if (pv.Lb <= 12) fold_forward<12, 2>(corrL, R, v.gram, pv.Lg, lcount, pslot, P, t, nev, ev_ix, ev_del, p);
                else fold_forward<HB_LBMAX, 2>(corrL, R, v.gram, pv.Lg, lcount, pslot, P, t, nev, ev_ix, ev_del, p);
            }
        } else {
            cacc[0] += active ? 1 : 0; // a quiet panel: nothing moved, nothing to write
        }
        if (wave == S - 1 && lane == 0 && nev == 0) { // (the update rows poll the count itself: a panel without moves says so — and the bound, unchanged)
            st_sc1(&v.ev_count[(size_t)p * HB_EVS], 0);
            if (group_end && v.mb) st_sc1(&v.mb[(size_t)(1 + gcount) * HB_MBS], mbr);
        }
        HB_STAMP(8);
        if (wave == S - 1 && group_end) {
            // last panel of its mat-vec group: the update of this group is waiting for exactly these moves, and the
            // next panel's take may itself have to wait for a later launch — publish now rather than at that take
            gcount++; // (its bound went out with the moves of its last panel)
            // (no drain before the flag any more, as in k_chain_group: every consumer of the counts, the bound and the move lists
            // validates the words themselves, and chain_done only paces k_fwd and k_warm. Waiting here for the acknowledgement of
            // this wave's write-through stores held the whole workgroup at the next panel's first barrier for ~9 300 cycles —
            // a fifth of a BayesR panel, profiles/r04_bayesr_chain_phases.txt)
            if (lane == 0) st_flag(pv.flags + HB_FLAG_CHAIN_DONE, (unsigned)(p + 1));
        }
        HB_STAMP(9);
        // ---- requests for the panels ahead, as the LAST thing of the panel: hipcc's own waits count only the loads it knows,
        // so any of them placed after a DMA piece would drain that piece as well (the queue is in-order); issued here, the
        // pieces have the whole next panel — which, when quiet, contains no vector-memory wait at all — to land ----
        // (0) ring group of panel p + HB_RD - 1, into the slot panel p - 1 has just left
        // (k_fwd's sums for the NEXT panel first — one 1-KiB piece per ring wave, P = 512 — so that the counted wait at the top of the
        // next panel, which lets the youngest ring group stay in flight, covers them; a word k_fwd has not written yet shows the
        // sentinel the sweep filled fcorr[] with and is polled at the take)
        if (!HB_R_EARLY && fwd && wave < RW && have_next && p + 1 >= pv.p0 + 2)
            dma_piece_s(reinterpret_cast<const char *>(pv.fcorr + (size_t)(p + 1) * P) + (__builtin_amdgcn_readfirstlane(wave) << 10),
                        (unsigned)(uintptr_t)fcring + (unsigned)(((p + 1) & 1) * P * 8) + ((unsigned)__builtin_amdgcn_readfirstlane(wave) << 10), true);
        if (wave < RW && p + HB_RD - 1 < np) issue_group(p + HB_RD - 1, (oslot + HB_RD - 1) % HB_RD);
        // (1) Gram rows of the next panel's hot markers, straight into the other half of the LDS row cache by LDS-DMA; the
        // first reader of that half — the first round of the next panel that has candidates — drains vmcnt before its
        // barrier. A quiet panel never waits for them. (Wave 0 is left out when there are other waves: its memory queue
        // then holds ring groups only, which is what makes its counted wait at the top of the panel exact.)
        const int *hpk = reinterpret_cast<const int *>(oslotp + OSZ); // packed list of panel p + 1 (came with group p)
        if (have_next) n_nhot = hpk[0];
        const bool tri = HB_ROW_TRI && P == 512;
        const int n2s = tri ? __builtin_amdgcn_readfirstlane(hpk[2]) : 0, shp = tri ? __builtin_amdgcn_readfirstlane(hpk[3]) : 0;
        const int n_total = n_nhot << lgP, n_items = tri ? n_nhot + n2s : (n_total + 255) >> 8;
        // (HB_FILL_ALL, panels of 256 and more: every wave issues its share — the four non-ring waves alone took ~8 000 cycles over
        // the ~80 pieces of a BayesR panel while the ring waves stood at the next panel's barrier; a ring wave's pieces go out behind
        // its ring group and its counted wait at the top of the next panel leaves them in flight too)
        const bool fill_all = HB_FILL_ALL && P >= 256 && S > 1;
        my_rowp = 0;
        if (have_next && (S == 1 || wave >= RW || fill_all)) {
            const unsigned rown_lds = (unsigned)(uintptr_t)rown;
            const int w0 = S == 1 ? 0 : fill_all ? wave : wave - RW, ws = S == 1 ? 1 : fill_all ? S : S - RW;
            if (P >= 256) { // a piece is (part of) ONE row: scalar base, invariant lane offset
                const int w0u = __builtin_amdgcn_readfirstlane(w0);
                const int lg = lgP - 8; // pieces per row = P / 256
                // (the list's markers in two registers, handed out with v_readlane: an LDS read and its wait per piece made this loop —
                // ~20 pieces per wave, on the path to the next panel's opening barrier — several thousand cycles long)
                const int ids0 = hpk[4 + lane], ids1 = hpk[4 + 64 + lane];
                const int n_it = __builtin_amdgcn_readfirstlane(n_items);
                const unsigned long long gpn_s = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(uintptr_t)gpn >> 32)) << 32) |
                                                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)(uintptr_t)gpn);
                unsigned keep_m0;
                asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0)); // (M0 — the LDS destination — is compiler-reserved: saved once around the loop, set by every piece)
                for (int it = w0u; it < n_it; it += ws) {
                    // (panel 512: whole rows first, then the second pieces of the rows of the panel's second half — k_hotlist)
                    const int r = tri ? (it < 2 * n2s ? it >> 1 : it - n2s) : it >> lg;
                    const int pc = tri ? (it < 2 * n2s ? (it & 1) << 8 : 256) : (it & ((1 << lg) - 1)) << 8; // first column of the piece
                    const int kk = r < 64 ? __builtin_amdgcn_readlane(ids0, r) : __builtin_amdgcn_readlane(ids1, r - 64);
                    const unsigned long long src = gpn_s + ((((unsigned long long)(unsigned)kk << lgP) + (unsigned)pc) << 2);
                    const unsigned dst = rown_lds + ((unsigned)(it + shp) << 10);
                    my_rowp++;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane16), "s"(src), "s"(dst) : "memory");
                }
                asm volatile("s_mov_b32 m0, %0" : : "s"(keep_m0));
            } else
            for (int it = w0; it < n_items; it += ws) {
                const int lin = (it << 8) + lane * 4;
                if (lin < n_total) {
                    const int32_t *src = gpn + ((size_t)hpk[4 + (lin >> lgP)] << lgP) + (lin & (P - 1));
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(src), "s"(__builtin_amdgcn_readfirstlane(rown_lds + ((unsigned)it << 10)))
                                 : "memory");
                }
            }
        }
        HB_STAMP(6);
        // no closing barrier: the next panel's opening barrier separates every reuse of the LDS lists, the candidate
        // staging and the row-cache halves; what is written before it (wcnt, hl, s_nh) alternates by panel parity
    }

    // ---- the last panel's moves: drain and publish ----
    if (wave == S - 1 && ok) {
        if (lane == 0 && v.mb) st_sc1(&v.mb[(size_t)(1 + gcount) * HB_MBS], mbr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) st_flag(pv.flags + HB_FLAG_CHAIN_DONE, (unsigned)np);
    }
    // ---- sweep totals for the hyper-parameter draws ----
    __syncthreads();
    const double wsum = block_sum(wacc, red);
    // (+=: a sweep may come in several ranges, hb_ctx_sweep_range; the sweep's first range starts from zeroed sums)
    if (t == 0) {
        v.acc[HB_ACC_SUMG2] += wsum;
        v.acc[HB_ACC_EVENTS] += (double)evacc;
    }
    {
        const double ms = block_sum((double)(lane == 0 ? missacc : 0), red);
        if (t == 0) v.acc[HB_ACC_MISS] += ms;
        if (t == 0) v.acc[HB_ACC_REDO] += (double)redoacc;
    }
#pragma unroll
    for (int c = 0; c <= K1; c++) {
        const double cs = block_sum((double)cacc[c], red);
        if (t == 0 && c < HB_MAX_FOLD) v.acc[HB_ACC_COUNT0 + c] += cs;
    }
    if (t == 0 && !ok) { // aborted: the host must see it (fetch_acc checks the flag), then release every waiter
        st_flag(pv.flags + HB_FLAG_ABORT, 1u);
        st_flag(pv.flags + HB_FLAG_CHAIN_DONE, 0x7fffffffu);
    }
}

#include "hb_chain_group.hpp"
#include "hb_chain_dense.hpp"

// k_gate: one lane on the mat-vec stream, ahead of the sweep's first launch, that waits until the chain workgroup is resident
// (it publishes HB_FLAG_XCC as its first act). The chain needs a compute unit with ALL of its LDS free; it is launched first,
// but the graph's branches start together, and once mat-vec blocks have touched every compute unit it only gets one when a
// compute unit drains completely — which never happens where a launch's update blocks, one per 64 rows, sit on every compute
// unit waiting for the chain (measured: the sweep then times out; with fewer update blocks than compute units the late start
// went unnoticed). While this lane waits the chip is empty, so the chain starts at once.
__global__ void k_gate(unsigned *flags)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (ld_flag(flags + HB_FLAG_XCC) == 0u) {
        if (ld_flag(flags + HB_FLAG_ABORT)) return;
        if (wall_clock64() - t0 > HB_TIMEOUT_TICKS) { st_flag(flags + HB_FLAG_ABORT, 1u); return; }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---------------------------------------------------------------------------------------------
// k_warm: the chain workgroup's memory traffic, pulled into ITS L2 ahead of time by other compute units.
// One compute unit gets ~18 bytes per clock out of HBM however many loads it keeps in flight (its miss queue is the limit;
// tools/rowfetch_bench.hip: 113 cycles per 2-KiB row), but 64 bytes per clock out of its XCD's L2 (31 cycles per row). What
// the chain will read is known a sweep ahead for every marker on a panel's hot list (k_hotlist: the markers in the model —
// certain to move — and the likely entries): the Gram row that goes into the row cache and the band rows its move folds
// forward. The workgroups of this kernel that landed on the chain's XCD (workgroups are dealt round-robin over the 8 XCDs;
// the chain publishes its own) read exactly those rows, `ahead` panels in front of the chain's published progress, and
// throw the data away. It is a hint: nothing waits for it, nothing depends on it, a late or missing row is only slower.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_warm(persist_view pv, chain_view v, int K1, const int32_t *__restrict__ gram, int P, int ahead,
                                              int per_xcd, int *__restrict__ sink)
{
    __shared__ int s_rank;
    const int t = threadIdx.x;
    if (t == 0) {
        unsigned my;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
        my &= 15u;
        unsigned want = 0;
        const unsigned long long t0 = wall_clock64();
        while ((want = ld_flag(pv.flags + HB_FLAG_XCC)) == 0u) {
            if (ld_flag(pv.flags + HB_FLAG_ABORT) || ld_flag(pv.flags + HB_FLAG_CHAIN_DONE) >= (unsigned)pv.npanels ||
                wall_clock64() - t0 > 100000000ull) break; // (1 s: the chain never started)
            __builtin_amdgcn_s_sleep(16);
        }
        s_rank = (want == my + 1u) ? (int)(blockIdx.x >> 3) % per_xcd : -1;
    }
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) return;
    const int np = pv.npanels, Lb = pv.Lb, Lg = pv.Lg;
    const size_t PP = (size_t)P * P, step = (size_t)(Lg + 2) * PP;
    const int quarter = P >> 2;            // int4 lanes per row
    const int rows_per_pass = 256 / quarter; // rows one instruction of this workgroup covers
    int acc = 0;
    for (int q = pv.p0; q < np; q++) {
        // pace: at most `ahead` panels in front of the chain's published progress
        unsigned done;
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            done = ld_flag(pv.flags + HB_FLAG_CHAIN_DONE);
            if ((int)done + ahead >= q || ld_flag(pv.flags + HB_FLAG_ABORT) || wall_clock64() - t0 > HB_TIMEOUT_TICKS) break;
            __builtin_amdgcn_s_sleep(32);
        }
        if ((int)done >= np || ld_flag(pv.flags + HB_FLAG_ABORT) || (int)done + ahead < q) break;
        if (q < (int)done) continue; // the chain is already past this panel
        if (rank == (q % per_xcd)) {
            // the panel's exact per-marker data (what its candidates fetch at the opening): 8 P bytes per array
            const size_t j0 = (size_t)q * P;
            for (int i = t * 2; i < P; i += 512) { // 16 bytes per lane
                const double2 a = *reinterpret_cast<const double2 *>(v.g + j0 + i), b = *reinterpret_cast<const double2 *>(v.xpx + j0 + i);
                acc += (int)(a.x + a.y + b.x + b.y);
                for (int c = 0; c < K1; c++) {
                    const double2 x = *reinterpret_cast<const double2 *>(v.thr + (size_t)c * v.m_pad + j0 + i);
                    const double2 y = *reinterpret_cast<const double2 *>(v.invv + (size_t)c * v.m_pad + j0 + i);
                    const double2 z = *reinterpret_cast<const double2 *>(v.sdz + (size_t)c * v.m_pad + j0 + i);
                    acc += (int)(x.x + y.y + z.x);
                }
            }
            for (int i = t * 4; i < P; i += 1024) acc += reinterpret_cast<const int4 *>(pv.slot_of + j0 + i)->x;
        }
        const int *hl = pv.hotpack + (size_t)q * HB_HS;
        const int cnt = max(hl[0], hl[1]); // (with and without a slot in the row cache)
        const int lmax = min(Lb, np - 1 - q);
        const int nitem = cnt * (1 + lmax);
        const int32_t *gp = gram + (size_t)q * (Lg + 1) * PP;
        const int32_t *fwd = gram + ((size_t)(q + 1) * (Lg + 1) + 1) * PP;
        const int sub = t / quarter, col = (t - sub * quarter) * 4;
        for (int it = rank * rows_per_pass + sub; it < nitem; it += per_xcd * rows_per_pass) {
            const int mi = it / (1 + lmax), l = it - mi * (1 + lmax);
            const int k = hl[4 + mi];
            const int32_t *src = (l == 0 ? gp : fwd + (size_t)(l - 1) * step) + (size_t)k * P + col;
            const int4 x = *reinterpret_cast<const int4 *>(src);
            acc += x.x ^ x.y ^ x.z ^ x.w;
        }
    }
    if (acc == 0x5a5a5a5a) sink[0] = acc; // (keeps the loads)
}

// ---------------------------------------------------------------------------------------------
// k_update: yadj -= sum_e x_e D_e, u += the same, r32 = (float)yadj, for the panel's changed markers.
// thread = 4 consecutive rows; the event list is staged in LDS once, then the column loads of 8
// events are in flight together.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_update_dense(int64_t ld, upd_view q)
{
    __shared__ __attribute__((aligned(16))) char smem[HBU_LDS];
    update_rows_dense(ld, q, blockIdx.x, gridDim.x, smem);
}

__global__ __launch_bounds__(256) void k_update(int64_t ld, upd_view q)
{
    __shared__ int s_ix[512];
    __shared__ double s_dl[512];
    __shared__ int s_ok[2];
    update_rows(ld, q, blockIdx.x, s_ix, s_dl, s_ok);
}

// ---------------------------------------------------------------------------------------------
// end-of-sweep reductions behind src/Bayes.cpp:819 (var(u), N-1, two-pass like arma::var) and
// :823 (yadj.yadj); also sum(yadj) for the next intercept draw (:480). One workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_reduce_ru(const double *__restrict__ r, const double *__restrict__ u,
                                                    int n, double *__restrict__ acc)
{
    __shared__ double red[16];
    double sr = 0, sr2 = 0, su = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double a = r[i];
        sr += a;
        sr2 = fma(a, a, sr2);
        su += u[i];
    }
    sr = block_sum(sr, red);
    sr2 = block_sum(sr2, red);
    su = block_sum(su, red);
    const double mean = su / n;
    double a2 = 0, a3 = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double d = mean - u[i];
        a2 = fma(d, d, a2);
        a3 += d;
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

// ---------------------------------------------------------------------------------------------
// helpers for the host blocks sharing yadj (reference src/Bayes.cpp:479-516)
// ---------------------------------------------------------------------------------------------
__global__ void k_shift(double *__restrict__ r, float *__restrict__ r32, int n, double a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = r[i] + a;
    r[i] = v;
    r32[i] = (float)v;
}

__global__ void k_axpy(double *__restrict__ r, float *__restrict__ r32, const double *__restrict__ x, int n, double a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = fma(a, x[i], r[i]);
    r[i] = v;
    r32[i] = (float)v;
}

__global__ __launch_bounds__(1024) void k_dot_vec(const double *__restrict__ x, const double *__restrict__ y, int n,
                                                  double *__restrict__ out)
{
    __shared__ double red[16];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s = fma(x[i], y[i], s);
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s;
}

// Z_t' yadj: per-level sums; one workgroup, LDS-free atomics on a zeroed buffer
__global__ void k_level_sums(const double *__restrict__ r, const int32_t *__restrict__ zid, int n,
                             double *__restrict__ sums)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAdd(&sums[zid[i]], r[i]);
}

__global__ void k_level_axpy(double *__restrict__ r, float *__restrict__ r32, const int32_t *__restrict__ zid, int n,
                             const double *__restrict__ delta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = r[i] + delta[zid[i]];
    r[i] = v;
    r32[i] = (float)v;
}

// ---- the covariate and random-effect blocks of one iteration entirely on the device (reference src/Bayes.cpp:484-516) ----
// The host pre-draws the deviates in the reference's order (they do not depend on the data) and passes them in; nothing
// comes back until the iteration's single fetch. One workgroup each: n is a few hundred KB.
__global__ __launch_bounds__(1024) void k_cov_step(double *__restrict__ r, float *__restrict__ r32, const double *__restrict__ ci, int n,
                                                   double v, double vare, double z, double *__restrict__ beta_i)
{
    __shared__ double red[16];
    double s = 0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) s = fma(ci[k], r[k], s);
    s = block_sum(s, red);                                   // rhs = C_i . yadj            (:487)
    const double old = *beta_i;
    const double rhs = s + v * old;                          // (:488)
    const double gi = rhs / v + sqrt(vare / v) * z;          // norm_sample(rhs / v, sqrt(vare / v))  (:489)
    const double d = old - gi;                               // (:490)
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) {      // daxpy (:491)
        const double a = fma(d, ci[k], r[k]);
        r[k] = a;
        r32[k] = (float)a;
    }
    if (threadIdx.x == 0) *beta_i = gi;
}

__global__ __launch_bounds__(1024) void k_lev_step(double *__restrict__ r, float *__restrict__ r32, const int32_t *__restrict__ zid, int n,
                                                   int qr, const double *__restrict__ zz, double *__restrict__ estR,
                                                   const double *__restrict__ z, double *__restrict__ work, double vare,
                                                   double *__restrict__ vrtmp, double *__restrict__ vr, double s2r_dfr, double chis)
{
    __shared__ double red[16];
    for (int q = threadIdx.x; q < qr; q += blockDim.x) st_sc1(work + q, 0.0);
    __threadfence();
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) atomicAdd(&work[zid[k]], r[k]);     // Z' yadj            (:501)
    __threadfence();
    __syncthreads();
    const double lam = vare / *vrtmp;
    double ss = 0, sm = 0;
    for (int q = threadIdx.x; q < qr; q += blockDim.x) {
        const double rhs = ld_sc1(work + q) + zz[q] * estR[q];                            // + ZZ estR          (:502)
        const double l = zz[q] + lam;                                                     // (:504)
        const double en = rhs / l + sqrt(vare / l) * z[q];                                // (:505)
        st_sc1(work + q, estR[q] - en);                                                   // what yadj moves by (:508-510)
        estR[q] = en;
        ss = fma(en, en, ss);
        sm += en;
    }
    ss = block_sum(ss, red);
    sm = block_sum(sm, red);
    const double mean = sm / qr;
    double a2 = 0, a3 = 0;
    for (int q = threadIdx.x; q < qr; q += blockDim.x) { // arma::var, two-pass, N - 1 (:513)
        const double d = mean - estR[q];
        a2 = fma(d, d, a2);
        a3 += d;
    }
    a2 = block_sum(a2, red);
    a3 = block_sum(a3, red);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        *vrtmp = (ss + s2r_dfr) / chis;                                                   // (:512)
        *vr = qr > 1 ? (a2 - a3 * a3 / qr) / (qr - 1) : 0.0;
    }
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double a = r[k] + ld_sc1(work + zid[k]);
        r[k] = a;
        r32[k] = (float)a;
    }
}

int hbk_cov_step(hb_ctx *c, int i, double v, double vare, double z, double *beta_i)
{
    hipLaunchKernelGGL(k_cov_step, dim3(1), dim3(1024), 0, c->stream, c->r, c->r32, c->Cmat + (size_t)i * c->n, c->n, v, vare, z, beta_i);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_lev_step(hb_ctx *c, int term, int q0, int qr, const double *zz, double *estR, const double *z, double vare, double *vrtmp,
                 double *vr, double s2r_dfr, double chis)
{
    hipLaunchKernelGGL(k_lev_step, dim3(1), dim3(1024), 0, c->stream, c->r, c->r32, c->zid + (size_t)term * c->n, c->n, qr, zz + q0,
                       estR + q0, z + q0, c->lev_buf, vare, vrtmp, vr, s2r_dfr, chis);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

__global__ void k_to_f32(const double *__restrict__ r, float *__restrict__ r32, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r32[i] = (float)r[i];
}

// multi-GPU exchange: pack (yadj - yadj_start, u - u_start) and unpack the summed deltas
__global__ void k_delta_pack(const double *__restrict__ r, const double *__restrict__ u,
                             const double *__restrict__ r0, const double *__restrict__ u0, int n,
                             double *__restrict__ buf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    buf[i] = r[i] - r0[i]; // (u moved by exactly the negative: u = X g, yadj = y - ... - X g)
}

__global__ void k_delta_unpack(double *__restrict__ r, double *__restrict__ u, float *__restrict__ r32,
                               const double *__restrict__ r0, const double *__restrict__ u0, int n,
                               const double *__restrict__ buf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = r0[i] + buf[i];
    r[i] = a;
    r32[i] = (float)a;
    u[i] = u0[i] - buf[i];
}

// ---------------------------------------------------------------------------------------------
// data paths upstream of X: f64 -> int8 check, .bed decode, synthetic generator
// ---------------------------------------------------------------------------------------------
__global__ void k_f64_to_i8(const double *__restrict__ src, int64_t lds, int n, int ncols,
                            int8_t *__restrict__ dst, int64_t ldd, int *__restrict__ bad)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * ncols) return;
    const int c = (int)(idx / n), i = (int)(idx % n);
    const double v = src[(int64_t)c * lds + i];
    const double rv = rint(v);
    if (!(rv == v) || rv < -127.0 || rv > 127.0) { atomicExch(bad, 1); return; }
    dst[(int64_t)c * ldd + i] = (int8_t)rv;
}

// PLINK .bed SNP-major: byte (i>>2) of SNP j, bits 2*(i&3); map 00->2, 01->NA, 10->1, 11->0
// (reference src/read_bed.cpp:116-120).  One workgroup per SNP: count genotypes over ALL nind
// individuals (the reference imputes before ibrm() subsets rows, :182-230), then write the
// selected rows.
__global__ __launch_bounds__(256) void k_bed_decode(const uint8_t *__restrict__ bed, int64_t bpc, int nind,
                                                    const int32_t *__restrict__ rows, int n, int8_t *__restrict__ dst,
                                                    int64_t ldd)
{
    __shared__ long long red[4];
    const int j = blockIdx.x;
    const uint8_t *p = bed + (int64_t)j * bpc;
    long long c0 = 0, c1 = 0, c2 = 0, cm = 0;
    for (int i = threadIdx.x; i < nind; i += blockDim.x) {
        const int code = (p[i >> 2] >> (2 * (i & 3))) & 3;
        c2 += (code == 0);
        cm += (code == 1);
        c1 += (code == 2);
        c0 += (code == 3);
    }
    c0 = block_sum(c0, red);
    c1 = block_sum(c1, red);
    c2 = block_sum(c2, red);
    cm = block_sum(cm, red);
    int8_t major = 0;
    long long best = 0;
    if (c0 > best) { best = c0; major = 0; }
    if (c1 > best) { best = c1; major = 1; }
    if (c2 > best) { best = c2; major = 2; }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int src = rows ? rows[i] : i;
        const int code = (p[src >> 2] >> (2 * (src & 3))) & 3;
        const int8_t gg = code == 0 ? 2 : code == 2 ? 1 : code == 3 ? 0 : major;
        dst[(int64_t)j * ldd + i] = gg;
    }
    (void)cm;
}

// out[row] = sum_j x[row][j] alpha[j]  (e -= X*alpha, reference src/Bayes.cpp:971); block = 1024 rows x 256 columns
__global__ __launch_bounds__(256) void k_xalpha(const int8_t *__restrict__ X, int64_t ld, const uint32_t *__restrict__ X2, int64_t ld2w, int m_pad,
                                                const double *__restrict__ alpha, double *__restrict__ out)
{
    const int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (row0 >= ld) return;
    const int j0 = blockIdx.y * 256, j1 = min(m_pad, j0 + 256);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int j = j0; j < j1; j++) {
        const double al = alpha[j];
        if (al == 0.0) continue;
        const int w = hb_ld4(X, ld, X2, ld2w, j, row0);
        a0 = fma((double)(int8_t)(w), al, a0);
        a1 = fma((double)(int8_t)(w >> 8), al, a1);
        a2 = fma((double)(int8_t)(w >> 16), al, a2);
        a3 = fma((double)(int8_t)(w >> 24), al, a3);
    }
    if (a0 != 0.0) atomicAdd(out + row0, a0);
    if (a1 != 0.0) atomicAdd(out + row0 + 1, a1);
    if (a2 != 0.0) atomicAdd(out + row0 + 2, a2);
    if (a3 != 0.0) atomicAdd(out + row0 + 3, a3);
}

// out[rec][row] = sum_e x[row][idx[e]] * val[e][rec] for 8 sample records at once: MCMCsamples$g = M %*% MCMCsamples$alpha,
// reference R/bayes.r:303-305. The host hands over only the columns where any of the 8 records is non-zero (the
// point-mass models keep ~0.1-5 % of the markers in the model), so the work is n x nnz x 8 instead of n x m x 8.
// thread = 4 rows x 8 records (32 fp64 accumulators, no atomics); the column index and its 8 effects are wave-uniform.
#define HB_XM_RB 8
__global__ __launch_bounds__(256) void k_xmat(const int8_t *__restrict__ X, int64_t ld, const uint32_t *__restrict__ X2, int64_t ld2w, const int *__restrict__ idx,
                                              const double *__restrict__ val, int nnz, double *__restrict__ out, int64_t ldo)
{
    const int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (row0 >= ld) return;
    double acc[4][HB_XM_RB];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < HB_XM_RB; r++) acc[a][r] = 0.0;
    for (int e0 = 0; e0 < nnz; e0 += 4) {
        int w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = hb_ld4(X, ld, X2, ld2w, idx[min(e0 + k, nnz - 1)], row0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (e0 + k < nnz) { // uniform
                const double *v = val + (size_t)(e0 + k) * HB_XM_RB;
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const double x = (double)(int8_t)(w[k] >> (8 * a));
#pragma unroll
                    for (int r = 0; r < HB_XM_RB; r++) acc[a][r] = fma(x, v[r], acc[a][r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < HB_XM_RB; r++)
#pragma unroll
        for (int a = 0; a < 4; a++) out[(int64_t)r * ldo + row0 + a] = acc[a][r];
}

int hbk_xmat(hb_ctx *c, const int *didx, const double *dval, int nnz, double *dout)
{
    hipLaunchKernelGGL(k_xmat, dim3((unsigned)((c->ld / 4 + 255) / 256)), dim3(256), 0, c->stream, c->X, c->ld, c->layout == 2 ? c->X2 : nullptr, c->ld2 / 4, didx, dval, nnz, dout, c->ld);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

// synthetic genotypes, SURVEY §8(d): p_j ~ U(0.05, 0.5), x ~ Binomial(2, p_j); thread = 4 rows
__global__ __launch_bounds__(256) void k_generate(int8_t *__restrict__ X, int64_t ld, int n, int m, int64_t m_offset,
                                                  uint64_t seed, int mono_every)
{
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= ld) return;
    for (int j = blockIdx.y; j < m; j += gridDim.y) {
    const uint64_t gj = (uint64_t)(m_offset + j);
    const uint64_t sub = hb_sub(HB_PURPOSE_DATA, gj);
    const double pj = 0.05 + 0.45 * hb_uniform_blk(seed, sub, 0xFFFFFFFFFFull);
    const unsigned thr16 = (unsigned)(pj * 65536.0);
    const bool mono = mono_every > 0 && (gj % (uint64_t)mono_every) == (uint64_t)(mono_every - 1);
    const uint4 w = hb_block(seed, sub, (uint64_t)i4);
    const unsigned ws[4] = {w.x, w.y, w.z, w.w};
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        unsigned x = ((ws[k] & 0xffffu) < thr16) + ((ws[k] >> 16) < thr16);
        if (mono || i4 * 4 + k >= n) x = 0;
        out |= x << (8 * k);
    }
    *reinterpret_cast<unsigned *>(X + (int64_t)j * ld + i4 * 4) = out;
    }
}

// =============================================================================================
// host side: launchers
// =============================================================================================
static inline int kpad_for(int model, int n_fold)
{
    if (model != 6) return 1;
    const int k1 = n_fold - 1;
    return k1 <= 1 ? 1 : (k1 <= 3 ? 3 : 7);
}

// LDS budget of k_chain: as many Gram rows as fit beside the event lists
static int chain_nslot(int P) { return (int)((158 * 1024 - ((size_t)P * 16 + 128 + 128 + 64)) / ((size_t)P * 4)); }
#define HB_PERSIST_RING(P) ((size_t)4 * ((((size_t)12 * (P) + 1023) >> 10 << 10) + 1024)) /* HB_RD slots of the opening ring */
// move lists (12 B per marker) + reduction / counter words + one round's candidate staging (sized by the model's K1 non-null classes)
// + the 64 x 64 block of mutual Gram entries + the correction ring + the opening ring; the rest is the double-buffered row cache
#define HB_PERSIST_FIXED(P, LB, K1) ((size_t)(P) * 12 + 128 + 512 + 64 * (8 * (3 + 3 * (K1)) + 12) + 64 * 64 * 4 + (size_t)((LB) + 1) * (P) * 8 + HB_PERSIST_RING(P) + (size_t)2 * (P) * 8 /* k_fwd's sums, two panels */ + (72 + 64) * 16 /* a round's moves as the apply reads them */)
static int persist_nslot(int P, int Lb, int K1) { return std::min(P, std::min(250, (int)((160 * 1024 - HB_PERSIST_FIXED(P, Lb, K1)) / ((size_t)P * 4)))); }
// the whole 160 KiB: nothing that needs LDS (mat-vec, update) can then be co-scheduled on the chain's CU
static size_t persist_smem(int) { return (size_t)160 * 1024; }
static size_t chain_smem(int P) { return (size_t)chain_nslot(P) * P * 4 + (size_t)P * 16 + 128 + 128 + 64; }

int hbk_init_attrs()
{
#define HB_PERSIST_ATTR(K1, NPL) HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_persist<K1, NPL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    HB_PERSIST_ATTR(1, 0); HB_PERSIST_ATTR(1, 1); HB_PERSIST_ATTR(1, 17); HB_PERSIST_ATTR(1, 20);
    HB_PERSIST_ATTR(3, 0); HB_PERSIST_ATTR(3, 2);
    HB_PERSIST_ATTR(7, 0); HB_PERSIST_ATTR(7, 2);
#define HB_GROUP_ATTR(K1, DM, FW, CH) HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<K1, DM, FW, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    HB_GROUP_ATTR(1, 8, 14, 3); HB_GROUP_ATTR(1, 8, 7, 4); HB_GROUP_ATTR(1, 2, 4, 10); HB_GROUP_ATTR(1, 1, 2, 20);
    HB_GROUP_ATTR(3, 1, 2, 20); HB_GROUP_ATTR(7, 1, 2, 20);
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_dense<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_dense<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return HB_OK;
}

template <int K1>
static hipError_t launch_chain(hb_ctx *c, const chain_view &cv, int p, hipStream_t st)
{
    const size_t smem = chain_smem(c->P);
    hipLaunchKernelGGL(k_chain<K1>, dim3(1), dim3(c->P), smem, st, c->d_in, cv, p, chain_nslot(c->P));
    return hipGetLastError();
}

// residual version v (moves of panels <= v applied; v = -1: start of the sweep) lives in slot (v+1) mod NB
static inline int ver_slot(const hb_ctx *c, int v) { return (v + 1) % c->NB; }

// tiles of one k_dotq launch: about three waves per compute unit, each a long run of stages (measured: fewer, longer
// waves stream better than many short ones; tools/dotq_bench.hip)
static void dotq_geometry(const hb_ctx *c, int ncols, int *ncg, int *NS, int *nsplit)
{
    const int nst = (int)(c->ld / HBQ_RS);
    *ncg = ncols / 64;
    int target = 768;
    if (const char *e = getenv("HB_DOTQ_TILES")) target = std::max(1, atoi(e));
    int ns = std::max(1, std::min(nst, (int)((double)target / *ncg + 0.5)));
    *NS = std::min(1024, (nst + ns - 1) / ns); // (int32 accumulators: NS * 128 rows * 127 * 128 < 2^31)
    *nsplit = (nst + *NS - 1) / *NS;
}

// the same launch on the 2-bit resident layout (hb_dotq2.hpp): about two long-lived waves per compute unit
static void launch_dotq2(hb_ctx *c, int col0, int ncols, int slot, hipStream_t st, int gidx, const upd_view *upd, int fin_col0,
                         int fin_ncols, int fin_gidx)
{
    if (c->dotq2_kind == 1) { // rows across the lanes, no LDS (k_dotq2r): tiles = row blocks x groups of NC columns
        int NC = c->dotq2_nc;
        while (NC > Q2R_CB && ncols % NC) NC -= Q2R_CB;
        if (ncols % NC == 0 && NC % Q2R_CB == 0 && c->ld >= 64) {
            upd_view uq{};
            if (upd) uq = *upd;
            dq_view v{};
            v.X = nullptr;
            v.X2 = reinterpret_cast<const uint8_t *>(c->X2) + (int64_t)col0 * c->ld2;
            v.ld2 = c->ld2;
            v.ld = c->ld;
            v.rq = c->rq + (size_t)slot * HB_ND * c->ld;
            v.vexp_in = c->vexp + slot;
            v.gexp_out = c->gexp + gidx;
            v.accq = c->accq + col0;
            v.accstride = c->m_pad;
            v.NS = NC;
            v.ncg = ncols / NC;
            v.nstages = (int)((c->ld2 * 4 + Q2R_RB - 1) / Q2R_RB);
            v.nupd = (uq.p1 > uq.p0) ? (int)(c->ld / (uq.dense ? 64 : 256)) : 0;
            v.nfin = fin_ncols > 0 ? (fin_ncols + 63) / 64 : 0;
            v.fin_acc = c->accq + fin_col0;
            v.fin_out = c->dsum + fin_col0;
            v.fin_exp = c->gexp + fin_gidx;
            v.fin_ncols = fin_ncols;
            const int nblk = v.nupd + v.nfin + v.ncg * v.nstages;
            v.stamp = nullptr;
            v.ldiag = (c->ldiag && gidx >= 0 && gidx <= c->npanels) ? c->ldiag + 4 * (size_t)gidx : nullptr;
            if (v.ldiag) c->ldiag_nblk[gidx] = nblk;
            if (c->lstamp && gidx >= 0 && gidx <= c->npanels && nblk <= HB_LSTAMP_BLOCKS) {
                v.stamp = c->lstamp + (size_t)gidx * HB_LSTAMP_BLOCKS * 2;
                c->lstamp_nblk[gidx] = nblk;
                c->lstamp_cols[gidx] = ncols;
            }
            hipLaunchKernelGGL(k_dotq2r, dim3(nblk), dim3(64), 0, st, v, uq);
            return;
        }
    }
    const bool mfma = c->dotq2_kind == 2; // (A/B: the digit-plane product on the matrix cores, k_dotq2m; 256-individual stages, 64 columns per wave)
    int cpl = (ncols % 128 == 0 && !mfma && c->dotq2_rs != 128) ? c->dotq2_cpl : 1;
    const int RS = mfma ? Q2M_RS : c->dotq2_rs;
    const int nst = (int)((c->ld + RS - 1) / RS);
    const int ncg = ncols / (64 * cpl);
    // (a tile is at least four stages: its first stage's load latency and its closing atomics are paid per tile)
    // (the matrix-core kernel streams best with few, long tiles — its per-stage work is an eighth of the v_dot4 kernel's, so a tile's fixed
    // costs weigh more: ~800 tiles per 3584-column launch)
    int ns = std::max(1, std::min(std::max(1, nst / 4), (int)((double)(mfma && !getenv("HB_DOTQ2_TILES") ? 800 : c->dotq2_tiles) / ncg + 0.5)));
    // (int32 accumulators of genotypes scaled by up to 32 — Q2_SCALED, k_dotq2m: rows x 96 x 128 < 2^31 bounds a tile at 174 000 individuals)
    ns = std::max(ns, (int)(((int64_t)nst * RS + 131071) / 131072));
    const int NS = (nst + ns - 1) / ns, nsplit = (nst + NS - 1) / NS;
    upd_view uq{};
    if (upd) uq = *upd;
    dq_view v{};
    v.X = nullptr;
    v.X2 = reinterpret_cast<const uint8_t *>(c->X2) + (int64_t)col0 * c->ld2;
    v.ld2 = c->ld2;
    v.ld = c->ld;
    v.rq = c->rq + (size_t)slot * HB_ND * c->ld;
    v.vexp_in = c->vexp + slot;
    v.gexp_out = c->gexp + gidx;
    v.accq = c->accq + col0;
    v.accstride = c->m_pad;
    v.nstages = nst;
    v.NS = NS;
    v.ncg = ncg;
    v.nupd = (uq.p1 > uq.p0) ? (int)(c->ld / (uq.dense ? 64 : 256)) : 0;
    v.nfin = fin_ncols > 0 ? (fin_ncols + 63) / 64 : 0;
    v.fin_acc = c->accq + fin_col0;
    v.fin_out = c->dsum + fin_col0;
    v.fin_exp = c->gexp + fin_gidx;
    v.fin_ncols = fin_ncols;
    const int nblk = v.nupd + v.nfin + ncg * nsplit;
    v.stamp = nullptr;
    v.ldiag = (c->ldiag && gidx >= 0 && gidx <= c->npanels) ? c->ldiag + 4 * (size_t)gidx : nullptr;
    if (v.ldiag) c->ldiag_nblk[gidx] = nblk;
    if (c->lstamp && gidx >= 0 && gidx <= c->npanels && nblk <= HB_LSTAMP_BLOCKS) {
        v.stamp = c->lstamp + (size_t)gidx * HB_LSTAMP_BLOCKS * 2;
        c->lstamp_nblk[gidx] = nblk;
        c->lstamp_cols[gidx] = ncols;
    }
    // (the update rows stage their lists in the tile buffers: 6152 bytes, below the smallest shape's 12416)
    if (mfma) hipLaunchKernelGGL(k_dotq2m, dim3(nblk), dim3(64), q2m_lds(), st, v, uq);
    else if (cpl == 2 && RS == 512) hipLaunchKernelGGL((k_dotq2<2, 512>), dim3(nblk), dim3(64), q2_lds(2, 512), st, v, uq);
    else if (cpl == 2) hipLaunchKernelGGL((k_dotq2<2, 256>), dim3(nblk), dim3(64), q2_lds(2, 256), st, v, uq);
    else if (RS == 512) hipLaunchKernelGGL((k_dotq2<1, 512>), dim3(nblk), dim3(64), q2_lds(1, 512), st, v, uq);
    else if (RS == 128) hipLaunchKernelGGL((k_dotq2<1, 128>), dim3(nblk), dim3(64), q2_lds(1, 128), st, v, uq); // (6208 bytes of LDS per wave: twice the waves per compute unit)
    else hipLaunchKernelGGL((k_dotq2<1, 256>), dim3(nblk), dim3(64), q2_lds(1, 256), st, v, uq);
}

static void launch_dotq(hb_ctx *c, int col0, int ncols, int slot, hipStream_t st, int gidx, const upd_view *upd, int fin_col0,
                        int fin_ncols, int fin_gidx)
{
    if (c->layout == 2) {
        launch_dotq2(c, col0, ncols, slot, st, gidx, upd, fin_col0, fin_ncols, fin_gidx);
        return;
    }
    int ncg, NS, nsplit;
    dotq_geometry(c, ncols, &ncg, &NS, &nsplit);
    upd_view uq{};
    if (upd) uq = *upd;
    dq_view v{};
    v.X = c->X + (int64_t)col0 * c->ld;
    v.ld = c->ld;
    v.rq = c->rq + (size_t)slot * HB_ND * c->ld;
    v.vexp_in = c->vexp + slot;
    v.gexp_out = c->gexp + gidx;
    v.accq = c->accq + col0;
    v.accstride = c->m_pad;
    v.nstages = (int)(c->ld / HBQ_RS);
    v.NS = NS;
    v.ncg = ncg;
    v.nupd = (uq.p1 > uq.p0) ? (int)(c->ld / (uq.dense ? 64 : 256)) : 0;
    v.nfin = fin_ncols > 0 ? (fin_ncols + 63) / 64 : 0;
    v.fin_acc = c->accq + fin_col0;
    v.fin_out = c->dsum + fin_col0;
    v.fin_exp = c->gexp + fin_gidx;
    v.fin_ncols = fin_ncols;
    const int nblk = v.nupd + v.nfin + ncg * nsplit;
    v.stamp = nullptr;
    v.ldiag = (c->ldiag && gidx >= 0 && gidx <= c->npanels) ? c->ldiag + 4 * (size_t)gidx : nullptr;
    if (v.ldiag) c->ldiag_nblk[gidx] = nblk;
    if (c->lstamp && gidx >= 0 && gidx <= c->npanels && nblk <= HB_LSTAMP_BLOCKS) {
        v.stamp = c->lstamp + (size_t)gidx * HB_LSTAMP_BLOCKS * 2;
        c->lstamp_nblk[gidx] = nblk;
        c->lstamp_cols[gidx] = ncols;
    }
    hipLaunchKernelGGL(k_dotq, dim3(nblk), dim3(64), uq.dense ? HBU_LDS : HBQ_LDS, st, v, uq);
}

static void launch_dot(hb_ctx *c, int col0, int ncols, int slot = 0, hipStream_t st = nullptr, bool pipeline = false,
                       const upd_view *upd = nullptr, int red_col0 = 0, int red_ncols = 0, int gidx = 0)
{
    if (!st) st = c->stream;
    upd_view uq{};
    if (upd) uq = *upd;
    if (!pipeline) red_ncols = 0;
    if (c->precise == 2) {
        launch_dotq(c, col0, ncols, slot, st, gidx, upd, red_col0, red_ncols, gidx - 1);
        return;
    }
    const dim3 grid(ncols / 8, c->nsplit + (uq.p1 > uq.p0 ? 1 : 0) + (red_ncols > 0 ? 1 : 0)), block(256);
    const int8_t *Xp = c->X + (int64_t)col0 * c->ld;
    double *part = c->partial + col0;
    const float *r32 = c->r32 + (size_t)slot * c->ld;
    const double *r64 = c->r + (size_t)slot * c->ld;
    const bool sgn = c->xmin < 0;
    dot_sync sy{c->partial + red_col0, c->dsum + red_col0, red_ncols, c->nsplit};
    if (c->precise) {
        if (sgn) hipLaunchKernelGGL((k_dot<true, true>), grid, block, c->dot_lds, st, Xp, c->ld, r32, r64, c->nchunks, 1, part, c->m_pad, sy, uq);
        else     hipLaunchKernelGGL((k_dot<true, false>), grid, block, c->dot_lds, st, Xp, c->ld, r32, r64, c->nchunks, 1, part, c->m_pad, sy, uq);
    } else {
        if (sgn) hipLaunchKernelGGL((k_dot<false, true>), grid, block, c->dot_lds, st, Xp, c->ld, r32, r64, c->nchunks, 1, part, c->m_pad, sy, uq);
        else     hipLaunchKernelGGL((k_dot<false, false>), grid, block, c->dot_lds, st, Xp, c->ld, r32, r64, c->nchunks, 1, part, c->m_pad, sy, uq);
    }
}

// digit-plane sums of [col0, col0 + ncols) -> doubles at out (the finalize that has no later launch to ride on)
static void launch_dotq_fin(hb_ctx *c, int col0, int ncols, int gidx, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_dotq_fin, dim3((ncols + 63) / 64), dim3(64), 0, st, c->accq + col0, (int64_t)c->m_pad, ncols, c->gexp + gidx, out);
}

// sweep start of the fixed-point path: digits of residual slot 0, bound mb[0], zeroed plane sums
static void launch_quant0(hb_ctx *c, hipStream_t st)
{
    (void)hipMemsetAsync(c->accq, 0, sizeof(long long) * (size_t)HB_ND * c->m_pad, st);
    hipLaunchKernelGGL(k_quant0, dim3(1), dim3(1024), 0, st, c->r, c->ld, c->rq, c->mb, c->vexp, (const double *)nullptr);
    if (c->row_reduce) { // the shards' maxima -> one exponent for all (host round trip: this is the cross-check mode)
        (void)hipStreamSynchronize(st);
        std::vector<double> slot((size_t)std::max(1, c->row_world), 0.0);
        double mxl = 0.0;
        (void)hipMemcpy(&mxl, c->mb, sizeof(double), hipMemcpyDeviceToHost);
        slot[c->row_rank] = mxl;
        if (c->row_reduce(c->row_user, slot.data(), slot.size())) { c->row_failed = true; return; }
        for (double v : slot) mxl = std::max(mxl, v);
        (void)hipMemcpy(c->scratch, &mxl, sizeof(double), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_quant0, dim3(1), dim3(1024), 0, st, c->r, c->ld, c->rq, c->mb, c->vexp, (const double *)c->scratch);
    }
}

// row-sharded cross-check mode: the digit-plane sums of columns [col0, col0 + ncols) summed over the shards (exact: integers
// below 2^53 travel as doubles), before they are finalized
static int row_reduce_accq(hb_ctx *c, int col0, int ncols, hipStream_t st)
{
    HB_HIP(hipStreamSynchronize(st));
    std::vector<long long> hq((size_t)HB_ND * ncols);
    HB_HIP(hipMemcpy2D(hq.data(), sizeof(long long) * ncols, c->accq + col0, sizeof(long long) * c->m_pad, sizeof(long long) * ncols, HB_ND, hipMemcpyDeviceToHost));
    std::vector<double> hd(hq.size());
    for (size_t i = 0; i < hq.size(); i++) hd[i] = (double)hq[i];
    if (c->row_reduce(c->row_user, hd.data(), hd.size())) return hb_fail(HB_ERR_COMM, "row-sharded mode: the all-reduce of the digit sums failed");
    for (size_t i = 0; i < hq.size(); i++) hq[i] = (long long)hd[i];
    HB_HIP(hipMemcpy2D(c->accq + col0, sizeof(long long) * c->m_pad, hq.data(), sizeof(long long) * ncols, sizeof(long long) * ncols, HB_ND, hipMemcpyHostToDevice));
    return HB_OK;
}

// the reduction of the last launch's partials (there is no next launch to carry it)
static void launch_reduce(hb_ctx *c, int col0, int ncols, hipStream_t st, int gidx = 0)
{
    if (c->precise == 2) {
        launch_dotq_fin(c, col0, ncols, gidx, c->dsum + col0, st);
        return;
    }
    dot_sync sy{c->partial + col0, c->dsum + col0, ncols, c->nsplit};
    hipLaunchKernelGGL(k_reduce_partials, dim3((ncols + 255) / 256), dim3(256), 0, st, sy, c->m_pad);
}

// mbi: index of the group (pipeline) or panel (serial kernels) whose moves are applied — addresses the chain's bound mb[1 + mbi]
static upd_view make_upd(hb_ctx *c, int p0, int p1, int sin, int sout, unsigned *flags, int mbi)
{
    const bool fx = c->precise == 2;
    return upd_view{c->X, c->P, p0, p1, c->ev_count, c->ev_idx, c->ev_delta, c->r + (size_t)sin * c->ld,
                    c->r + (size_t)sout * c->ld, c->u, c->r32 + (size_t)sout * c->ld, flags,
                    fx ? c->rq + (size_t)sout * HB_ND * c->ld : nullptr, c->mb + (size_t)(1 + mbi) * HB_MBS, c->vexp + sout,
                    c->layout == 2 ? c->X2 : nullptr, c->ld2 / 4, 0, c->ddense};
}

struct phase_timer {
    hb_ctx *c;
    bool on;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> spans;
    size_t next = 0;
    phase_timer(hb_ctx *ctx, bool enable) : c(ctx), on(enable) {}
    hipEvent_t get()
    {
        if (next == c->ev_pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            c->ev_pool.push_back(e);
        }
        return c->ev_pool[next++];
    }
    hipEvent_t begin()
    {
        if (!on) return nullptr;
        hipEvent_t e = get();
        (void)hipEventRecord(e, c->stream);
        return e;
    }
    void end(int phase, hipEvent_t b)
    {
        if (!on) return;
        hipEvent_t e = get();
        (void)hipEventRecord(e, c->stream);
        spans.push_back({phase, {b, e}});
    }
};

// Enqueue one whole sweep (parameters already in c->d_in).
//
// Pipeline (look-ahead L panels).  Three streams; panel p's kernels and their dependencies:
//   A  mat-vec(p)   reads residual version p-L-1                 after update(p-L-1)
//   B  chain(p)     folds in the moves of panels p-L..p-1        after mat-vec(p) and chain(p-1)
//   C  update(p)    residual version p-1 -> p                     after chain(p) and update(p-1)
// so the serial chain of panel p runs while the mat-vecs of the next panels stream from HBM.  Version v
// of the residual lives in slot (v+1) mod (L+1); update(p) overwrites the slot mat-vec(p) has just read.
// `timed` serialises everything on stream A with HIP events around each kernel (same kernels, same results).
static int enqueue_sweep_kernels(hb_ctx *c, int model, int n_fold, bool timed)
{
    phase_timer tm(c, timed);
    const int kp = kpad_for(model, n_fold);
    const int L = c->Lv, np = c->npanels; // per-panel launches: lag Lv with one panel per mat-vec
    hipStream_t sA = c->stream, sB = timed ? c->stream : c->s_chain, sC = timed ? c->stream : c->s_upd;
    HB_HIP(hipMemsetAsync(c->acc, 0, sizeof(double) * HB_ACC_N, sA));
    const bool fx = c->precise == 2;
    if (fx) launch_quant0(c, sA);
    hipEvent_t t_all = tm.begin();
    {
        hipEvent_t b = tm.begin();
        pre_view pv{c->m, c->m_pad, c->m_offset, c->seed, c->xpx, c->vx, c->g, c->vargL, c->thr, c->invv, c->sdz, kp};
        hipLaunchKernelGGL(k_pre, dim3((c->m_pad + 255) / 256), dim3(256), 0, sA, c->d_in, pv);
        tm.end(3, b);
    }
    if (!timed) { // fork the chain and update streams off stream A
        HB_HIP(hipEventRecord(c->ev_fork, sA));
        HB_HIP(hipStreamWaitEvent(sB, c->ev_fork, 0));
        HB_HIP(hipStreamWaitEvent(sC, c->ev_fork, 0));
    }
    const double xabs = std::max(std::abs((double)c->xmin), std::abs((double)c->xmax));
    chain_view cv{c->m_pad, c->P, fx ? 1 : c->nsplit, L, c->Lg, c->xpx, c->vx, c->g, c->tracker, c->nzrate, c->alpha_sum, c->alpha_sq,
                  c->thr, c->invv, c->sdz, c->gram, c->partial, c->dsum, c->ev_count, c->ev_idx, c->ev_delta, c->acc,
                  c->wind, c->wflag, c->dbg, fx ? c->mb : nullptr, xabs};
    const int upd_blocks = (int)((c->ld / 4 + 255) / 256);
    // software pipeline in issue order: mat-vec runs L panels ahead of chain/update in program order too,
    // so that a plain in-order replay of the captured graph is still dependency-correct
    for (int step = 0; step < np + L; step++) {
        const int pd = step;     // panel whose mat-vec is issued now
        const int pc = step - L; // panel whose chain + update are issued now
        if (pd < np) {
            hipEvent_t b = tm.begin();
            const int vread = pd - L - 1;
            if (!timed && vread >= 0) HB_HIP(hipStreamWaitEvent(sA, c->ev_upd[vread], 0));
            launch_dot(c, pd * c->P, c->P, ver_slot(c, vread < -1 ? -1 : vread), sA, false, nullptr, 0, 0, pd);
            if (fx && c->row_reduce)
                if (int rcr = row_reduce_accq(c, pd * c->P, c->P, sA)) return rcr;
            if (fx) launch_dotq_fin(c, pd * c->P, c->P, pd, c->partial + (size_t)pd * c->P, sA); // the chain sums one "split"
            if (!timed) HB_HIP(hipEventRecord(c->ev_dot[pd], sA));
            tm.end(0, b);
        }
        if (pc >= 0) {
            hipEvent_t b = tm.begin();
            if (!timed) HB_HIP(hipStreamWaitEvent(sB, c->ev_dot[pc], 0));
            hipError_t e = kp == 1 ? launch_chain<1>(c, cv, pc, sB) : kp == 3 ? launch_chain<3>(c, cv, pc, sB) : launch_chain<7>(c, cv, pc, sB);
            if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("k_chain launch: ") + hipGetErrorString(e));
            if (!timed) HB_HIP(hipEventRecord(c->ev_chain[pc], sB));
            tm.end(1, b);
            b = tm.begin();
            if (!timed) HB_HIP(hipStreamWaitEvent(sC, c->ev_chain[pc], 0));
            const int sin = ver_slot(c, pc - 1), sout = ver_slot(c, pc);
            hipLaunchKernelGGL(k_update, dim3(upd_blocks), dim3(256), 0, sC, c->ld, make_upd(c, pc, pc + 1, sin, sout, nullptr, pc));
            if (!timed) HB_HIP(hipEventRecord(c->ev_upd[pc], sC));
            tm.end(2, b);
        }
    }
    if (!timed) { // join
        HB_HIP(hipStreamWaitEvent(sA, c->ev_upd[np - 1], 0));
        HB_HIP(hipStreamWaitEvent(sA, c->ev_chain[np - 1], 0));
    }
    {
        hipEvent_t b = tm.begin();
        const int sfin = ver_slot(c, np - 1);
        if (sfin != 0) { // the residual between sweeps lives in slot 0
            HB_HIP(hipMemcpyAsync(c->r, c->r + (size_t)sfin * c->ld, sizeof(double) * c->ld, hipMemcpyDeviceToDevice, sA));
            HB_HIP(hipMemcpyAsync(c->r32, c->r32 + (size_t)sfin * c->ld, sizeof(float) * c->ld, hipMemcpyDeviceToDevice, sA));
        }
        if (model == 5) {
            hipLaunchKernelGGL(k_bayesl_post, dim3((c->m + 255) / 256), dim3(256), 0, sA, c->d_in, c->m,
                               c->m_offset, c->seed, c->vx, c->g, c->vargL, 0);
            hipLaunchKernelGGL(k_sum_vec, dim3(1), dim3(1024), 0, sA, c->vargL, c->m, c->acc + HB_ACC_SUMVARGL);
        }
        hipLaunchKernelGGL(k_reduce_ru, dim3(1), dim3(1024), 0, sA, c->r, c->u, c->n, c->acc);
        tm.end(3, b);
    }
    tm.end(4, t_all);
    HB_HIP(hipGetLastError());
    if (timed) {
        HB_HIP(hipStreamSynchronize(c->stream));
        hb_sweep_timing T{};
        for (auto &sp : tm.spans) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, sp.second.first, sp.second.second);
            switch (sp.first) {
            case 0: T.dot_ms += ms; T.dot_launches++; break;
            case 1: T.chain_ms += ms; break;
            case 2: T.update_ms += ms; break;
            case 3: T.other_ms += ms; break;
            case 4: T.total_ms += ms; break;
            }
        }
        c->timing = T;
    }
    return HB_OK;
}

template <int K1, int NPL>
static hipError_t launch_chain_persist2(hb_ctx *c, const chain_view &cv, const persist_view &pv, hipStream_t st)
{
    hipLaunchKernelGGL((k_chain_persist<K1, NPL>), dim3(1), dim3(c->P), persist_smem(c->P), st, c->d_in, cv, pv, persist_nslot(c->P, c->L, K1));
    return hipGetLastError();
}

template <int K1>
static hipError_t launch_chain_persist(hb_ctx *c, const chain_view &cv, const persist_view &pv, hipStream_t st)
{
    // candidate rows ahead (NPL == Lb) for the band widths of the default geometries; any other band goes without
    if (K1 == 1) {
        if (pv.Lb == 20) return launch_chain_persist2<1, 20>(c, cv, pv, st); // (Lv, D) = (2, 7)
        if (pv.Lb == 17) return launch_chain_persist2<1, 17>(c, cv, pv, st); // (Lv, D) = (2, 6)
        if (pv.Lb == 1) return launch_chain_persist2<1, 1>(c, cv, pv, st);
        return launch_chain_persist2<1, 0>(c, cv, pv, st);
    }
    if (pv.Lb == 2 && !pv.fcorr) return launch_chain_persist2<K1 == 1 ? 3 : K1, 2>(c, cv, pv, st); // (with k_fwd beside it the chain requests its fold rows itself, after the rounds)
    return launch_chain_persist2<K1, 0>(c, cv, pv, st);
}

// Persistent pipeline: stream A = mat-vec launches (each also carrying an update row and a partial-sum row),
// stream B = ONE chain workgroup for the whole sweep.  Device-side hand-offs: mat-vec -> chain through dsum[]
// (NaN-prefilled, written through by the partial-sum row of the next launch); chain -> update through
// chain_done; update -> mat-vec is a kernel boundary on stream A.
// The panels [pb, pe) of a sweep (pb a multiple of D): the whole sweep, or one block of a sweep whose shards exchange their
// residual deltas every few mat-vec groups (hb_ctx_sweep_range). A range is self-contained: the residual holds every earlier
// move when it starts, so its corrections start from zero and its version ring from slot 0. `first` also prepares the
// per-sweep data (k_pre, k_hotlist, zeroed sums), `last` closes the sweep (BayesL's variances, the residual's sums).
// debug hook (hb_ctx_debug_inject_abort): raise the abort flag once the chain has published `panel` panels — what a waiter that
// timed out does — so that the tests can show a replayed sweep to be the same chain
__global__ void k_inject_abort(unsigned *flags, unsigned panel)
{
    const unsigned long long t0 = wall_clock64();
    while (ld_flag(flags + HB_FLAG_CHAIN_DONE) < panel && !ld_flag(flags + HB_FLAG_ABORT) && wall_clock64() - t0 < HB_TIMEOUT_TICKS)
        __builtin_amdgcn_s_sleep(32);
    st_flag(flags + HB_FLAG_ABORT, 1u);
}

static int enqueue_sweep_pipeline(hb_ctx *c, int model, int n_fold, int pb, int pe, bool first, bool last)
{
    const int kp = kpad_for(model, n_fold);
    const int np = pe, D = c->D, Lv = c->Lv;
    const int g0 = pb / D;                             // absolute index of the range's first mat-vec group
    const int ngroups = (np - pb + D - 1) / D;         // groups in the range
    hipStream_t sA = c->stream, sB = c->s_chain;
    // sweep start: one kernel clears the sweep sums, the flag block, the event counts (quiet panels do not write theirs) and
    // fills dsum[] with "not written yet" (a NaN no sum can produce); the residual's digit planes are then written (k_quant0,
    // one workgroup) beside k_pre / k_hotlist, which need all the other compute units. The chain must be launched BEFORE the
    // first mat-vec launch (it needs a compute unit with all of its LDS free, and back-to-back mat-vec launches never leave
    // one), so both branches start together after the join.
    // the models in which every marker moves (BayesRR / A / L) at panel 512: k_chain_dense + k_fold_dense (hb_chain_dense.hpp)
    const bool dense = kp == 1 && (model == 1 || model == 2 || model == 5) && c->P == 512 && c->dense_chain && !c->chain_alone &&
                       getenv("HB_CHAIN_ALONE") == nullptr && c->L <= HB_LBMAX;
    const bool dense_upd = dense && getenv("HB_DENSE_UPD") == nullptr;
    if (c->ldiag) HB_HIP(hipMemsetAsync(c->ldiag, 0, sizeof(unsigned long long) * 4 * ((size_t)c->npanels + 2), sA));
    hipLaunchKernelGGL(k_sweep_init, dim3(256), dim3(256), 0, sA, first ? c->acc : nullptr, c->flags, c->ev_count, c->npanels,
                       reinterpret_cast<unsigned long long *>(c->dsum), c->m_pad, pb,
                       (c->fwd_group || dense) ? reinterpret_cast<unsigned long long *>(c->fcorr) : nullptr,
                       dense ? reinterpret_cast<unsigned long long *>(c->ddense) : nullptr,
                       c->precise == 2 ? reinterpret_cast<unsigned long long *>(c->mb) : nullptr, 1 + pb / c->D, c->npanels + 2,
                       dense ? reinterpret_cast<unsigned long long *>(c->fcorr2) : nullptr, c->ev_idx,
                       reinterpret_cast<unsigned long long *>(c->ev_delta), c->P);
    const bool fx = c->precise == 2;
    if (fx) {
        HB_HIP(hipEventRecord(c->ev_dot[0], sA));
        HB_HIP(hipStreamWaitEvent(sB, c->ev_dot[0], 0));
        launch_quant0(c, sB);
        HB_HIP(hipEventRecord(c->ev_upd[1 % c->npanels], sB));
    }
    if (first) {
        pre_view pvw{c->m, c->m_pad, c->m_offset, c->seed, c->xpx, c->vx, c->g, c->vargL, c->thr, c->invv, c->sdz, kp};
        hipLaunchKernelGGL(k_pre, dim3((c->m_pad + 255) / 256), dim3(256), 0, sA, c->d_in, pvw);
    }
    const int ns = std::max(0, persist_nslot(c->P, std::min(c->L, HB_LBMAX), kp));
    // (the hot lists are rebuilt for every range: they hold the effects as they are when the range starts)
    hipLaunchKernelGGL(k_hotlist, dim3(c->npanels), dim3(c->P), 0, sA, c->d_in, c->vx, c->g, c->thr, c->xpx, c->kappa, c->P, ns, c->hot_slot,
                       c->hot_list, c->thr0f, c->tracker);
    if (fx) HB_HIP(hipStreamWaitEvent(sA, c->ev_upd[1 % c->npanels], 0));
    HB_HIP(hipEventRecord(c->ev_fork, sA));
    HB_HIP(hipStreamWaitEvent(sB, c->ev_fork, 0));
    const double xabs = std::max(std::abs((double)c->xmin), std::abs((double)c->xmax));
    chain_view cv{c->m_pad, c->P, c->nsplit, Lv, c->Lg, c->xpx, c->vx, c->g, c->tracker, c->nzrate, c->alpha_sum, c->alpha_sq,
                  c->thr, c->invv, c->sdz, c->gram, c->partial, c->dsum, c->ev_count, c->ev_idx, c->ev_delta, c->acc,
                  c->wind, c->wflag, c->dbg, fx ? c->mb : nullptr, xabs};
    const int last_panels = np - (g0 + ngroups - 1) * D;
    persist_view pv{np, D, Lv, c->L, c->Lg, pb, c->flags,
                    c->hot_slot, c->hot_list, c->thr0f, c->candf, nullptr};
    // HB_CHAIN_ALONE=1 / hb_ctx_set_profiling(c, 4) — a TIMING AND COUNTER DIAGNOSTIC, results are meaningless (it needs no
    // co-resident kernels, so it is also how k_chain_persist runs under a counter-collecting profiler, tools/chain_counters.py): the mat-vec launches run first against a pre-set
    // chain_done (their update rows find empty event lists), the chain afterwards with the device to itself; the stamped span
    // (tools/chain_timeline.py with CT_ALONE=1) is then what the chain costs without the mat-vec's memory traffic beside it.
    const bool alone = c->chain_alone || getenv("HB_CHAIN_ALONE") != nullptr;
    // the point-mass models run the group-granular chain (hb_chain_group.hpp); HB_CHAIN=panel keeps the per-panel one
    // (chain_kind bit 0: BayesB / BayesC; bit 1: the dense models too — BayesR and RR / A / L at one panel per group)
    const int shape = (D <= 1 && Lv * D <= 2) ? 2 : (D <= 2 && Lv * D <= 4) ? 1 : (D <= 8 && Lv * D <= 14) ? 0 : (c->fwd_group && Lv == 3 && D == 7 && c->P == 512) ? 0 : -1;
    const bool sparse_model = kp == 1 && (model == 3 || model == 4);
    const bool group_chain = !dense && shape >= 0 && !c->chain_alone && (sparse_model ? (c->chain_kind & 1) != 0 : ((c->chain_kind & 2) != 0 && shape == 2));
    // k_fwd beside the wide group chain: the chain folds a move into its own group and the next (15 rows, four moves per trip),
    // a second workgroup into the group after that (HB_FWD=0: the chain does all 22 rows itself, three moves per trip)
    const bool fwd = group_chain && kp == 1 && (Lv == 2 || Lv == 3) && D == 7 && c->P == 512 && c->fwd_group && !alone;
    if (fwd) pv.fcorr = c->fcorr;
    if (c->L > HB_LBMAX && !fwd)
        return hb_fail(HB_ERR_UNSUPPORTED, "three groups of seven panels of look-ahead need the group chain with k_fwd (BayesB / BayesC, panel 512)");
    if (dense) pv.fcorr = c->fcorr;
    // BayesR on the per-panel chain (round 4): k_fwd folds a panel's moves into the panels two (and, at Lv = 3, three) ahead, the chain
    // itself only into the next one — half (two thirds) of the band rows of a dense sweep leave the chain's compute unit (HB_FWD=0: off)
    const bool fwd_persist = !dense && !group_chain && kp == 3 && c->P == 512 && D == 1 && (Lv == 2 || Lv == 3) && c->fwd_group && !alone &&
                             np - pb > 2 && getenv("HB_FWD_R") == nullptr;
    if (fwd_persist) pv.fcorr = c->fcorr;
    auto launch_the_chain = [&](hipStream_t st) -> int {
        if (dense) {
            if (model == 5) hipLaunchKernelGGL((k_chain_dense<true>), dim3(1), dim3(512), persist_smem(c->P), st, c->d_in, cv, pv, c->ddense, c->fcorr2);
            else hipLaunchKernelGGL((k_chain_dense<false>), dim3(1), dim3(512), persist_smem(c->P), st, c->d_in, cv, pv, c->ddense, c->fcorr2);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("k_chain_dense launch: ") + hipGetErrorString(e));
            return HB_OK;
        }
        if (group_chain) {
            const size_t sm = persist_smem(c->P);
            if (fwd) hipLaunchKernelGGL((k_chain_group<1, 8, 7, 4>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else if (kp == 1 && shape == 0) hipLaunchKernelGGL((k_chain_group<1, 8, 14, 3>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else if (kp == 1 && shape == 1) hipLaunchKernelGGL((k_chain_group<1, 2, 4, 10>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else if (kp == 1) hipLaunchKernelGGL((k_chain_group<1, 1, 2, 20>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else if (kp == 3) hipLaunchKernelGGL((k_chain_group<3, 1, 2, 20>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else hipLaunchKernelGGL((k_chain_group<7, 1, 2, 20>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("k_chain_group launch: ") + hipGetErrorString(e));
            return HB_OK;
        }
        hipError_t e = kp == 1 ? launch_chain_persist<1>(c, cv, pv, st) : kp == 3 ? launch_chain_persist<3>(c, cv, pv, st)
                                                                                 : launch_chain_persist<7>(c, cv, pv, st);
        if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("k_chain_persist launch: ") + hipGetErrorString(e));
        return HB_OK;
    };
    if (alone) { // (the update rows poll the move counts themselves: "no moves" for every panel)
        HB_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->flags + HB_FLAG_CHAIN_DONE), 0x7ffffff0, 1, sA));
        HB_HIP(hipMemsetAsync(c->ev_count, 0, sizeof(int32_t) * (size_t)c->npanels * HB_EVS, sA));
        if (fx) HB_HIP(hipMemsetAsync(c->mb + HB_MBS, 0, sizeof(double) * ((size_t)c->npanels + 1) * HB_MBS, sA));
    }
    else {
        if (int rc = launch_the_chain(sB)) return rc;
        // (the first mat-vec launch starts when the chain is resident; HB_GATE=0 / 1 overrides: by default only where a launch's
        // update blocks can sit on every compute unit)
        bool gate = dense;
        if (const char *e = getenv("HB_GATE")) gate = atoi(e) != 0;
        if (gate) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sA, c->flags);
    }
    // the L2 warmers (k_warm): a third branch of the graph, 4 workgroups per XCD of which only the chain's XCD's stay
    int warm = 4;
    if (const char *e = getenv("HB_WARM")) warm = std::max(0, std::min(16, atoi(e)));
    if (alone || (group_chain && !c->warm_group) || fwd || dense || fwd_persist) warm = 0; // (k_fwd has the third stream)
    // BayesR with k_fwd beside the chain: the warmers on a stream of their own (HB_WARM_R: workgroups per XCD, 0 = off). They read
    // the Gram rows of EVERY marker on a panel's hot list, with or without a slot in the chain's row cache, and the rows their
    // moves fold into the next panel (the chain's share of the band)
    int warm_r = 0;
    if (fwd_persist && c->s_warm) {
        warm_r = 4;
        if (const char *e = getenv("HB_WARM_R")) warm_r = std::max(0, std::min(16, atoi(e)));
    }
    if (dense) { // (Lb + 1 target panels are open at any time: Lb ahead for their band, the chain's own for its far sub-blocks)
        HB_HIP(hipStreamWaitEvent(c->s_upd, c->ev_fork, 0));
        hipLaunchKernelGGL(k_fold_dense, dim3(8 * (c->L + 1)), dim3(256), 0, c->s_upd, cv, pv, c->ddense, c->fcorr2, c->L + 1);
        HB_HIP(hipGetLastError());
    }
    bool warm_dense = false;
    if (dense) { // the chain's own Gram reads, into its XCD's L2 ahead of it (HB_WARM_DENSE=0: off; = workgroups per XCD)
        int wd = 0; // (measured at n = 50k: 4.45-4.52 ms per 200 panels with 0, 2, 4, 8 or 16 workgroups per XCD, 2 or 4 panels ahead: no gain, off by default)
        if (const char *e = getenv("HB_WARM_DENSE")) wd = std::max(0, std::min(16, atoi(e)));
        if (wd > 0) {
            int ahead = D + 1;
            if (const char *e = getenv("HB_WARM_AHEAD")) ahead = std::max(1, atoi(e));
            hipLaunchKernelGGL(k_warm_dense, dim3(8 * wd), dim3(256), 0, c->s_upd, cv, pv, wd, ahead, reinterpret_cast<int *>(c->flags + 48));
            HB_HIP(hipGetLastError());
            warm_dense = true;
        }
    }
    if (fwd) {
        HB_HIP(hipStreamWaitEvent(c->s_upd, c->ev_fork, 0));
        if (Lv == 2) hipLaunchKernelGGL((k_fwd<7, 1, 8>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
        else hipLaunchKernelGGL((k_fwd<7, 2, 4>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
        HB_HIP(hipGetLastError());
    }
    if (fwd_persist) {
        HB_HIP(hipStreamWaitEvent(c->s_upd, c->ev_fork, 0));
        if (Lv == 2) hipLaunchKernelGGL((k_fwd<1, 1, 16>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
        else hipLaunchKernelGGL((k_fwd<1, 2, 8>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
        HB_HIP(hipGetLastError());
    }
    if (warm) {
        HB_HIP(hipStreamWaitEvent(c->s_upd, c->ev_fork, 0));
        int ahead = D + 4;
        if (const char *e = getenv("HB_WARM_AHEAD")) ahead = std::max(1, atoi(e));
        hipLaunchKernelGGL(k_warm, dim3(8 * warm), dim3(256), 0, c->s_upd, pv, cv, kp, c->gram, c->P, ahead, warm, reinterpret_cast<int *>(c->flags + 48));
        HB_HIP(hipGetLastError());
    }
    if (warm_r) {
        HB_HIP(hipStreamWaitEvent(c->s_warm, c->ev_fork, 0));
        int ahead = 2; // (measured, BayesR at n = 50k, m = 500k: off 48.3 sweeps/s, 2 panels ahead 51.2, 4 ahead 50.5, 8 ahead 50.0)
        if (const char *e = getenv("HB_WARM_AHEAD")) ahead = std::max(1, atoi(e));
        persist_view pw = pv;
        pw.Lb = 1; // (the chain folds into the next panel only)
        hipLaunchKernelGGL(k_warm, dim3(8 * warm_r), dim3(256), 0, c->s_warm, pw, cv, kp, c->gram, c->P, ahead, warm_r, reinterpret_cast<int *>(c->flags + 48));
        HB_HIP(hipGetLastError());
    }
    const bool inject = c->inject_abort_panel >= 0 && c->s_dbg && !alone;
    if (inject) { // (debug hook: a fourth branch that aborts the sweep in mid-flight)
        HB_HIP(hipStreamWaitEvent(c->s_dbg, c->ev_fork, 0));
        hipLaunchKernelGGL(k_inject_abort, dim3(1), dim3(1), 0, c->s_dbg, c->flags, (unsigned)std::min(c->inject_abort_panel, np));
        HB_HIP(hipGetLastError());
    }
    const int upd_blocks = (int)((c->ld / 4 + 255) / 256);
    // Residual versions advance per mat-vec group: version h = every panel of groups <= h applied. Mat-vec launch g
    // reads version g - Lv - 1 and, in one extra grid row, carries update(h = g - Lv): version h-1 -> h, which the
    // NEXT launch reads. Two buffers ping-pong (slot = (version + 1) & 1). No third stream, no cross-stream events.
    auto slot2 = [](int v) { return v < 0 ? 0 : ((v + 1) & 1); };
    // (g, h: group indices within the range — they drive the version slots; ga, ha: the absolute ones — they address panels)
    for (int g = 0; g < ngroups; g++) {
        const int ga = g0 + g;
        const int p0 = ga * D, p1 = std::min(np, p0 + D);
        const int h = g - Lv, ha = g0 + h;
        upd_view uq{};
        if (h >= 0) uq = make_upd(c, ha * D, std::min(np, ha * D + D), slot2(h - 1), slot2(h), c->flags, ha);
        uq.dense = (dense_upd && fx && c->layout == 8 && D <= 2) ? 1 : 0; // (one row per lane where every marker moved; the fixed-point mat-vec's single-wave update blocks)
        bool ride = h >= 0;
        if (h >= 0 && dense_upd && !fx && c->layout == 8 && D <= 2) {
            // fp32 / fp64 mat-vec (k_dot: 256-thread blocks): the dense update as its own kernel AHEAD of the launch instead of a
            // grid row in it (11.5 sweeps/s with the fused row at n = 50k, m = 500k). It writes the slot the previous launch read
            // and this launch does not touch.
            uq.dense = 1;
            hipLaunchKernelGGL(k_update_dense, dim3((unsigned)(c->ld / 64)), dim3(64), 0, sA, c->ld, uq);
            ride = false;
        }
        launch_dot(c, p0 * c->P, (p1 - p0) * c->P, slot2(g - Lv - 1), sA, true, ride ? &uq : nullptr,
                   g > 0 ? (ga - 1) * D * c->P : 0, g > 0 ? D * c->P : 0, ga);
    }
    launch_reduce(c, (g0 + ngroups - 1) * D * c->P, last_panels * c->P, sA, g0 + ngroups - 1);
    if (alone)
        if (int rc = launch_the_chain(sA)) return rc;
    for (int h = std::max(0, ngroups - Lv); h < ngroups; h++) { // the updates that had no later mat-vec to ride on
        upd_view uq = make_upd(c, (g0 + h) * D, std::min(np, (g0 + h) * D + D), slot2(h - 1), slot2(h), c->flags, g0 + h);
        if (dense_upd && c->layout == 8 && D <= 2) {
            uq.dense = 1;
            hipLaunchKernelGGL(k_update_dense, dim3((unsigned)(c->ld / 64)), dim3(64), 0, sA, c->ld, uq);
        } else hipLaunchKernelGGL(k_update, dim3(upd_blocks), dim3(256), 0, sA, c->ld, uq);
    }
    HB_HIP(hipEventRecord(c->ev_chain[0], sB));
    HB_HIP(hipStreamWaitEvent(sA, c->ev_chain[0], 0));
    if (warm || fwd || dense || warm_dense || fwd_persist) {
        HB_HIP(hipEventRecord(c->ev_upd[0], c->s_upd));
        HB_HIP(hipStreamWaitEvent(sA, c->ev_upd[0], 0));
    }
    if (warm_r) {
        HB_HIP(hipEventRecord(c->ev_chain[1 % c->npanels], c->s_warm));
        HB_HIP(hipStreamWaitEvent(sA, c->ev_chain[1 % c->npanels], 0));
    }
    if (inject) {
        HB_HIP(hipEventRecord(c->ev_dot[0], c->s_dbg));
        HB_HIP(hipStreamWaitEvent(sA, c->ev_dot[0], 0));
    }
    const int sfin = slot2(ngroups - 1);
    if (sfin != 0) {
        HB_HIP(hipMemcpyAsync(c->r, c->r + (size_t)sfin * c->ld, sizeof(double) * c->ld, hipMemcpyDeviceToDevice, sA));
        HB_HIP(hipMemcpyAsync(c->r32, c->r32 + (size_t)sfin * c->ld, sizeof(float) * c->ld, hipMemcpyDeviceToDevice, sA));
    }
    if (model == 5 && last) {
        hipLaunchKernelGGL(k_bayesl_post, dim3((c->m + 255) / 256), dim3(256), 0, sA, c->d_in, c->m, c->m_offset, c->seed,
                           c->vx, c->g, c->vargL, 0);
        hipLaunchKernelGGL(k_sum_vec, dim3(1), dim3(1024), 0, sA, c->vargL, c->m, c->acc + HB_ACC_SUMVARGL);
    }
    if (last) hipLaunchKernelGGL(k_reduce_ru, dim3(1), dim3(1024), 0, sA, c->r, c->u, c->n, c->acc);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hb_sweep_enqueue(hb_ctx *c, const hb_sweep_in *in, bool timed)
{
    if (int rc = hbk_set_timeout(c)) return rc;
    *c->h_in = *in;
    HB_HIP(hipMemcpyAsync(c->d_in, c->h_in, sizeof(hb_sweep_in), hipMemcpyHostToDevice, c->stream));
    if (timed || c->row_reduce) return enqueue_sweep_kernels(c, in->model_index, in->n_fold, true); // (row-sharded mode: host round trips inside the sweep)
    const int pb = c->rng_pe ? c->rng_pb : 0, pe = c->rng_pe ? c->rng_pe : c->npanels;
    const bool first = c->rng_pe ? c->rng_first : true, last = c->rng_pe ? c->rng_last : true;
    auto enqueue = [&]() {
        return c->pipeline ? enqueue_sweep_pipeline(c, in->model_index, in->n_fold, pb, pe, first, last)
                           : enqueue_sweep_kernels(c, in->model_index, in->n_fold, false);
    };
    if (c->inject_abort_panel >= 0 && c->pipeline) { // (debug hook: such a sweep is launched directly, never from a cached graph)
        const int rc = enqueue();
        if (last && --c->inject_abort_times <= 0) c->inject_abort_panel = -1;
        return rc;
    }
    if (!c->use_graph) return enqueue();
    if (c->graph_model == -1) { // stale: something the graphs point at has moved
        for (auto &ge : c->gcache) {
            if (ge.e) (void)hipGraphExecDestroy(ge.e);
            if (ge.g) (void)hipGraphDestroy(ge.g);
        }
        c->gcache.clear();
        c->gexec = nullptr;
        c->graph = nullptr;
        c->graph_model = 0;
    }
    c->gexec = nullptr;
    for (auto &ge : c->gcache)
        if (ge.model == in->model_index && ge.fold == in->n_fold && ge.pipeline == c->pipeline && ge.Lv == c->Lv && ge.D == c->D &&
            ge.pb == pb && ge.pe == pe)
            c->gexec = ge.e;
    if (!c->gexec) {
        HB_HIP(hipStreamSynchronize(c->stream));
        HB_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
        int rc = enqueue();
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc) return rc;
        if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        hipGraphExec_t ge = nullptr;
        HB_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        c->gcache.push_back({in->model_index, in->n_fold, c->pipeline, c->Lv, c->D, pb, pe, g, ge});
        c->gexec = ge;
    }
    HB_HIP(hipGraphLaunch(c->gexec, c->stream));
    return HB_OK;
}

// ---- co-residency probe of the persistent pipeline ----
// The pipeline's two graph branches hand-shake through memory (the chain polls dsum[], the update rows poll chain_done), so
// it only makes progress where kernels on two streams really run at the same time. Environments that serialise kernels
// (AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING, a counter-collecting profiler, a time-sliced GPU) would stall every sweep
// until its 3 s timeout. The probe: two one-lane kernels on the two streams, each raises its word and waits (<= 10 ms) for the
// other's. Both see each other only if they were co-resident.
__global__ void k_probe(unsigned *w, int me, int other)
{
    st_flag(w + me, 1u);
    const unsigned long long t0 = wall_clock64();
    while (ld_flag(w + other) == 0u) {
        if (wall_clock64() - t0 > 1000000ull) { // 10 ms at 100 MHz
            st_flag(w + me, 2u);
            return;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

int hbk_probe_concurrency(hb_ctx *c, int *concurrent)
{
    unsigned *w = c->flags + 32;
    HB_HIP(hipMemsetAsync(w, 0, 2 * sizeof(unsigned), c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(1), 0, c->s_chain, w, 0, 1);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(1), 0, c->stream, w, 1, 0);
    HB_HIP(hipGetLastError());
    HB_HIP(hipStreamSynchronize(c->s_chain));
    HB_HIP(hipStreamSynchronize(c->stream));
    unsigned h[2] = {0, 0};
    HB_HIP(hipMemcpy(h, w, sizeof(h), hipMemcpyDeviceToHost));
    *concurrent = (h[0] == 1u && h[1] == 1u) ? 1 : 0;
    return HB_OK;
}

// ---- thin launch wrappers used by hb_ctx.cpp ----
int hbk_stats(hb_ctx *c)
{
    int init[2] = {127, -128};
    HB_HIP(hipMemcpyAsync(c->xinfo, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_stats, dim3(c->m_pad), dim3(256), 0, c->stream, c->X, c->ld, c->n, c->m, c->xpx, c->vx, c->xinfo, c->s1);
    HB_HIP(hipGetLastError());
    int info[2];
    HB_HIP(hipMemcpyAsync(info, c->xinfo, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    c->xmin = info[0];
    c->xmax = info[1];
    c->stats_ready = true;
    return HB_OK;
}

int hbk_dot_all(hb_ctx *c)
{
    if (c->precise == 2) {
        launch_quant0(c, c->stream);
        for (int p = 0; p < c->npanels; p++) launch_dot(c, p * c->P, c->P);
        launch_dotq_fin(c, 0, c->m_pad, 0, c->dots, c->stream);
        HB_HIP(hipGetLastError());
        return HB_OK;
    }
    for (int p = 0; p < c->npanels; p++) launch_dot(c, p * c->P, c->P);
    hipLaunchKernelGGL(k_sum_partials, dim3((c->m_pad + 255) / 256), dim3(256), 0, c->stream, c->partial, c->m_pad,
                       c->nsplit, c->m_pad, c->dots);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_dot_panels(hb_ctx *c, int reps)
{
    for (int r = 0; r < reps; r++)
        for (int p = 0; p < c->npanels; p++) launch_dot(c, p * c->P, c->P);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_reduce_ru(hb_ctx *c)
{
    hipLaunchKernelGGL(k_reduce_ru, dim3(1), dim3(1024), 0, c->stream, c->r, c->u, c->n, c->acc);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_shift(hb_ctx *c, double a)
{
    hipLaunchKernelGGL(k_shift, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->r, c->r32, c->n, a);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_to_f32(hb_ctx *c)
{
    hipLaunchKernelGGL(k_to_f32, dim3((int)((c->ld + 255) / 256)), dim3(256), 0, c->stream, c->r, c->r32, (int)c->ld);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_cov_dot(hb_ctx *c, int i, double *dev_out)
{
    hipLaunchKernelGGL(k_dot_vec, dim3(1), dim3(1024), 0, c->stream, c->Cmat + (size_t)i * c->n, c->r, c->n, dev_out);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_cov_axpy(hb_ctx *c, int i, double a)
{
    hipLaunchKernelGGL(k_axpy, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->r, c->r32, c->Cmat + (size_t)i * c->n, c->n, a);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_level_sums(hb_ctx *c, int term, double *dev_sums, int nlev)
{
    HB_HIP(hipMemsetAsync(dev_sums, 0, sizeof(double) * nlev, c->stream));
    hipLaunchKernelGGL(k_level_sums, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->r, c->zid + (size_t)term * c->n, c->n, dev_sums);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_level_axpy(hb_ctx *c, int term, const double *dev_delta)
{
    hipLaunchKernelGGL(k_level_axpy, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->r, c->r32, c->zid + (size_t)term * c->n, c->n, dev_delta);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

// ---- snapshot / restore of the state a sweep changes (hb_ctx_snapshot / hb_ctx_restore): every segment in one launch ----
#define HB_SNAP_MAXSEG 12
struct seg_args {
    char *a[HB_SNAP_MAXSEG];       // destination
    const char *b[HB_SNAP_MAXSEG]; // source
    size_t bytes[HB_SNAP_MAXSEG];
};
__global__ __launch_bounds__(256) void k_copy_segs(seg_args s)
{
    const int k = blockIdx.y;
    char *dst = s.a[k];
    const char *src = s.b[k];
    const size_t nb = s.bytes[k], n16 = nb / 16;
    // (every buffer is a hipMalloc allocation or a 256-byte aligned offset into one)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    if (blockIdx.x == 0)
        for (size_t i = n16 * 16 + threadIdx.x; i < nb; i += blockDim.x) dst[i] = src[i];
}

int hbk_copy_segs(hb_ctx *c, const std::vector<hb_ctx::snap_seg> &segs, bool restore)
{
    if (segs.empty()) return HB_OK;
    if (segs.size() > HB_SNAP_MAXSEG) return hb_fail(HB_ERR_INVALID, "hbk_copy_segs: too many segments");
    seg_args s{};
    for (size_t k = 0; k < segs.size(); k++) {
        char *live = static_cast<char *>(segs[k].live), *copy = c->snap + segs[k].off;
        s.a[k] = restore ? live : copy;
        s.b[k] = restore ? copy : live;
        s.bytes[k] = segs[k].bytes;
    }
    hipLaunchKernelGGL(k_copy_segs, dim3(128, (unsigned)segs.size()), dim3(256), 0, c->stream, s);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

// sharded sweep: the rank whose pipeline gave up turns the event count it contributes to the exchange into a NaN, which the
// all-reduce hands to every rank — all of them then restore and replay the sweep together (hb_run::step)
__global__ void k_abort_poison(const unsigned *flags, double *sums)
{
    if (ld_flag(flags + HB_FLAG_ABORT)) sums[HB_ACC_EVENTS] = __longlong_as_double(0x7ff8000000000001ll);
}
int hbk_abort_poison(hb_ctx *c, double *sums)
{
    hipLaunchKernelGGL(k_abort_poison, dim3(1), dim3(1), 0, c->stream, c->flags, sums);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

// The bound is ONE device global per device (every kernel of the module reads it), so two contexts on one device share it: the cache of
// what was uploaded is keyed by the device ordinal and guarded — two host threads driving two contexts must not race on it — and a
// context whose value differs from the device's re-uploads before its sweep. (Contexts on one device that want DIFFERENT bounds at
// the same time get the later one for both: the bound only decides how soon a stalled sweep is given up, never a result.)
int hbk_set_timeout(hb_ctx *c)
{
    static std::mutex mu;
    static std::map<int, int> uploaded_ms;
    const int ms = std::max(1, c->timeout_ms);
    std::lock_guard<std::mutex> lk(mu);
    auto it = uploaded_ms.find(c->device);
    if (it != uploaded_ms.end() && it->second == ms) return HB_OK;
    const unsigned long long ticks = (unsigned long long)ms * 100000ull;
    HB_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(hb_timeout_ticks), &ticks, sizeof ticks, 0, hipMemcpyHostToDevice, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream)); // (the source is a stack word; this happens once per change of the value)
    uploaded_ms[c->device] = ms;
    return HB_OK;
}

unsigned hbk_long_wait_flushes()
{
    unsigned v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(hb_long_wait_flushes), sizeof v);
    return v;
}

int hbk_windows(hb_ctx *c)
{
    if (c->nw) hipLaunchKernelGGL(k_windows, dim3((c->nw + 255) / 256), dim3(256), 0, c->stream, c->wflag, c->wppa, c->nw);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_f64_to_i8(hb_ctx *c, const double *dsrc, int64_t lds, int ncols, int8_t *dst, int *dbad)
{
    const int64_t tot = (int64_t)c->n * ncols;
    hipLaunchKernelGGL(k_f64_to_i8, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, dsrc, lds, c->n, ncols, dst, c->ld, dbad);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_bed_decode(hb_ctx *c, const uint8_t *dbed, int64_t bpc, int nind, const int32_t *drows, int col0, int ncols)
{
    hipLaunchKernelGGL(k_bed_decode, dim3(ncols), dim3(256), 0, c->stream, dbed, bpc, nind, drows, c->n, c->X + (int64_t)col0 * c->ld, c->ld);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_generate(hb_ctx *c, uint64_t seed, int mono_every)
{
    const dim3 grid((unsigned)((c->ld / 4 + 255) / 256), (unsigned)std::min(c->m, 32768));
    hipLaunchKernelGGL(k_generate, grid, dim3(256), 0, c->stream, c->X, c->ld, c->n, c->m, c->m_offset, seed, mono_every);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_delta_pack(hb_ctx *c, const double *r0, const double *u0, double *buf)
{
    hipLaunchKernelGGL(k_delta_pack, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->r, c->u, r0, u0, c->n, buf);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_delta_unpack(hb_ctx *c, const double *r0, const double *u0, const double *buf)
{
    hipLaunchKernelGGL(k_delta_unpack, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->r, c->u, c->r32, r0, u0, c->n, buf);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_xalpha(hb_ctx *c, const double *dev_alpha, double *dev_out)
{
    HB_HIP(hipMemsetAsync(dev_out, 0, sizeof(double) * (size_t)c->ld, c->stream));
    const dim3 grid((unsigned)((c->ld / 4 + 255) / 256), (unsigned)((c->m_pad + 255) / 256));
    hipLaunchKernelGGL(k_xalpha, grid, dim3(256), 0, c->stream, c->X, c->ld, c->layout == 2 ? c->X2 : nullptr, c->ld2 / 4, c->m_pad, dev_alpha, dev_out);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_pack2(hb_ctx *c)
{
    const int64_t ld2w = c->ld2 / 4;
    hipLaunchKernelGGL(k_pack2, dim3((unsigned)(((ld2w + 255) / 256) * c->m_pad)), dim3(256), 0, c->stream, c->X, c->ld, c->X2, ld2w, c->m_pad);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_unpack2(hb_ctx *c, int col0, int ncols, int8_t *dst)
{
    const int64_t ld2w = c->ld2 / 4;
    hipLaunchKernelGGL(k_unpack2, dim3((unsigned)(((c->ld / 16 + 255) / 256) * (int64_t)ncols)), dim3(256), 0, c->stream,
                       c->X2 + (int64_t)col0 * ld2w, ld2w, dst, c->ld, ncols);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

#include "hb_sbayes.hpp"

// hb_ctx_time_matvec: the panel mat-vec launches of one sweep, as the pipeline issues them (same grouping, same
// partial-sum rows; no update row), back to back on the context's stream between two HIP events
int hbk_time_matvec(hb_ctx *c, int D, int reps, int as_pipeline, double *avg_us, int *launches)
{
    HB_HIP(hipSetDevice(c->device));
    hipEvent_t e0, e1;
    HB_HIP(hipEventCreate(&e0));
    HB_HIP(hipEventCreate(&e1));
    const int ngroups = (c->npanels + D - 1) / D;
    HB_HIP(hipMemsetAsync(c->flags, 0, sizeof(unsigned) * HB_NFLAGS, c->stream));
    if (c->precise == 2) launch_quant0(c, c->stream);
    // one pass over the sweep's launches, captured into a graph as the sweep itself is (the launches then follow each other
    // as closely as they do in a run), replayed once untimed and `reps` times between the two events
    HB_HIP(hipStreamSynchronize(c->stream));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HB_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    for (int gi = 0; gi < ngroups; gi++) {
        const int p0 = gi * D, p1 = std::min(c->npanels, p0 + D);
        launch_dot(c, p0 * c->P, (p1 - p0) * c->P, 0, c->stream, as_pipeline != 0, nullptr,
                   gi > 0 ? (gi - 1) * D * c->P : 0, gi > 0 ? D * c->P : 0, gi);
    }
    if (as_pipeline) launch_reduce(c, (ngroups - 1) * D * c->P, (c->npanels - (ngroups - 1) * D) * c->P, c->stream, ngroups - 1);
    HB_HIP(hipStreamEndCapture(c->stream, &g));
    HB_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HB_HIP(hipGraphLaunch(ge, c->stream));
    HB_HIP(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; r++) HB_HIP(hipGraphLaunch(ge, c->stream));
    HB_HIP(hipEventRecord(e1, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    float ms = 0;
    HB_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1e3 / ((double)reps * ngroups);
    if (launches) *launches = ngroups;
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return HB_OK;
}

extern "C" int hbk_dot_bench(hb_ctx *c, int D, int reps, int as_pipeline, double *avg_us)
{
    return hbk_time_matvec(c, D, reps, as_pipeline, avg_us, nullptr);
}
