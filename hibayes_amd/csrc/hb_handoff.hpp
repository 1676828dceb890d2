// hb_handoff.hpp — the hand-off words of the persistent pipeline: flag block, agent-scope loads / write-through stores, bounded waits, the abort log; wave and block reductions.
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

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
// Round 6: the same decided by the CALLER (`every`: 0 = never; a kernel argument or a template constant — never a device global: read in the update rows'
// first look, its scalar load cost every launch of the pipeline 0.6 us, 446 -> 421 sweeps/s). Under the launches of the pipeline a reader's stale line lives
// at most until the next launch starts — every kernel begins by invalidating its XCD's L2 — which is why the stale-line stall was rare. With the PERSISTENT
// mat-vec (hb_mvp.hpp) nothing is launched for the whole sweep: a line fetched before its word was published is served from the reader's L2 until something
// evicts it (the chain workgroup polled its first group's dots too early and never saw them: every sweep timed out). Every fourth look of every wait is
// then a memory-side one.
// (uniform) is this look of a wait a memory-side one?
__device__ __forceinline__ bool hb_fresh_look(unsigned looks, unsigned every = 0u)
{
#if HB_FRESH_EVERY > 0
    if ((looks % HB_FRESH_EVERY) == HB_FRESH_EVERY - 1) return true;
#endif
    return every != 0u && (looks % every) == every - 1u;
}
__device__ __forceinline__ int ld_poll(const int *p, unsigned looks, unsigned every = 0u) { return hb_fresh_look(looks, every) ? ld_fresh(p) : ld_sc1(p); }
__device__ __forceinline__ double ld_poll(const double *p, unsigned looks, unsigned every = 0u) { return hb_fresh_look(looks, every) ? ld_fresh(p) : ld_sc1(p); }
__device__ __forceinline__ unsigned ld_poll_flag(const unsigned *p, unsigned looks, unsigned every = 0u) { return hb_fresh_look(looks, every) ? ld_flag_fresh(p) : ld_flag(p); }

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

