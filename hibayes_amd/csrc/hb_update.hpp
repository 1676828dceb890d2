// hb_update.hpp — residual update rows: yadj -= X[:, moved] delta, u += the same (the rows that ride in a mat-vec launch, and the dense form for the models in which every marker moves).
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

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
template <bool WT = false>
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
        if constexpr (WT) __hip_atomic_store(reinterpret_cast<unsigned *>(rq + (int64_t)k * ld + row0), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *reinterpret_cast<unsigned *>(rq + (int64_t)k * ld + row0) = w;
    }
}

// rows [row0, row0 + 4) of the residual: yadj -= sum_e x_e D_e, u += the same, r32 = (float)yadj
// (MVP: inside the persistent mat-vec, hb_mvp.hpp — memory-side looks every fourth look of a wait, planes and exponent written through; a template
// constant, so that the update rows of the pipeline's launches are the code they were)
template <bool MVP = false>
__device__ __forceinline__ void update_rows(int64_t ld, const upd_view &q, int blk, int *s_ix,
                                            double *s_dl, int *s_ok, unsigned long long *ust = nullptr, int mvp_fresh = 0, int mvp_sleep = 0)
// (mvp_fresh, mvp_sleep — update_rows<true> only: every mvp_fresh-th look of a wait at the memory side, mvp_sleep extra naps of 64 x 64 cycles per look)
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
        unsigned looks = 0;
        (void)looks;
        for (;;) {
            if constexpr (MVP) {
                if (q.rq) mbv = ld_poll(q.mbv, looks, (unsigned)mvp_fresh);
#pragma unroll
                for (int i = 0; i < 8; i++) nevs[i] = ld_poll(q.ev_count + (size_t)min(q.p0 + i, q.p1 - 1) * HB_EVS, looks, (unsigned)mvp_fresh);
                looks++;
            } else {
            if (q.rq) mbv = ld_sc1(q.mbv); // (every thread the same word: one broadcast load per wave, in flight with the counts)
#pragma unroll
            for (int i = 0; i < 8; i++) nevs[i] = ld_sc1(q.ev_count + (size_t)min(q.p0 + i, q.p1 - 1) * HB_EVS);
            }
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
            if constexpr (MVP) for (int z = 0; z < mvp_sleep; z++) __builtin_amdgcn_s_sleep(64);
        }
        if (q.rq) { // (uniform) exponent of the new version, the same number in every workgroup
            fixE = hb_fix_exp(mbv);
            if (blk == 0 && threadIdx.x == 0) { if constexpr (MVP) st_sc1(q.vexp_out, fixE); else *q.vexp_out = fixE; }
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
                unsigned looks = 0;
                (void)looks;
                while (ix < 0 || __double_as_longlong(dl) == -1ll) {
                    if (ld_flag(q.flags + HB_FLAG_ABORT) || wall_clock64() - t1 > HB_TIMEOUT_TICKS) { st_flag(q.flags + HB_FLAG_ABORT, 1u); ix = 0; dl = 0.0; break; }
                    __builtin_amdgcn_s_sleep(2);
                    if constexpr (MVP) {
                        ix = ld_poll(q.ev_idx + src, looks, (unsigned)mvp_fresh);
                        dl = ld_poll(q.ev_delta + src, looks, (unsigned)mvp_fresh);
                        looks++;
                    } else {
                    ix = ld_sc1(q.ev_idx + src);
                    dl = ld_sc1(q.ev_delta + src);
                    }
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
    if (q.rq) hb_store_digits<MVP>(q.rq, ld, row0, fixE, r01.x, r01.y, r23.x, r23.y);
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

