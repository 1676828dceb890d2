// hb_chain_group.hpp — k_chain_group: the persistent chain workgroup of the sparse regime (BayesB / BayesC), one step per
// MAT-VEC GROUP instead of one per panel. Included by hb_kernels.hip (it uses that file's views and hand-off helpers).
//
// Why. k_chain_persist pays its fixed path — opening barrier, ring bookkeeping, DMA issue, ~600 instructions per wave — once
// per panel of 512 markers, 977 times a sweep, although in the stationary regime of a point-mass model fewer than one marker
// per panel moves: the chain was bound by its own instruction stream (round 2: 1 340 instructions per wave and panel, half
// of them scalar bookkeeping), not by the work the moves themselves need. Here a step covers the D panels of one mat-vec
// launch (3 584 markers at D = 7): thread t holds marker t of each of the D panels in registers, so the fixed path is paid
// 140 times a sweep, and what remains per group is three dependent memory round trips:
//   1. the group's dots and filter words (the kernel polls its own dots: dsum[] is NaN-prefilled, every value lands whole);
//   2. the exact per-marker data of the CANDIDATES (in the model, or q at its entry threshold) and, in the same trip, the
//      Gram entries among them (single int32 entries of the band blocks: candidates of different panels of the group meet in
//      block l = panel distance);
//   3. the Gram rows of the markers that moved: onto the group's later markers (then the violation check: a marker pushed
//      over its threshold by a move joins the candidates and the round is repeated — the result is always the exact
//      sequential chain) and forward into the correction ring of the next Lv groups.
// The serial pass itself (wave 0, one candidate per lane, in marker order) is LDS-only.
// Same chain as k_chain / k_chain_persist: same decisions and move lists; effects differ in the last bits only through the
// order in which corrections are summed (tests: draw for draw against the oracle, tests/test_gpu_depth.py).
//
// Reference: the per-marker conditionals are src/Bayes.cpp:627-717 (BayesB / BayesC) — restated as thresholds by k_pre.
#pragma once

#if HB_STAMPS
// Stamped build (tools/group_timeline.py): per mat-vec group 32 words — [0..9] cycles ACCUMULATED per phase over all rounds of the group, rolled-back
// and repeated ones included (round 4 kept one stamp per phase, overwritten by every repetition of the first round, which made the
// candidate ranking look like the longest phase: it was the roll-backs) — 0 opening (waiting for the dots), 1 ranking, 2 exact data,
// 3 gather, 4 serial pass, 5 fold + verify, 6 commit + publish, 7 forward, 8 end of group; counters: 10 moves, 11 waited for
// dots, 12 candidates of the first round, 13 committed rounds, 14 rolled-back rounds; 16 / 17 clock at
// the group's start / end.
// The stamps accumulate in LDS (no-return ds_add: a few cycles each) and reach v.dbg once per group: a read-modify-write of global memory per stamp cost
// ~500 cycles each, twenty times per group — a fifth of what was being measured.
#define HBG_ADD(k, x) (void)__hip_atomic_fetch_add(&hbg_lds[k], (unsigned long long)(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define HBG_BEGIN() do { if (v.dbg && t == 0) { hbg_tl = clock64(); hbg_t0 = hbg_tl; } } while (0)
#define HBG_ACC(k) do { if (v.dbg && t == 0) { const long long now_ = clock64(); HBG_ADD(k, now_ - hbg_tl); hbg_tl = now_; } } while (0)
#define HBG_CNT(k, x) do { if (v.dbg && t == 0) HBG_ADD(k, x); } while (0)
#define HBG_MARK(k) do { if (v.dbg && t == 0) HBG_ADD(k, clock64() - hbg_tl); } while (0) /* cycles into the current phase */
#define HBG_END() do { if (v.dbg && t == 0) { hbg_lds[16] = (unsigned long long)hbg_t0; hbg_lds[17] = (unsigned long long)clock64(); \
        for (int z_ = 0; z_ < 32; z_++) { v.dbg[(size_t)gcount * 32 + z_] = (long long)hbg_lds[z_]; hbg_lds[z_] = 0ull; } } } while (0)
#else
#define HBG_BEGIN() do { } while (0)
#define HBG_ACC(k) do { } while (0)
#define HBG_CNT(k, x) do { } while (0)
#define HBG_MARK(k) do { } while (0)
#define HBG_END() do { } while (0)
#endif
// Template shape: HBG_DM = panels per group the register arrays are sized for (>= D), HBG_FW = panels ahead a move is folded into
// (>= Lv * D), HBG_CH = moves whose rows are requested together — HBG_CH * (HBG_DM + HBG_FW) loads per lane and trip, ~60: a narrow
// geometry (few rows per move) takes many moves per trip, the wide one of the stationary point-mass sweep three.
// G16 (round 5): the rows of a move come from the compact band (hb_ctx.gram16: int16 residuals of G - ga (x) gB, half the bytes), one short per
// lane and row, and every entry is rebuilt exactly, G[k][j] = g16[k][j] + ga[k] * gB[j] (one integer multiply-add), before it is used: the fold's
// arithmetic is the int32 band's bit for bit. (tools/rowfetch2_bench.hip: 80 cycles per row this way against 111 for int32 rows; whole rows per
// wave-load staged through LDS, by dwordx4 loads or by LDS-DMA: 127-131 — tried in the kernel too, profiles/r05_group_phases_g16b_*.txt.)
// CERT (round 5): the violation check of a round WITHOUT the Gram rows of the group's own panels. With G[k][j] = ga[k] gB[j] + c[k][j] and
// |c[k][j]| <= gcmax[k] (hb_ctx gcert arrays: exact integers from the band), a passed-over marker j ends the round at
//     rhs_j - gB[j] * A_j - eps_j,   A_j = sum of ga[k] d_k over the moves before j,   |eps_j| <= E = sum_k gcmax[k] |d_k|,
// so (|rhs_j - gB[j] A_j| + E)^2 < thr_j PROVES that j stayed below its threshold, and (|...| - E)^2 >= thr_j proves that it crossed. The rank-one
// part is what a move does to every other marker (n mean_k mean_j: hundreds), E what is left (n cov + rounding: a few tens for a round's moves):
// nearly every marker is decided by the certificate. A round whose passed-over markers are all proven to stay fetches only the rows of the
// NEXT group's panels (7 of 15 per move, eight moves per trip instead of four); a marker that is not proven to stay (nearly always: a real
// crosser) joins the candidates and the round is repeated before anything is fetched — a candidate is decided exactly, so an unnecessary
// one costs a lane of the serial pass and nothing else. The passed-over markers' own right-hand sides are not needed again once a round
// reaches the group's end, which is the only case certified; a round that does not (more than 64 candidates: cold and dense sweeps) takes
// the full fold and the exact check as before. Decisions and move lists are the plain path's; where a group needs one round (the stationary
// regime) every sum is too, bit for bit; where the two paths cut a crowded group into rounds differently the forward sums are grouped
// differently and effects agree to the last bits' rounding (tests: HB_CERT=0 against 1).
#ifndef HBG_CH2MAX
#define HBG_CH2MAX 31
#endif
#ifndef HB_CHAINDBG
#define HB_CHAINDBG 0 /* development aid: progress markers of the chain workgroup in flags[20..22] (group, phase, rounds) */
#endif
#define HBG_DBGW(code) do { if (HB_CHAINDBG && lane == 0) st_flag(pv.flags + 24 + wave, (unsigned)(code)); } while (0)
#define HBG_DBG(code) do { if (HB_CHAINDBG && t == 0) { st_flag(pv.flags + 20, (unsigned)gcount); st_flag(pv.flags + 21, (unsigned)(code)); } } while (0)
#ifndef HBG_CERT_MARGIN
#define HBG_CERT_MARGIN 1.0
#endif
template <int K1, int HBG_DM, int HBG_FW, int HBG_CH, bool G16 = false, bool CERT = false, bool FRESH = false>
__device__ __forceinline__ void chain_group_body(const hb_sweep_in *__restrict__ pin, const chain_view &v, const persist_view &pv, char *smem)
{
    const int P = v.P, S = P >> 6;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int lgP = 31 - __clz(P);
    const int D = pv.D, np = pv.npanels;
    const int R = (pv.fcorr ? 2 : pv.Lv + 1) * D; // correction ring: this group's panels and those of the next Lv groups (with k_fwd: the next one)
    const size_t PP = (size_t)P * P;
    const unsigned long long lt = (1ull << lane) - 1ull;

    double *corr = reinterpret_cast<double *>(smem);            // [R][P]
    // CERT: the STAGED opening. What a group's opening needs of its markers that no other workgroup of the sweep writes — candidate threshold, filter
    // word, gB: 16 bytes per marker, pv.opn, written by k_hotlist — is brought here for the NEXT group by LDS-DMA (no registers: holding the words
    // in flight in registers spilled 90 of them) by the seven waves that stand at the serial pass's barrier anyway; row D holds NaN records for
    // the panels past the sweep's end. The opening then asks memory only for the dots and k_fwd's corrections.
    int4 *nx = reinterpret_cast<int4 *>(corr + (size_t)R * P);  // [D + 1][P]
    double *cs_d = reinterpret_cast<double *>(nx + (CERT ? (size_t)(D + 1) * P : 0)); // one round's candidates: rhs, gold, thr[K1], invv[K1], sdz[K1]
    double *res_g = cs_d + (2 + 3 * K1) * 64;                   // ... their new effects
    double *ev_del = res_g + 64;                                // the round's moves: change of effect
    double *red = ev_del + 64;                                  // [16]
    double *spre = red + 16;                                    // [68] CERT: [k] = sum of ga * change over the round's candidates before candidate k (k = 0..64); [66] E, [67] max |prefix|
    int *cs_pos = reinterpret_cast<int *>(spre + 68);           // candidate -> position in the group (panel * P + marker)
    int *res_c = cs_pos + 64;                                   // ... new classes
    int *ev_pos = res_c + 64;                                   // the round's moves: position
    int *cg = ev_pos + 64;                                      // [64][64] Gram entries among the round's candidates (k < c)
    int *cs_ga = cg + 64 * 64;                                  // [64] G16: the candidates' ga[]
    int *ev_ga = cs_ga + 64;                                    // [64] G16: ga[] of the round's movers
    int *cs_cm = ev_ga + 64;                                    // [64] CERT: the candidates' gcmax[]
    int *wcnt = cs_cm + 64;                                     // [HBG_DM][8] candidates per (panel of the group, wave)
    int *misc = wcnt + 64;                                      // [0] moves of the round, [1] position the round ends at, [2] abort, [8..15] violations per wave, [16..23] moves published per panel, [24..31] CERT: per wave, a marker not proven to stay
    for (int l = 0; l < R; l++) corr[(size_t)l * P + t] = 0.0;
    if (t < 64) { wcnt[t] = 0; if (t < 32) misc[t] = 0; }

    const int model = pin->model_index;
    const int count_pip = pin->count_pip, store = pin->store;

    double wacc = 0.0;
    int nact = 0, cacc[K1 + 1];
#pragma unroll
    for (int c = 0; c <= K1; c++) cacc[c] = 0;
    int evacc = 0, redoacc = 0;
    double mbr = v.mb ? v.mb[0] : 0.0;
    int gcount = pv.p0 / D;
    bool ok = true;
    if (t == 0) { // where this workgroup runs: k_warm's workgroups on the same XCD (= the same L2) fetch the Gram rows ahead of it
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st_flag(pv.flags + HB_FLAG_XCC, (xcc & 15u) + 1u);
    }
    __syncthreads();

    long long hbg_tl = 0, hbg_t0 = 0;
    (void)hbg_tl; (void)hbg_t0;
#if HB_STAMPS
    unsigned long long *hbg_lds = reinterpret_cast<unsigned long long *>(misc + 32); // [32]
    if (t < 32) hbg_lds[t] = 0ull;
#endif
    // the records of the group that starts at panel gn0 as 1-KiB pieces (64 markers), dealt over the waves w0 .. S - 1
    auto stage_next = [&](int gn0, int w0) {
        const int npc = min(D, np - gn0) * (P >> 6);
        const char *src = reinterpret_cast<const char *>(pv.opn + (size_t)gn0 * P);
        const unsigned long long sb = (unsigned long long)(uintptr_t)src;
        const unsigned dst0 = (unsigned)(uintptr_t)nx;
        for (int k = __builtin_amdgcn_readfirstlane(wave) - w0; k < npc; k += S - w0) {
            const unsigned long long a = sb + ((unsigned long long)k << 10);
            const char *sbase = reinterpret_cast<const char *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                                                           (unsigned)__builtin_amdgcn_readfirstlane((int)a)));
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(dst0 + ((unsigned)k << 10)));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"((unsigned)lane * 16u), "s"(sbase), "s"(dst) : "memory");
        }
    };
    (void)stage_next;
    if constexpr (CERT) {
        nx[(size_t)D * P + t] = make_int4(0, 0x7ff80000, 0x7fc00000, 0); // (threshold NaN, filter word NaN, gB 0)
        if (pv.p0 < np) stage_next(pv.p0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    int gslot = 0; // ring slot of the group's first panel
    HBG_DBG(100); // staged, about to open the first group
    if (HB_CHAINDBG && t == 0) st_flag(pv.flags + 43, (unsigned)wall_clock64());
    for (int gp0 = pv.p0; ok && gp0 < np; gp0 += D) {
        const int Dg = min(D, np - gp0);
        HBG_BEGIN();
        // ---- (1) the group's dots, filter words and owed corrections ----
        double r0[HBG_DM];
        float fl[HBG_DM];
        int gBi[(G16 || CERT) ? HBG_DM : 1], gBf[G16 ? HBG_FW : 1]; // gB of this thread's markers in the group (G16, CERT) and in the panels ahead (G16)
        float sqf[CERT ? HBG_DM : 1];                                // CERT: lower bound of sqrt(threshold) of this thread's markers
        (void)gBi; (void)gBf; (void)sqf;
        if constexpr (G16 && !CERT) {
#pragma unroll
            for (int i = 0; i < HBG_DM; i++) gBi[i] = v.gB[(size_t)(gp0 + min(i, Dg - 1)) * P + t];
        }
        double thc[HBG_DM];  // candidate thresholds of the group's rounds (candf * filter word, as a double; hot: -1, filtered out: NaN)
        bool staged = false; // (uniform) the next group's records have been requested
        (void)staged;
        if constexpr (G16) {
#pragma unroll
            for (int x = 0; x < HBG_FW; x++) gBf[x] = v.gB[(size_t)min(gp0 + D + x, np - 1) * P + t];
        }
        {
            double dj[HBG_DM], fc[HBG_DM];
            // (k_fwd writes the corrections the moves of the group before the last owe this one: sentinel-prefilled like the dots)
            const bool far_in = pv.fcorr != nullptr && gp0 - pv.p0 >= 2 * D;
            bool bad = false;
            // (every load of the opening AND of a poll is unconditional, on a clamped address: 2 HBG_DM loads in flight, one round
            // trip per look. Re-reading only what is missing puts each load behind a branch — a dozen dependent trips per look,
            // most of them after the values have arrived)
            const double *fcp = far_in ? pv.fcorr : v.dsum; // (without k_fwd's share: any readable words, not looked at)
#pragma unroll
            for (int i = 0; i < HBG_DM; i++) {
                const size_t j = (size_t)(gp0 + min(i, Dg - 1)) * P + t;
                dj[i] = ld_sc1(&v.dsum[j]);
                if constexpr (!CERT) fl[i] = pv.thr0f[j];
                fc[i] = ld_sc1(&fcp[j]);
            }
            if constexpr (CERT) { // (the staged records: read while the loads above fly)
#pragma unroll
                for (int i = 0; i < HBG_DM; i++) {
                    const int4 rec = nx[(size_t)(i < Dg ? i : D) * P + t]; // (scalar row select: a panel past the group's end reads the NaN row)
                    thc[i] = __hiloint2double(rec.y, rec.x);
                    fl[i] = __int_as_float(rec.z);
                    gBi[i] = rec.w;
                    // a LOWER bound of the square root of the (float) threshold: the certificate compares magnitudes, not squares (v_sqrt_f32: 1 ulp;
                    // the correctly rounded root is thirty instructions per marker)
                    sqf[i] = __builtin_amdgcn_sqrtf(fl[i]) * (1.0f - 1e-6f);
                }
            } else {
#pragma unroll
                for (int i = 0; i < HBG_DM; i++) fl[i] = i < Dg ? fl[i] : __int_as_float(0x7fc00000);
#pragma unroll
                for (int i = 0; i < HBG_DM; i++) thc[i] = (fl[i] == -__int_as_float(0x7f800000)) ? -1.0 : pv.candf * (double)fl[i];
            }
            // (the sentinel is a NaN and no delivered word is one: a NaN anywhere shows in the sum — fifteen adds and one compare instead of
            // sixteen 64-bit compares under scalar masks. A panel past the group's end reads the last panel's words again; without k_fwd's
            // share fc[] holds the dots a second time: neither needs a mask)
            {
                double sdf = dj[0] + fc[0];
#pragma unroll
                for (int i = 1; i < HBG_DM; i++) sdf += dj[i] + fc[i];
                bad = sdf != sdf;
            }
            HBG_CNT(11, bad ? 1 : 0);
            HBG_MARK(19); // (the opening's values are in registers)
            if (__any(bad)) { // the mat-vec (or k_fwd) has not delivered (all of) this group yet: look again
                const unsigned long long t0 = wall_clock64();
                if (HB_CHAINDBG && t == 0) st_flag(pv.flags + 44, (unsigned)t0);
                for (unsigned looks = 0;; looks++) {
                    bad = false;
                    if (HB_CHAINDBG && t == 0) st_flag(pv.flags + 45, (unsigned)wall_clock64());
                    if (FRESH || hb_fresh_look(looks)) { // (uniform; what is still missing is read at the memory side, ld_fresh in hb_handoff.hpp: once in
                        // HB_FRESH_EVERY looks — or, beside the persistent mat-vec (FRESH: an instantiation of its own — as a run-time switch this branch cost the headline kernel, which sits at 256 registers, six spilled ones and 446 -> 421 sweeps/s), at every look: there no kernel boundary ever drops a line this XCD's L2
                        // took before its words were published, and a word that HAS arrived is kept — it never changes again)
#pragma unroll
                        for (int i = 0; i < HBG_DM; i++) {
                            const size_t j = (size_t)(gp0 + min(i, Dg - 1)) * P + t;
                            if (__double_as_longlong(dj[i]) == -1ll) dj[i] = ld_fresh(&v.dsum[j]);
                            if (far_in && __double_as_longlong(fc[i]) == -1ll) fc[i] = ld_fresh(&fcp[j]);
                            if (!far_in) fc[i] = dj[i]; // (fc[] aliases the dots there: it must not keep the sum a NaN once the dots are in hand)
                        }
                    } else {
#pragma unroll
                    for (int i = 0; i < HBG_DM; i++) {
                        const size_t j = (size_t)(gp0 + min(i, Dg - 1)) * P + t;
                        dj[i] = ld_sc1(&v.dsum[j]);
                        fc[i] = ld_sc1(&fcp[j]);
                    }
                    }
                    {
                        double sdf = dj[0] + fc[0];
#pragma unroll
                        for (int i = 1; i < HBG_DM; i++) sdf += dj[i] + fc[i];
                        bad = sdf != sdf;
                    }
                    if (!__any(bad)) break;
                    if (HB_CHAINDBG) { // what is missing, as the chain sees it: lanes with a NaN sum per wave, and thread 0's first words
                        const unsigned long long bm = __ballot(bad);
                        if (lane == 0) st_flag(pv.flags + 32 + wave, (unsigned)__popcll(bm));
                        if (t == 0) { st_flag(pv.flags + 22, looks); st_flag(pv.flags + 40, (unsigned)(__double_as_longlong(dj[0]) >> 32)); st_flag(pv.flags + 41, (unsigned)(__double_as_longlong(fc[0]) >> 32)); st_flag(pv.flags + 42, far_in ? 1u : 0u); }
                    }
                    {   // (advisor finding, round 5) "not delivered" is the sentinel's BIT PATTERN; a delivered word that is a NaN or an infinity (an upstream
                        // overflow) makes the sum a NaN too and would be polled until the time-out, replayed three times and reported as a time-out. If
                        // every word of a lane with a NaN sum is there, the fault is numerical: stop now and say so (fetch_acc: h_flags[15])
                        bool missing = false;
#pragma unroll
                        for (int i = 0; i < HBG_DM; i++) missing |= __double_as_longlong(dj[i]) == -1ll || (far_in && __double_as_longlong(fc[i]) == -1ll);
                        if (__any(bad && !missing)) {
                            if (lane == 0) { st_flag(pv.flags + 15, 1u); st_flag(pv.flags + HB_FLAG_ABORT, 1u); misc[2] = 1; }
                            break;
                        }
                    }
                    if (ld_flag(pv.flags + HB_FLAG_ABORT) || wall_clock64() - t0 > HB_TIMEOUT_TICKS) {
                        if (lane == 0) { st_flag(pv.flags + HB_FLAG_ABORT, 1u); misc[2] = 1; }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    hb_long_wait(looks);
                }
            }
            HBG_MARK(20);
#pragma unroll
            for (int i = 0; i < HBG_DM; i++) fc[i] = far_in ? fc[i] : 0.0;
#pragma unroll
            for (int i = 0; i < HBG_DM; i++) {
                r0[i] = 0.0;
                if (i < Dg) {
                    double *cp = corr + (size_t)(gslot + i) * P + t;
                    r0[i] = dj[i] - *cp - fc[i];
                    *cp = 0.0; // the slot belongs to a panel Lv + 1 groups ahead from now on
                    nact += (fl[i] == fl[i]) ? 1 : 0;
                }
            }
        }
        HBG_ACC(0);
        HBG_DBG(1); // the group's dots are in hand
        const int32_t *gblk0 = v.gram + (size_t)gp0 * (pv.Lg + 1) * PP; // block l = 0 of the group's first panel
        const size_t pstep = (size_t)(pv.Lg + 2) * PP;                   // block l of panel p -> block l + 1 of panel p + 1
        // panels ahead that are owed the corrections by THIS workgroup (with k_fwd beside it: the next group's only)
        const int nfw = max(0, min(pv.fcorr ? D : pv.Lv * D, np - (gp0 + D)));
        const bool have_fw = nfw > 0;
        int nround = 0, nmv_grp = 0;
        (void)nround; (void)nmv_grp;
        bool hbg_first = true;
        (void)hbg_first;
        int pos_lo = 0;      // markers of the group before this position are decided
        unsigned forced = 0; // (bit i: marker i * P + t was pushed over its threshold by a move of a rolled-back round)
        double absd_grp = 0.0;
        for (;;) {
            // ---- (2) the round's candidates, ranked in marker order ----
            // (the masks are integers, bit i = marker i * P + t: written with && and || over the eight markers the compiler kept sixteen lane masks
            // in scalar register pairs across the round — more than there are — and moved them in and out of vector lanes all the way)
            unsigned long long rkp = 0; // rank of marker i * P + t among its wave's candidates of panel i: 8 bits per i
            int cntv = 0;               // lane i: candidates of panel i in this wave
            const int pl = pos_lo >> lgP; // (< Dg: the loop ends when pos_lo reaches the group's end)
            const unsigned lom = ((0x1feu << pl) & 0xffu) | ((t >= (pos_lo & (P - 1))) ? 1u << pl : 0u); // marker at or after pos_lo
            unsigned gem = 0;
#pragma unroll
            for (int i = 0; i < HBG_DM; i++) // (thc[]: NaN for a filtered-out marker and for a panel past the group's end — never a candidate, and never forced —, -1 for a hot one — always)
                gem |= (r0[i] * r0[i] >= thc[i]) ? 1u << i : 0u;
            const unsigned iscm = (gem | forced) & lom;
#pragma unroll
            for (int i = 0; i < HBG_DM; i++) {
                const unsigned long long cm = __ballot(((iscm >> i) & 1u) != 0u);
                rkp |= (unsigned long long)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u)) << (8 * i);
                {   // lane i takes the count: one v_writelane, no lane mask (this clang has no builtin for it)
                    const int pc = __builtin_amdgcn_readfirstlane(__popcll(cm));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(pc), "n"(i));
                }
            }
            if (lane < HBG_DM) wcnt[lane * 8 + wave] = cntv; // (one write per wave)
            if (t == 0) misc[1] = Dg * P;
            __syncthreads(); // B1
            HBG_DBG(2);
            if (misc[2]) { ok = false; break; }
            int total, myscan;
            {   // exclusive scan of the (panel, wave) counts in every wave: lane = panel * 8 + wave
                const int cnt = wcnt[lane];
                const int inc = wave_scan_incl(cnt);
                total = __builtin_amdgcn_readlane(inc, 63);
                myscan = inc - cnt;
            }
            HBG_ACC(1);
            if (hbg_first) { HBG_CNT(12, total); hbg_first = false; }
            if (total == 0) break; // nobody (left) in the group can move
            const int ncr = min(total, 64);
            unsigned inrm = 0;
            unsigned long long rk = 0; // rank in the round of marker i * P + t, 8 bits per i (valid where inrm has the bit)
            // (2a) the candidates' ranks and positions (a thread almost never holds more than one candidate)
            for (unsigned left = iscm; __any(left != 0u);) {
                const bool mine = left != 0u;
                const int i = mine ? __ffs((int)left) - 1 : 0;
                left &= left - 1u;
                const int basew = __shfl(myscan, i * 8 + wave, 64); // candidates before this (panel, wave)
                const int rank = basew + (int)((rkp >> (8 * i)) & 0xffull);
                if (mine && rank == 64) misc[1] = i * P + t;
                if (mine && rank < 64) {
                    cs_pos[rank] = i * P + t;
                    rk |= (unsigned long long)rank << (8 * i);
                    inrm |= 1u << i;
                }
            }
            HBG_DBGW(21);
            __syncthreads(); // B2
            HBG_DBGW(22);
            HBG_ACC(2);
            const int pos_hi = misc[1];
            // (2b) + (3) ONE round trip for both (round 5: they used to be two, the gather's loads issued only after the exact data had come back —
            // and, above 8 candidates, in two dependent batches): the Gram entries among the round's candidates, cg[k][c] = x_k . x_c for k < c
            // (zero elsewhere) — every load unconditional on a clamped address, eight per thread at P = 512 —, then the candidates' exact
            // per-marker data; everything is stored when it has all arrived.
            int gq[8];
            const bool wide = P == 512;
            auto gather_one = [&](int q) {
                const int idx = t + q * 512, k = idx >> 6, c = idx & 63;
                const bool valid = k < c && c < ncr; // (k < c < ncr <= 64 implies idx < ncr * 64)
                const int a = cs_pos[valid ? k : 0], b = cs_pos[valid ? c : 0];
                const int pa = a >> lgP, ia = a & (P - 1), pb = b >> lgP, ib = b & (P - 1);
                const size_t off = valid ? ((size_t)pb * (pv.Lg + 1) + (pb - pa)) * PP + (size_t)ia * P + ib : (size_t)t;
                return gblk0[off];
            };
#pragma unroll
            for (int q = 0; q < 8; q++) gq[q] = 0;
            const int nq = (ncr + 7) >> 3; // (uniform) blocks of eight rows of cg that hold anything: rows k >= ncr are never read
            if (wide) {
                gq[0] = gather_one(0); // rows k < 8: all there is up to eight candidates (the usual round)
#pragma unroll
                for (int q = 1; q < 8; q++)
                    if (q < nq) gq[q] = gather_one(q); // (round 6: seven unconditional batches above eight candidates cost BayesR's 17-candidate rounds five batches of index arithmetic for nothing)
            }
            HBG_DBGW(23);
            for (unsigned left = inrm; __any(left != 0u);) {
                const bool mine = left != 0u;
                const int i = mine ? __ffs((int)left) - 1 : 0;
                left &= left - 1u;
                double r0i = r0[0];
#pragma unroll
                for (int x = 1; x < HBG_DM; x++) r0i = (i == x) ? r0[x] : r0i;
                const int rank = (int)((rk >> (8 * i)) & 0xffull);
                if (mine) {
                    // (for an effect of zero fma(xx, 0, rhs) is rhs exactly: the product is taken unconditionally — behind "gold != 0" the load
                    // of x'x was a second round trip)
                    const size_t j = (size_t)(gp0 + i) * P + t;
                    const double gold = v.g[j], xx = v.xpx[j];
                    double thc[K1], ivc[K1], szc[K1];
#pragma unroll
                    for (int c = 0; c < K1; c++) {
                        thc[c] = v.thr[(size_t)c * v.m_pad + j];
                        ivc[c] = v.invv[(size_t)c * v.m_pad + j];
                        szc[c] = v.sdz[(size_t)c * v.m_pad + j];
                    }
                    int gac = 0, cmc = 0;
                    if constexpr (G16 || CERT) gac = v.ga[j];
                    if constexpr (CERT) cmc = v.gcmax[j];
                    cs_d[rank] = fma(xx, gold, r0i);
                    cs_d[64 + rank] = gold;
#pragma unroll
                    for (int c = 0; c < K1; c++) {
                        cs_d[(2 + c) * 64 + rank] = thc[c];
                        cs_d[(2 + K1 + c) * 64 + rank] = ivc[c];
                        cs_d[(2 + 2 * K1 + c) * 64 + rank] = szc[c];
                    }
                    if constexpr (G16 || CERT) cs_ga[rank] = gac;
                    if constexpr (CERT) cs_cm[rank] = cmc;
                }
            }
            if (wide) {
                if (t < ncr * 64) cg[t] = ((t >> 6) < (t & 63) && (t & 63) < ncr) ? gq[0] : 0;
#pragma unroll
                for (int q = 1; q < 8; q++) {
                    if (q < nq) {
                        const int idx = t + q * 512, k = idx >> 6, c = idx & 63;
                        if (idx < ncr * 64) cg[idx] = (k < c && c < ncr) ? gq[q] : 0;
                    }
                }
            } else {
                for (int idx = t; idx < ncr * 64; idx += P) {
                    const int k = idx >> 6, c = idx & 63;
                    int gval = 0;
                    if (k < c && c < ncr) {
                        const int a = cs_pos[k], b = cs_pos[c];
                        const int pa = a >> lgP, ia = a & (P - 1), pb = b >> lgP, ib = b & (P - 1);
                        gval = v.gram[((size_t)(gp0 + pb) * (pv.Lg + 1) + (pb - pa)) * PP + (size_t)ia * P + ib];
                    }
                    cg[idx] = gval;
                }
            }
            HBG_DBGW(25);
            __syncthreads(); // B3
            HBG_DBG(3);
            HBG_ACC(3);
            bool stage_wait = false; // (per wave) this wave has pieces in flight
            if constexpr (CERT) {    // (the next group's records: by the waves that would otherwise stand at B4 while wave 0 walks the serial chain)
                if (!staged) {
                    if (wave != 0 && gp0 + D < np) { stage_next(gp0 + D, 1); stage_wait = true; }
                    staged = true;
                }
            }

            // ---- (4) the exact serial chain over the round's candidates: wave 0, one candidate per lane, in marker order ----
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
                    if (K1 == 1 && model == 5 && fabs(gn) < 1e-6) gn = 1e-6; // src/Bayes.cpp:728
                };
                // (Round 6, measured and dropped — profiles/r06_spec_serial.txt: the loop on SPECULATED classes, a step being fma - subtract - readlane - fma
                // and the decision checked afterwards. The genotypes are not centred: one move shifts every later right-hand side by n mean_k mean_c
                // times its change — a large part of the distance to a threshold — so a candidate's class at the start of a block of four steps is wrong
                // for one block in 2.4 (from the opening value: worse), and the re-runs cost more than the 270-cycle steps they replace.)
                int rnext = cg[lane]; // row k of cg, one step ahead
                HBG_MARK(21); // (cycles into the serial phase: the candidates' data are in registers)
                for (int k = 0; k < ncr; k++) {
                    const int rcur = rnext;
                    rnext = cg[min(k + 1, ncr - 1) * 64 + lane];
                    int cls;
                    double gn;
                    decide(crhs, cls, gn);
                    const double dk = readlane_f64(gn - cgold, k);
                    crhs = fma(-(double)rcur, dk, crhs); // (row k is zero at and before lane k; a marker that stays adds an exact zero)
                }
                HBG_MARK(22); // (... the serial loop is done)
                int cls;
                double gn;
                decide(crhs, cls, gn); // lane k's rhs was not touched after its own step: its outcome, for all lanes at once
                const double dmine = lv ? gn - cgold : 0.0;
                const unsigned long long moved = __ballot(lv && dmine != 0.0);
                if (lv && dmine != 0.0) {
                    const int pos = __popcll(moved & lt);
                    ev_pos[pos] = cp;
                    ev_del[pos] = dmine;
                    if constexpr (G16) ev_ga[pos] = cs_ga[lane];
                }
                res_c[lane] = cls;
                res_g[lane] = gn;
                if (lane == 0) misc[0] = __popcll(moved);
                if constexpr (CERT) { // what the certificate needs: prefix sums of ga * change in marker order, their largest magnitude, and E
                    double w = lv ? (double)cs_ga[lane] * dmine : 0.0;
                    double e = lv ? (double)cs_cm[lane] * fabs(dmine) : 0.0;
                    double am;
                    if (ncr <= 16) { // (the usual case) inclusive scans inside the first row of 16 lanes, by DPP: sum of w, running max of |prefix|, sum of e
#define HBG_SCAN_STEP(N)                                                                                                           \
                        {                                                                                                          \
                            const double uw = dpp_row_shr_f64<N>(w), ue = dpp_row_shr_f64<N>(e);                                   \
                            w += uw;                                                                                               \
                            e += ue;                                                                                               \
                        }
                        HBG_SCAN_STEP(1) HBG_SCAN_STEP(2) HBG_SCAN_STEP(4) HBG_SCAN_STEP(8)
#undef HBG_SCAN_STEP
                        am = fabs(w);
#define HBG_MAX_STEP(N) am = fmax(am, dpp_row_shr_f64<N>(am));
                        HBG_MAX_STEP(1) HBG_MAX_STEP(2) HBG_MAX_STEP(4) HBG_MAX_STEP(8)
#undef HBG_MAX_STEP
                        // (lane ncr - 1 of row 0 holds the totals; lanes of the other rows hold zeros)
                        e = readlane_f64(e, ncr - 1);
                        am = readlane_f64(am, ncr - 1);
                    } else { // (lanes at and past ncr hold zeros: the scans run over the whole wave)
                        w = wave_scan_incl_f64(w);
                        am = readlane_f64(wave_scan_max_f64(fabs(w)), 63);
                        e = readlane_f64(wave_scan_incl_f64(e), 63);
                    }
                    if (lane < ncr) spre[lane + 1] = w;
                    if (lane == 0) { spre[0] = 0.0; spre[66] = e; spre[67] = am; }
                }
                HBG_MARK(23); // (... wave 0 is at the barrier)
            }
            if constexpr (CERT) {
                if (stage_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the pieces have landed: these waves had nothing else to do)
            }
            __syncthreads(); // B4
            HBG_DBG(4);
            HBG_ACC(4);
            const int nmoves = misc[0];
            // ---- (4b) CERT: decide the passed-over markers from the rank-one part of the moves and the bound on the rest ----
            bool need_full = true; // (uniform) the round goes through the full fold and the exact check
            if constexpr (CERT) {
                if (nmoves > 0 && pos_hi >= Dg * P) {
                    const double E = spre[66] * (1.0 + 1e-9), Amax = spre[67] * (1.0 + 1e-9);
                    // stage 1 (registers only): |rhs| + |gB| max|A| + E below the square root of the threshold — true for all but the few per cent
                    // of the markers that the shift could reach at all
                    unsigned st2 = 0, open = 0, close_by = 0;
#pragma unroll
                    for (int i = 0; i < HBG_DM; i++) {
                        const double b1 = fma(fabs((double)gBi[i]), Amax, fabs(r0[i]) + E);
                        st2 |= (b1 >= (double)sqf[i]) ? 1u << i : 0u; // (sqf NaN — the inactive and the hot markers, the panels past the group's end: false)
                    }
                    st2 &= lom & ~iscm; // (every marker is before pos_hi: the round reaches the group's end)
                    // stage 2: the shift this marker really sees, from the prefix sum at the number of candidates before it
                    for (unsigned left = st2; __any(left != 0u); left &= left - 1u) {
                        const bool mine = left != 0u;
                        const int i = mine ? __ffs((int)left) - 1 : 0;
                        double r0i = r0[0];
                        float sqi = sqf[0];
                        int Bi = gBi[0];
#pragma unroll
                        for (int x = 1; x < HBG_DM; x++) {
                            r0i = (i == x) ? r0[x] : r0i;
                            sqi = (i == x) ? sqf[x] : sqi;
                            Bi = (i == x) ? gBi[x] : Bi;
                        }
                        const int before = min(64, __shfl(myscan, i * 8 + wave, 64) + (int)((rkp >> (8 * i)) & 0xffull)); // the round's candidates before this marker
                        const double hi = (fabs(fma(-(double)Bi, spre[before], r0i)) + E) * (1.0 + 1e-9);
                        if (mine && hi >= (double)sqi) open |= 1u << i; // not proven to stay
                        // (within HBG_CERT_MARGIN of its threshold: taken along IF the round is repeated anyway — the new candidate's move may push
                        // it over, which would cost one more repetition; an extra candidate is decided exactly and costs a lane)
                        if (mine && hi >= HBG_CERT_MARGIN * (double)sqi) close_by |= 1u << i;
                    }
                    {
                        const unsigned long long bo = __ballot(open != 0u);
                        if (lane == 0) misc[24 + wave] = bo != 0ull;
                    }
                    __syncthreads(); // B4b
                    int any = 0;
                    {
                        int w8[8];
                        hb_read8(misc + 24, w8);
#pragma unroll
                        for (int w = 0; w < 8; w++) any |= w8[w];
                    }
                    HBG_ACC(9);
                    if (any) { // a marker that may have crossed (nearly always: has) joins the candidates and the round is repeated — nothing was fetched for it
                        forced |= open | close_by;
                        HBG_CNT(15, 1);
                        continue;
                    }
                    need_full = false;
                }
            }
            // ---- (5) the round's moves onto the later markers of the group AND forward, all rows of up to HBG_CH moves in one
            // trip (a lone compute unit's loads take microseconds beside the streaming mat-vec: the number of dependent trips is
            // what a group costs). The forward contributions are summed in registers and reach the correction ring only when the
            // round is committed. ----
            double rnew[HBG_DM], fw[HBG_FW];
#pragma unroll
            for (int i = 0; i < HBG_DM; i++) rnew[i] = r0[i];
#pragma unroll
            for (int x = 0; x < HBG_FW; x++) fw[x] = 0.0;
            // (row addresses are "wave-uniform pointer"[t]: scalar base + one vector offset, no 64-bit vector arithmetic)
            using gram_t = typename std::conditional<G16, int16_t, int32_t>::type;
            const gram_t *gbase = G16 ? reinterpret_cast<const gram_t *>(v.gram16) + (size_t)gp0 * (pv.Lg + 1) * PP : reinterpret_cast<const gram_t *>(gblk0);
            if (CERT && !need_full) {
                // every passed-over marker is proven to stay and the round reaches the group's end: only the next group's panels need the moves
                // (rows requested together: at most 63 loads per lane; round 6: up to 15 moves per trip where a move has few rows — BayesR's 16 moves per
                // two-panel group were two dependent trips of ~4 500 cycles each beside the streaming mat-vec)
                constexpr int CH2 = (63 / HBG_FW) < HBG_CH2MAX ? (63 / HBG_FW) : HBG_CH2MAX;
                if (have_fw) {
#pragma unroll 1
                    for (int e0 = 0; e0 < nmoves; e0 += CH2) {
                        int gf2[CH2][HBG_FW];
                        double dl2[CH2];
#pragma unroll
                        for (int f = 0; f < CH2; f++) {
                            const int e = min(e0 + f, nmoves - 1);
                            const int a = __builtin_amdgcn_readfirstlane(ev_pos[e]);
                            const int pa = a >> lgP, ia = a & (P - 1);
                            dl2[f] = (e0 + f < nmoves) ? ev_del[e] : 0.0;
                            const int32_t *row = gblk0 + (size_t)ia * P + (size_t)D * pstep - (size_t)pa * PP; // first panel ahead
#pragma unroll
                            for (int x = 0; x < HBG_FW; x++) {
                                gf2[f][x] = row[t];
                                row += (x + 1 < nfw) ? pstep : 0;
                            }
                        }
#pragma unroll
                        for (int f = 0; f < CH2; f++)
#pragma unroll
                            for (int x = 0; x < HBG_FW; x++) fw[x] = (x < nfw) ? fma((double)gf2[f][x], dl2[f], fw[x]) : fw[x];
                    }
                }
            } else {
#pragma unroll 1
            for (int e0 = 0; e0 < nmoves; e0 += HBG_CH) {
                int gv[HBG_CH][HBG_DM], gf[HBG_CH][HBG_FW];
                int pae[HBG_CH], iae[HBG_CH], gaf[HBG_CH], latm[HBG_CH];
                double dl[HBG_CH];
#pragma unroll
                for (int f = 0; f < HBG_CH; f++) {
                    const int e = min(e0 + f, nmoves - 1);
                    const int a = __builtin_amdgcn_readfirstlane(ev_pos[e]);
                    pae[f] = a >> lgP;
                    iae[f] = a & (P - 1);
                    dl[f] = (e0 + f < nmoves) ? ev_del[e] : 0.0;
                    gaf[f] = G16 ? __builtin_amdgcn_readfirstlane(ev_ga[e]) : 0;
                    // bit i: marker (i, t) comes after the mover in the order (an integer mask, used through a one-bit signed field extract: as
                    // a chain of && and || the compiler built it from scalar branches and sixteen lane masks per trip)
                    latm[f] = (int)(((0x1feu << pae[f]) & 0xffu) | ((t > iae[f]) ? 1u << pae[f] : 0u));
                }
#pragma unroll
                for (int f = 0; f < HBG_CH; f++) {
                    // panel i of the group meets the mover (panel pae) in its block l = i - pae, at
                    // gblk0 + (i (Lg + 1) + i - pae) PP = (gblk0 - pae PP) + i (Lg + 2) PP; the panels ahead continue the same walk.
                    // Branch-free: a panel before the mover's or past the group's end reads a neighbouring valid row instead (its
                    // value is not used) — behind a branch every load would be waited for on the spot.
                    const gram_t *row = gbase + (size_t)iae[f] * P + (size_t)pae[f] * (pstep - PP); // i = pae
#pragma unroll
                    for (int i = 0; i < HBG_DM; i++) {
                        gv[f][i] = row[t];
                        row += (i >= pae[f] && i + 1 < Dg) ? pstep : 0; // (scalar select)
                    }
                    // (no panel ahead at the end of the sweep: the loads stay, on an address that exists; their values are not used)
                    row = have_fw ? gbase + (size_t)iae[f] * P + (size_t)D * pstep - (size_t)pae[f] * PP : gbase; // first panel ahead
#pragma unroll
                    for (int x = 0; x < HBG_FW; x++) {
                        gf[f][x] = row[t];
                        row += (x + 1 < nfw) ? pstep : 0;
                    }
                }
#pragma unroll
                for (int f = 0; f < HBG_CH; f++) {
#pragma unroll
                    for (int i = 0; i < HBG_DM; i++) {
                        // marker (i, t) takes the move of (pae, iae) if it comes later in the order: the others add an exact zero (a panel past
                        // the group's end takes whatever its stand-in row holds — its right-hand side is never looked at)
                        // (G16: G[k][j] = g16[k][j] + ga[k] gB[j], an exact integer — the fold's arithmetic is the int32 band's, bit for bit)
                        const int gm = (G16 ? gv[f][i] + gaf[f] * gBi[G16 ? i : 0] : gv[f][i]) & __builtin_amdgcn_sbfe(latm[f], i, 1);
                        rnew[i] = fma(-(double)gm, dl[f], rnew[i]);
                    }
#pragma unroll
                    for (int x = 0; x < HBG_FW; x++) fw[x] = (x < nfw) ? fma((double)(G16 ? gf[f][x] + gaf[f] * gBf[G16 ? x : 0] : gf[f][x]), dl[f], fw[x]) : fw[x];
                }
            }
            }
            // ---- (6) did every marker the round passed over really stay below its threshold? ----
            unsigned violm = 0;
            bool anyv = false;
            if (need_full) {
#pragma unroll
                for (int i = 0; i < HBG_DM; i++) violm |= (rnew[i] * rnew[i] >= (double)fl[i]) ? 1u << i : 0u; // (NaN filter, also past the group's end: false)
                const int ph = pos_hi >> lgP;
                const unsigned him = ((1u << ph) - 1u) | ((t < (pos_hi & (P - 1))) ? 1u << ph : 0u); // marker before pos_hi
                violm &= lom & him & ~iscm;
                {
                    const unsigned long long vm = __ballot(violm != 0u);
                    if (lane == 0) misc[8 + wave] = vm != 0ull;
                }
                __syncthreads(); // B5
                int w8[8];
                hb_read8(misc + 8, w8);
#pragma unroll
                for (int w = 0; w < 8; w++) anyv |= w8[w] != 0;
            }
            HBG_ACC(5);
            if (anyv) { // roll the round back: the markers that crossed join the candidates
                forced |= violm;
                if (t == 0) redoacc++;
                HBG_CNT(14, 1);
                continue;
            }
            // ---- (7) commit the round ----
#pragma unroll
            for (int i = 0; i < HBG_DM; i++) r0[i] = rnew[i];
            for (unsigned left = inrm; left != 0u; left &= left - 1u) { // (divergent: a handful of threads)
                const int i = __ffs((int)left) - 1;
                const int rnk = (int)((rk >> (8 * i)) & 0xffull);
                const size_t j = (size_t)(gp0 + i) * P + t;
                const int cls_f = res_c[rnk];
                const double g_f = res_g[rnk];
                v.g[j] = g_f;
                v.tracker[j] = (uint8_t)cls_f;
                if (count_pip && cls_f != 0) {
                    __hip_atomic_fetch_add(&v.nzrate[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v.wind) v.wflag[v.wind[j] - 1u] = 1;
                }
                if (store && g_f != 0.0) {
                    unsafeAtomicAdd(&v.alpha_sum[j], g_f);
                    unsafeAtomicAdd(&v.alpha_sq[j], g_f * g_f);
                }
                if (cls_f > 0) wacc += (model == 6) ? g_f * g_f / pin->fold[cls_f] : g_f * g_f;
#pragma unroll
                for (int c = 1; c <= K1; c++) cacc[c] += cls_f == c ? 1 : 0;
            }
            evacc += nmoves;
            // the moves, appended to their panels' lists (write-through: the update rows of this group read them)
            if (wave == S - 1 && nmoves > 0) {
                const bool have = lane < nmoves;
                const int a = have ? ev_pos[lane] : 0;
                const double dlt = have ? ev_del[lane] : 0.0;
                const int pa = a >> lgP, ia = a & (P - 1);
                int myidx = 0, addme = 0;
#pragma unroll
                for (int i = 0; i < HBG_DM; i++) {
                    if (i < Dg) {
                        const unsigned long long mi = __ballot(have && pa == i);
                        if (have && pa == i) myidx = misc[16 + i] + __popcll(mi & lt);
                        if (lane == i) addme = __popcll(mi);
                    }
                }
                if (have) {
                    st_sc1(&v.ev_idx[(size_t)(gp0 + pa) * P + myidx], ia);
                    st_sc1(&v.ev_delta[(size_t)(gp0 + pa) * P + myidx], dlt);
                }
                if (lane < HBG_DM) misc[16 + lane] += addme;
                absd_grp += wave_sum(fabs(dlt));
            }
            HBG_ACC(6);
            // ---- (8) ... and the forward sums into the corrections owed to the panels of the next Lv groups ----
            if (nmoves > 0) {
                int sl = gslot + D;
                sl = sl >= R ? sl - R : sl;
#pragma unroll
                for (int x = 0; x < HBG_FW; x++) {
                    if (x < nfw) corr[(size_t)sl * P + t] += fw[x]; // this thread's own word: no synchronisation needed
                    sl = (sl + 1 == R) ? 0 : sl + 1;
                }
            }
            HBG_ACC(7);
            nmv_grp += nmoves;
            nround++;
            pos_lo = pos_hi;
            if (pos_lo >= Dg * P) break;
        }
        if (!ok) break;
        if constexpr (CERT) { // (a group without a single candidate never got to its serial pass)
            if (!staged) {
                if (gp0 + D < np) stage_next(gp0 + D, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        // ---- end of the group: counts, bound on max |yadj|, chain_done (the update rows of this group wait for it) ----
        if (wave == S - 1) {
            if (lane < Dg) {
                const int c = misc[16 + lane];
                st_sc1(&v.ev_count[(size_t)(gp0 + lane) * HB_EVS], c); // (zero included: the update rows poll the counts themselves)
                misc[16 + lane] = 0;
            }
            if (v.mb) {
                mbr = fma(v.xabs, absd_grp, mbr);
                if (lane == 0) st_sc1(&v.mb[(size_t)(1 + gcount) * HB_MBS], mbr);
            }
            // (no drain before the flag any more: every consumer of the counts, the bound and the move lists validates the words
            // themselves — waiting for the acknowledgement of stores to lines that 196 update blocks poll cost this wave, and through
            // the next group's first barrier the whole workgroup, ~25 000 cycles per group: profiles/r04_group_timeline_*.txt)
            if (lane == 0) st_flag(pv.flags + HB_FLAG_CHAIN_DONE, (unsigned)(gp0 + Dg));
        }
        HBG_DBG(9);
        HBG_ACC(8);
        HBG_CNT(10, nmv_grp);
        HBG_CNT(13, nround);
        HBG_END();
        gcount++;
        gslot += D;
        gslot = gslot >= R ? gslot - R : gslot;
    }
    // ---- sweep totals for the hyper-parameter draws ----
    __syncthreads();
    const double wsum = block_sum(wacc, red);
    if (t == 0) {
        v.acc[HB_ACC_SUMG2] += wsum;
        v.acc[HB_ACC_EVENTS] += (double)evacc;
        v.acc[HB_ACC_REDO] += (double)redoacc;
    }
    int nin = 0;
#pragma unroll
    for (int c = 1; c <= K1; c++) {
        nin += cacc[c];
        const double cs = block_sum((double)cacc[c], red);
        if (t == 0 && c < HB_MAX_FOLD) v.acc[HB_ACC_COUNT0 + c] += cs;
    }
    {
        const double cs = block_sum((double)(nact - nin), red); // class 0: the polymorphic markers that are not in the model
        if (t == 0) v.acc[HB_ACC_COUNT0] += cs;
    }
    if (t == 0 && !ok) { // aborted: the host must see it (fetch_acc checks the flag), then release every waiter
        st_flag(pv.flags + HB_FLAG_ABORT, 1u);
        st_flag(pv.flags + HB_FLAG_CHAIN_DONE, 0x7fffffffu);
    }
}

template <int K1, int HBG_DM, int HBG_FW, int HBG_CH, bool G16 = false, bool CERT = false, bool FRESH = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chain_group(const hb_sweep_in *__restrict__ pin, chain_view v,
                                                                                                 persist_view pv)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    chain_group_body<K1, HBG_DM, HBG_FW, HBG_CH, G16, CERT, FRESH>(pin, v, pv, smem);
}

// ---------------------------------------------------------------------------------------------
// k_fwd: the forward fold of the groups AFTER the next one, on a second compute unit.
// A move of group g owes corrections to the dots of every panel whose mat-vec ran without it: the D panels of each of the groups
// g + 1 .. g + Lv. k_chain_group needs those of g + 1 at once — it opens group g + 1 as soon as it has closed g — and keeps
// them; the others have at least a whole group of slack, and what a fold trip costs is one compute unit's memory pipeline
// (DESIGN §6), so they go to another one: this workgroup waits for chain_done(g), reads the group's published move lists, sums
// G[move][q] delta over the moves for every marker q of the groups g + 2 .. g + Lv (thread = marker of each of those panels,
// HBF_CH moves per trip, the chain's row walk continued) and writes to fcorr[] what group g + 2 is owed in all — its own sums
// plus what the groups before g left for it, carried in registers — sentinel-prefilled like the dots, so the chain polls both in
// the same trip. Every entry of group g + 2 is written, moves or not. One fma per move, in move order, like the chain's own
// forward sums: the same chain in exact arithmetic (tests: draw for draw).
// HBF_D = D panels per group (slot y of fs[] is panel y counted from the first panel of group g + 2), HBF_G = Lv - 1 groups.
// ---------------------------------------------------------------------------------------------
// (round 6: the body as a device function on a caller-supplied LDS region, so that it can also run as the SECOND workgroup of the chain's own kernel —
// k_chain_group_fwd below: one graph branch and one hardware queue instead of two)
template <int HBF_D, int HBF_G>
constexpr int hbf_lds_bytes(bool g16) { return HBF_D * 512 * 8 + HBF_D * 512 * 4 + (g16 ? HBF_D * 512 * 4 : 16) + (HBF_D + 1) * 4 + 16; }
template <int HBF_D, int HBF_G, int HBF_CH, bool G16 = false, bool FRESH = false>
__device__ __forceinline__ void fwd_body(const chain_view &v, const persist_view &pv, char *lds)
{
    constexpr int NF = HBF_D * HBF_G;
    double *s_del = reinterpret_cast<double *>(lds);                 // [HBF_D * 512] the group's moves: changes of effect
    int *s_pos = reinterpret_cast<int *>(s_del + HBF_D * 512);       // [HBF_D * 512] ... panel * P + marker
    int *s_ga = s_pos + HBF_D * 512;                                 // [G16 ? HBF_D * 512 : 4] G16: ga[] of the movers
    int *s_cnt = s_ga + (G16 ? HBF_D * 512 : 4);                     // [HBF_D + 1]
    int &s_ok = s_cnt[HBF_D + 1];
    const int P = v.P, t = threadIdx.x, lgP = 31 - __clz(P);
    const int D = pv.D, np = pv.npanels, G = pv.Lv - 1;
    const size_t PP = (size_t)P * P, pstep = (size_t)(pv.Lg + 2) * PP;
    double fs[NF]; // [k * HBF_D + x]: owed to panel x of group g + 2 + k by the groups up to g
    if (D != HBF_D || G != HBF_G) return; // (the host launches the matching shape)
#pragma unroll
    for (int x = 0; x < NF; x++) fs[x] = 0.0;
    for (int gp0 = pv.p0; gp0 < np; gp0 += D) {
        const int Dg = min(D, np - gp0);
        const int nfar = max(0, min(G * D, np - (gp0 + 2 * D))); // panels of the groups g + 2 .. g + Lv
        if (nfar == 0) break;
        if (t == 0) s_ok = wait_ge(pv.flags, HB_FLAG_CHAIN_DONE, (unsigned)(gp0 + Dg)) ? 1 : 0;
        __syncthreads();
        if (!s_ok) return;
        // (round 4: chain_done only paces this workgroup — the chain no longer drains its stores before raising it. Counts and list entries
        // are pre-filled by the sweep with a pattern no value has and validated here, like the update rows do: the data is the flag)
        if (t <= HBF_D) { // exclusive scan of the panels' move counts
            int a = 0;
            for (int i = 0; i < t && i < Dg; i++) {
                int c = ld_sc1(&v.ev_count[(size_t)(gp0 + i) * HB_EVS]);
                const unsigned long long t0 = wall_clock64();
                unsigned looks = 0;
                while (c < 0 && !ld_flag(pv.flags + HB_FLAG_ABORT) && wall_clock64() - t0 < HB_TIMEOUT_TICKS) {
                    __builtin_amdgcn_s_sleep(2);
                    c = ld_poll(&v.ev_count[(size_t)(gp0 + i) * HB_EVS], looks++, FRESH ? 4u : 0u);
                }
                if (c < 0) { st_flag(pv.flags + HB_FLAG_ABORT, 1u); c = 0; s_ok = 0; }
                a += c;
            }
            s_cnt[t] = a;
        }
        __syncthreads();
        if (!s_ok) return;
        const int nev = s_cnt[min(Dg, HBF_D)];
        for (int i = 0; i < Dg; i++) {
            const int b = s_cnt[i], c = s_cnt[i + 1] - b;
            for (int k = t; k < c; k += P) {
                const size_t src = (size_t)(gp0 + i) * P + k;
                int ix = ld_sc1(&v.ev_idx[src]);
                double dl = ld_sc1(&v.ev_delta[src]);
                const unsigned long long t0 = wall_clock64();
                unsigned looks = 0;
                while ((ix < 0 || __double_as_longlong(dl) == -1ll) && !ld_flag(pv.flags + HB_FLAG_ABORT) && wall_clock64() - t0 < HB_TIMEOUT_TICKS) {
                    __builtin_amdgcn_s_sleep(2);
                    ix = ld_poll(&v.ev_idx[src], looks, FRESH ? 4u : 0u);
                    dl = ld_poll(&v.ev_delta[src], looks, FRESH ? 4u : 0u);
                    looks++;
                }
                if (ix < 0 || __double_as_longlong(dl) == -1ll) { st_flag(pv.flags + HB_FLAG_ABORT, 1u); ix = 0; dl = 0.0; }
                s_pos[b + k] = i * P + ix;
                s_del[b + k] = dl;
                if constexpr (G16) s_ga[b + k] = v.ga[(size_t)(gp0 + i) * P + ix];
            }
        }
        __syncthreads();
        using gram_t = typename std::conditional<G16, int16_t, int32_t>::type;
        const gram_t *gblk0 = (G16 ? reinterpret_cast<const gram_t *>(v.gram16) : reinterpret_cast<const gram_t *>(v.gram)) + (size_t)gp0 * (pv.Lg + 1) * PP;
        int gBy[G16 ? NF : 1]; // G16: gB of this thread's marker in each far panel
        (void)gBy;
        if constexpr (G16) {
#pragma unroll
            for (int y = 0; y < NF; y++) gBy[y] = v.gB[(size_t)min(gp0 + 2 * D + y, np - 1) * P + t];
        }
#pragma unroll 1
        for (int e0 = 0; e0 < nev; e0 += HBF_CH) {
            int gf[HBF_CH][NF], gaf[HBF_CH];
            double dl[HBF_CH];
#pragma unroll
            for (int f = 0; f < HBF_CH; f++) {
                const int e = min(e0 + f, nev - 1);
                const int a = __builtin_amdgcn_readfirstlane(s_pos[e]);
                const int pa = a >> lgP, ia = a & (P - 1);
                dl[f] = (e0 + f < nev) ? s_del[e] : 0.0;
                gaf[f] = G16 ? __builtin_amdgcn_readfirstlane(s_ga[G16 ? e : 0]) : 0;
                // panel y (counted from the first panel of group g + 2) meets the mover in block l = 2 D + y - pa: the chain's walk, 2 D panels on
                const gram_t *row = gblk0 + (size_t)ia * P + (size_t)(2 * D) * pstep - (size_t)pa * PP;
#pragma unroll
                for (int y = 0; y < NF; y++) {
                    gf[f][y] = row[t];
                    row += (y + 1 < nfar) ? pstep : 0; // (past the last panel: the same row again, its value is not used)
                }
            }
#pragma unroll
            for (int f = 0; f < HBF_CH; f++)
#pragma unroll
                for (int y = 0; y < NF; y++) fs[y] = (y < nfar) ? fma((double)(G16 ? gf[f][y] + gaf[f] * gBy[G16 ? y : 0] : gf[f][y]), dl[f], fs[y]) : fs[y];
        }
        // group g + 2 has now heard from every group that owes it: publish, and shift what the later ones have so far
#pragma unroll
        for (int x = 0; x < HBF_D; x++)
            if (x < nfar) st_sc1(&pv.fcorr[(size_t)(gp0 + 2 * D + x) * P + t], fs[x]);
#pragma unroll
        for (int k = 0; k + 1 < HBF_G; k++)
#pragma unroll
            for (int x = 0; x < HBF_D; x++) fs[k * HBF_D + x] = fs[(k + 1) * HBF_D + x];
#pragma unroll
        for (int x = 0; x < HBF_D; x++) fs[(HBF_G - 1) * HBF_D + x] = 0.0;
        __syncthreads(); // (the lists are rewritten for the next group)
    }
}

template <int HBF_D, int HBF_G, int HBF_CH, bool G16 = false, bool FRESH = false>
__global__ __launch_bounds__(512) void k_fwd(chain_view v, persist_view pv)
{
    __shared__ __attribute__((aligned(16))) char lds[hbf_lds_bytes<HBF_D, HBF_G>(G16)];
    fwd_body<HBF_D, HBF_G, HBF_CH, G16, FRESH>(v, pv, lds);
}

// ---------------------------------------------------------------------------------------------
// k_chain_group_fwd (round 6): the chain workgroup and k_fwd as the two workgroups of ONE kernel (each takes a compute unit: both ask for the whole LDS) —
// one graph branch and one hardware queue instead of two, which is what lets the overlapped launch stream (two tile streams + the update kernels) fit the
// four queues a process gets (profiles/r06_overlap.txt).
// ---------------------------------------------------------------------------------------------
template <int K1, int HBG_DM, int HBG_FW, int HBG_CH, bool CERT, int HBF_D, int HBF_G, int HBF_CH>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chain_group_fwd(const hb_sweep_in *__restrict__ pin, chain_view v,
                                                                                                     persist_view pv)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x == 0) chain_group_body<K1, HBG_DM, HBG_FW, HBG_CH, false, CERT>(pin, v, pv, smem);
    else fwd_body<HBF_D, HBF_G, HBF_CH, false>(v, pv, smem);
}
