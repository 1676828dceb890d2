// hb_chain_dense.hpp — the chain of the models in which EVERY marker moves every sweep (BayesRR / BayesA / BayesL), and the
// workgroups that fold its moves forward. Included by hb_kernels.hip (it uses that file's views and hand-off helpers).
//
// Why. For these models (reference src/Bayes.cpp:587-604 RR, :606-625 A, :719-741 L) there is nothing to decide: marker k's
// new effect is an affine function of its right-hand side, gn_k = rhs_k / v_k + sd_k z_k, and the order of the moves is the
// marker order. k_chain_persist's machinery — candidates, speculative rounds, violation checks, a row cache filled from a
// hot-list — is all overhead here (round 2/3: 500 000 cycles per panel of 512, 6.3 sweeps/s at n = 50k, m = 500k), and what a
// panel needs from memory, its own 1-MB Gram block and Lb more of the band, is three times what ONE compute unit can pull in the
// time (~41 GB/s). So:
//   * k_chain_dense — one persistent workgroup, thread = marker of the panel, wave w = sub-block w of 64 markers. A panel is
//     eight steps: wave s runs the serial pass of its 64 markers with the 64 x 64 diagonal block of the panel's Gram block in
//     REGISTERS (one column per lane, requested a panel ahead — the order is static, so everything is), the loop fully
//     unrolled, in units of the CHANGE of effect: sv_j = rhs_j / v_j + sd_j z_j - g_j is what marker j's change would be if
//     it were drawn now, marker k's change is sv_k as it stands at step k and moves the later markers' by -G[k][j] / v_j —
//     two v_readlane and one fma per marker on the dependent chain. Then the later waves take the 64 changes with the strip
//     G[64 s .. 64 s + 63][t] they requested two steps earlier (64 registers per thread, double-buffered) — the next
//     HBD_NEAR sub-blocks only: what sub-block s owes the sub-blocks more than HBD_NEAR after it comes back from
//     k_fold_dense through fcorr2[] (10 of a panel's 28 strips at HBD_NEAR = 3, 15 at 2, never touch the chain's compute
//     unit; the serial wave of step s looks at fcorr2[] a step before it needs it). Nothing is ever decided, gathered or
//     rolled back (BayesL's clamp of tiny effects, src/Bayes.cpp:728, is a select inside the loop).
//   * k_fold_dense — the band: panel q is owed  sum_l G_l[q]^T delta_{q-l}  by the panels whose moves its mat-vec has not
//     seen, a dense 512 x 512 product per band block — and, since the near window, the panel's OWN far sub-blocks (block 0
//     of its Gram blocks, rows of the sub-blocks more than HBD_NEAR before the target's). 8 (Lb + 1) small workgroups spread
//     over the chip (one per 64 columns of a target panel, its four waves a quarter of the rows each; Lb + 1 target panels are
//     open at any time: Lb ahead for their band, the chain's own for its far sub-blocks) take the chain's changes sub-block
//     by sub-block as they are published (dd[], sentinel-prefilled like the dots: no flag) and hand the finished sums to the
//     chain through fcorr[] (band, polled with the panel's dots) and fcorr2[] (own panel, polled by the serial wave). The
//     chain's compute unit never reads a band row.
// Same chain as k_chain / k_chain_persist in exact arithmetic: the moves in the same order; a change is computed in its own
// units instead of from the carried right-hand side and the band sums are added per row quarter, i.e. effects agree to the
// last bits' rounding (tests: draw for draw against the oracle at 1e-9, BayesL 1e-6 — tests/test_gpu_depth.py
// test_dense_chain_*, test_gpu_configs.py test_all_move_models_at_n50k_*, test_gpu_parity.py).
#pragma once

#define HBD_P 512
#ifndef HBD_NEAR
#define HBD_NEAR 2
#endif
// HBD_NEAR: a sub-block's changes are applied by the chain itself to the next HBD_NEAR sub-blocks of its panel; the farther ones get them from k_fold_dense (fcorr2[])
#define HBD_SENT(x) (__double_as_longlong(x) == -1ll)

template <bool LASSO>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_chain_dense(const hb_sweep_in *__restrict__ pin, chain_view v,
                                                                                                 persist_view pv, double *__restrict__ dd, const double *__restrict__ fcorr2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *dl = reinterpret_cast<double *>(smem); // [512] the panel's changes of effect, by marker
    double *red = dl + HBD_P;                      // [16]
    double *sabs = red + 16;                       // [2] sum |change| of the panel so far
    int *misc = reinterpret_cast<int *>(sabs + 2); // [0] moves of the panel so far, [2] abort
    constexpr int P = HBD_P;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int np = pv.npanels, D = pv.D;
    const size_t PP = (size_t)P * P, pblk = (size_t)(pv.Lg + 1) * PP;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int count_pip = pin->count_pip, store = pin->store;
    if (t < 8) misc[t] = 0;
    if (t < 2) sabs[t] = 0.0;
    if (t == 0) { // "the chain is resident": k_gate holds the first mat-vec launch back until then
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st_flag(pv.flags + HB_FLAG_XCC, (xcc & 15u) + 1u);
    }

    double wacc = 0.0, mbr = v.mb ? v.mb[0] : 0.0, absd_grp = 0.0;
    int nact = 0, evacc = 0, gcount = pv.p0 / D;
    bool ok = true;

    // per-marker data of the panel this wave handles next, and the diagonal block of its sub-block (column `lane`, rows 0..63)
    double c_invv, c_sdz, c_gold, c_xx, c_thr;
    int dg[64], bufA[64], bufB[64];
    auto load_coeffs = [&](int pp) {
        const size_t j = (size_t)pp * P + t;
        c_invv = v.invv[j];
        c_sdz = v.sdz[j];
        c_gold = v.g[j];
        c_xx = v.xpx[j];
        c_thr = v.thr[j];
    };
    auto load_dg = [&](int pp) {
        const int32_t *gp = v.gram + (size_t)pp * pblk + (size_t)(64 * wave) * P + 64 * wave + lane;
#pragma unroll
        for (int k = 0; k < 64; k++) dg[k] = gp[(size_t)k * P];
    };
    // strip rs of panel pp: rows 64 rs .. 64 rs + 63 of its Gram block, this thread's column
#define HBD_REQUEST(BUF, pp, rs)                                                                 \
    do {                                                                                         \
        const int32_t *gp_ = v.gram + (size_t)(pp) * pblk + (size_t)(64 * (rs)) * P + t;         \
        _Pragma("unroll") for (int k = 0; k < 64; k++) BUF[k] = gp_[(size_t)k * P];              \
    } while (0)
    load_coeffs(pv.p0);
    load_dg(pv.p0);
    if (wave > 0 && wave <= HBD_NEAR) HBD_REQUEST(bufA, pv.p0, 0);
#define HBD_PREP(w_)                                                                                               \
    do { /* the serial wave's diagonal block: row k only reaches the lanes after k */                                  \
        _Pragma("unroll") for (int k = 0; k < 64; k++) dg[k] = lane > k ? dg[k] : 0;                                   \
    } while (0)
    if (wave == 0) HBD_PREP(0);
    __syncthreads();

    for (int p = pv.p0; ok && p < np; p++) {
        const size_t j = (size_t)p * P + t;
        const bool group_end = ((p - pv.p0) % D == D - 1) || p == np - 1;
        if (p > pv.p0) {
            // every wave drains the stores of the previous panel (its moves, results) before this panel's first barrier, after which
            // the last wave raises chain_done for it — no wave ever waits for a store to reach memory on the critical path
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave == 7) { // (the others requested these a panel ahead, right after their serial pass)
                if (lane == 0 && ((p - 1 - pv.p0) % D == D - 1)) st_flag(pv.flags + HB_FLAG_CHAIN_DONE, (unsigned)p);
                load_coeffs(p);
                load_dg(p);
            }
            if (wave > 0 && wave <= HBD_NEAR) HBD_REQUEST(bufA, p, 0);
        }
        HB_STAMP(11);
        // ---- opening: the panel's dots and what the band owes it ----
        const bool use_fc = p > pv.p0 && pv.Lb > 0;
        double rhs;
        {
            const double *fcp = use_fc ? pv.fcorr : v.dsum;
            double dj = ld_sc1(&v.dsum[j]), fc = ld_sc1(&fcp[j]);
            bool bad = HBD_SENT(dj) || (use_fc && HBD_SENT(fc));
#if HB_STAMPS
            long long c12 = HBD_SENT(dj) ? 0 : 1;
#endif
            if (__any(bad)) {
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    dj = ld_sc1(&v.dsum[j]);
                    fc = ld_sc1(&fcp[j]);
#if HB_STAMPS
                    if (v.dbg && t == 0 && !HBD_SENT(dj) && c12 == 0) c12 = clock64();
#endif
                    bad = HBD_SENT(dj) || (use_fc && HBD_SENT(fc));
                    if (!__any(bad)) break;
                    const bool own = wall_clock64() - t0 > HB_TIMEOUT_TICKS;
                    if (ld_flag(pv.flags + HB_FLAG_ABORT) || own) {
                        if (lane == 0) { st_flag(pv.flags + HB_FLAG_ABORT, 1u); misc[2] = 1; }
                        if (bad) { st_flag(pv.flags + 9, (unsigned)p + 1u); st_flag(pv.flags + 10, (HBD_SENT(dj) ? 1u : 2u) + (own ? 0x100u : 0u)); }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            rhs = dj - (use_fc ? fc : 0.0);
#if HB_STAMPS
            if (v.dbg && t == 0) v.dbg[(size_t)p * 32 + 12] = c12;
#endif
        }
        const bool act = c_thr < 0.0; // -inf: in the model (always, for these models); +inf: monomorphic or padding (src/Bayes.cpp:589)
        const double gold = c_gold, invv = c_invv, sdz = c_sdz;
        if (gold != 0.0) rhs = fma(c_xx, gold, rhs); // :594 / :616 / :725
        nact += act ? 1 : 0;
        double fc2 = 0.0; // what the sub-blocks more than HBD_NEAR before this one owe it (k_fold_dense; waves >= HBD_NEAR + 1)
        HB_STAMP(0);

#define HBD_STEP(s, BUF)                                                                                                          \
    do {                                                                                                                          \
        if (HB_STAMPS && (s) == 2 && t == 128 && v.dbg) v.dbg[(size_t)p * 32 + 13] = clock64(); /* the serial wave of step 2 enters the step */ \
        if ((s) > 0 && wave >= (s) && wave - ((s) - 1) <= HBD_NEAR) { /* the changes of sub-block s - 1 onto the NEAR later markers of the panel */                    \
            const double2 *d2_ = reinterpret_cast<const double2 *>(dl + 64 * ((s) - 1));                                          \
            _Pragma("unroll") for (int k = 0; k < 64; k += 2) {                                                                   \
                const double2 dk_ = d2_[k >> 1];                                                                                  \
                rhs = fma(-(double)BUF[k], dk_.x, rhs);                                                                           \
                rhs = fma(-(double)BUF[k + 1], dk_.y, rhs);                                                                       \
            }                                                                                                                     \
        }                                                                                                                         \
        if (HB_STAMPS && (s) == 2 && lane == 0 && v.dbg) v.dbg[(size_t)p * 32 + 24 + wave] = clock64(); /* after the apply */     \
        {                                                                                                                         \
            const int rs_ = (s) + 1;                                                                                              \
            if (rs_ < 7 && wave > rs_ && wave - rs_ <= HBD_NEAR) HBD_REQUEST(BUF, p, rs_);                                                                  \
        }                                                                                                                         \
        if (wave == (((s) + 1) & 7)) HBD_PREP((s) + 1); /* the next serial wave prepares its diagonal block (wave 0: the next panel's) */ \
        if ((s) + 1 > HBD_NEAR && wave == (s) + 1) fc2 = ld_sc1(&fcorr2[j]); /* (looked at a step before it is needed) */          \
        double gn_f = 0.0, dmine = 0.0;                                                                                           \
        if (wave == (s)) {                                                                                                        \
            if ((s) > HBD_NEAR) { /* the far sub-blocks' share, summed by k_fold_dense while the near ones were being applied */     \
                if (__any(HBD_SENT(fc2))) {                                                                                       \
                    const unsigned long long t0_ = wall_clock64();                                                                \
                    for (;;) {                                                                                                    \
                        fc2 = ld_sc1(&fcorr2[j]);                                                                                 \
                        if (!__any(HBD_SENT(fc2))) break;                                                                         \
                        const bool own_ = wall_clock64() - t0_ > HB_TIMEOUT_TICKS;                                                \
                        if (ld_flag(pv.flags + HB_FLAG_ABORT) || own_) {                                                          \
                            if (lane == 0) { st_flag(pv.flags + HB_FLAG_ABORT, 1u); st_flag(pv.flags + 9, (unsigned)p + 1u); st_flag(pv.flags + 10, 3u); misc[2] = 1; } \
                            fc2 = 0.0;                                                                                            \
                            break;                                                                                                \
                        }                                                                                                         \
                        __builtin_amdgcn_s_sleep(1);                                                                              \
                    }                                                                                                             \
                }                                                                                                                 \
                rhs -= fc2;                                                                                                       \
            }                                                                                                                     \
            /* The serial pass in units of the CHANGE: sv_j = what marker j's change would be if it were drawn now              */ \
            /* (rhs_j / v_j + sd_j z_j - g_j); marker k's change is then sv_k as it stands at step k, and it moves the later      */ \
            /* markers' by H[k][j] = -G[k][j] / v_j: per step two v_readlane and ONE fused multiply-add on the dependent chain     */ \
            /* (the convert and the product with -1 / v_j do not depend on it and are issued ahead; carrying the right-hand side   */ \
            /* itself is fma - sub - readlane - fma on the chain: 70 cycles per marker, the first version).                        */ \
            double sv_ = fma(rhs, invv, sdz) - gold;                                                                              \
            const double ninvv_ = -invv;                                                                                          \
            if (HB_STAMPS && (s) == 2 && lane == 0 && v.dbg) v.dbg[(size_t)p * 32 + 14] = clock64(); /* ... starts its serial pass */ \
            if (!LASSO) {                                                                                                         \
                _Pragma("unroll") for (int k = 0; k < 64; k++) sv_ = fma((double)dg[k] * ninvv_, readlane_f64(sv_, k), sv_);      \
                gn_f = act ? gold + sv_ : 0.0;                                                                                    \
                dmine = act ? sv_ : 0.0;                                                                                          \
            } else { /* :728 clamps an effect with |g| < 1e-6 to 1e-6 (common while the effects are small): decided in the loop   */ \
                const double forced_ = 1e-6 - gold;                                                                               \
                /* (round 4: the test sits on the dependent chain — add, compare, two selects before the broadcast: BayesL ran 39   */ \
                /* sweeps/s against BayesRR's 55 — and in the stationary regime it almost never fires. So: the pass WITHOUT it      */ \
                /* first; a marker's sv is not touched after its own step, so whether any marker of the sub-block would have been   */ \
                /* clamped at its step can be read off the final values; if none, the pass was the exact one — same operands, same  */ \
                /* operations; if one, again from the saved values with the test in the loop.)                                       */ \
                const double sv0_ = sv_;                                                                                          \
                _Pragma("unroll") for (int k = 0; k < 64; k++) sv_ = fma((double)dg[k] * ninvv_, readlane_f64(sv_, k), sv_);      \
                if (__any(act && fabs(gold + sv_) < 1e-6)) {                                                                      \
                    sv_ = sv0_;                                                                                                   \
                    double ninvv2_ = ninvv_;                                                                                      \
                    asm volatile("" : "+v"(ninvv2_)); /* (hipcc must not keep the first pass's 64 products for this one: 455 spills) */ \
                    _Pragma("unroll") for (int k = 0; k < 64; k++) {                                                              \
                        const double cur_ = (act && fabs(gold + sv_) < 1e-6) ? forced_ : sv_;                                     \
                        int dgk_ = dg[k];                                                                                         \
                        asm volatile("" : "+v"(dgk_));                                                                            \
                        sv_ = fma((double)dgk_ * ninvv2_, readlane_f64(cur_, k), sv_);                                            \
                    }                                                                                                             \
                }                                                                                                                 \
                const bool cl_ = act && fabs(gold + sv_) < 1e-6;                                                                  \
                gn_f = act ? (cl_ ? 1e-6 : gold + sv_) : 0.0;                                                                     \
                dmine = act ? (cl_ ? forced_ : sv_) : 0.0;                                                                        \
            }                                                                                                                     \
            if (HB_STAMPS && (s) == 2 && lane == 0 && v.dbg) v.dbg[(size_t)p * 32 + 15] = clock64(); /* ... has finished it */    \
            dl[t] = dmine;                                                                                                        \
            st_sc1(&dd[j], dmine); /* k_fold_dense is waiting for exactly this */                                                 \
        }                                                                                                                         \
        if (HB_STAMPS && (s) == 2 && lane == 0 && v.dbg) v.dbg[(size_t)p * 32 + 16 + wave] = clock64(); /* at the barrier */      \
        __syncthreads();                                                                                                          \
        HB_STAMP(2 + (s));                                                                                                        \
        if ((s) == 0 && misc[2]) { ok = false; break; }                                                                           \
        if (wave == (s)) { /* off the critical path: the move list, the results, next panel's requests */                         \
            const unsigned long long moved_ = __ballot(dmine != 0.0);                                                             \
            const int evb_ = (s) == 0 ? 0 : misc[0];                                                                              \
            if (dmine != 0.0) {                                                                                                   \
                const int pos_ = evb_ + __popcll(moved_ & lt);                                                                    \
                st_sc1(&v.ev_idx[(size_t)p * P + pos_], t);                                                                       \
                st_sc1(&v.ev_delta[(size_t)p * P + pos_], dmine);                                                                 \
            }                                                                                                                     \
            const double ab_ = wave_sum(fabs(dmine)) + ((s) == 0 ? 0.0 : sabs[0]);                                                \
            const int nev_ = evb_ + __popcll(moved_);                                                                             \
            if (lane == 0) { misc[0] = nev_; sabs[0] = ab_; }                                                                     \
            if (gn_f != gold) v.g[j] = gn_f;                                                                                      \
            v.tracker[j] = act ? (uint8_t)1 : (uint8_t)0;                                                                         \
            if (count_pip && act) {                                                                                               \
                __hip_atomic_fetch_add(&v.nzrate[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                             \
                if (v.wind) v.wflag[v.wind[j] - 1u] = 1;                                                                          \
            }                                                                                                                     \
            if (store && gn_f != 0.0) {                                                                                           \
                unsafeAtomicAdd(&v.alpha_sum[j], gn_f);                                                                           \
                unsafeAtomicAdd(&v.alpha_sq[j], gn_f * gn_f);                                                                     \
            }                                                                                                                     \
            if (act) wacc += gn_f * gn_f;                                                                                         \
            if ((s) == 7) {                                                                                                       \
                if (lane == 0) st_sc1(&v.ev_count[(size_t)p * HB_EVS], nev_);                                                                      \
                evacc += nev_;                                                                                                    \
                absd_grp += ab_;                                                                                                  \
                if (group_end) {                                                                                                  \
                    if (v.mb) {                                                                                                   \
                        mbr = fma(v.xabs, absd_grp, mbr);                                                                         \
                        if (lane == 0) st_sc1(&v.mb[(size_t)(1 + gcount) * HB_MBS], mbr);                                                            \
                    }                                                                                                             \
                    absd_grp = 0.0;                                                                                               \
                }                                                                                                                 \
            } else if (p + 1 < np) {                                                                                              \
                load_coeffs(p + 1);                                                                                               \
                load_dg(p + 1);                                                                                                   \
            }                                                                                                                     \
        }                                                                                                                         \
    } while (0)

        // (strip s lives in bufA for even s, bufB for odd s: step s consumes strip s - 1 and requests strip s + 1 into the same registers)
        HBD_STEP(0, bufB); if (!ok) break;
        HBD_STEP(1, bufA);
        HBD_STEP(2, bufB);
        HBD_STEP(3, bufA);
        HBD_STEP(4, bufB);
        HBD_STEP(5, bufA);
        HBD_STEP(6, bufB);
        HBD_STEP(7, bufA);
        if (t == 0 && p == pv.p0) { const unsigned long long now = wall_clock64(); st_flag(pv.flags + 16, (unsigned)now); st_flag(pv.flags + 17, (unsigned)(now >> 32)); }
        if (group_end) gcount++;
        HB_STAMP(1);
    }
#undef HBD_STEP
#undef HBD_REQUEST

    // ---- the last panel's moves: drain and publish ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 7 && ok && lane == 0) st_flag(pv.flags + HB_FLAG_CHAIN_DONE, (unsigned)np);
    // ---- sweep totals for the hyper-parameter draws (:603 g.g; class counts exclude monomorphic markers) ----
    const double wsum = block_sum(wacc, red);
    const double ev = block_sum((double)(lane == 0 ? evacc : 0), red);
    const double na = block_sum((double)nact, red);
    if (t == 0) {
        v.acc[HB_ACC_SUMG2] += wsum;
        v.acc[HB_ACC_EVENTS] += ev;
        v.acc[HB_ACC_COUNT0 + 1] += na;
        if (!ok) { // aborted: the host must see it (fetch_acc checks the flag), then release every waiter
            st_flag(pv.flags + HB_FLAG_ABORT, 1u);
            st_flag(pv.flags + HB_FLAG_CHAIN_DONE, 0x7fffffffu);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_fold_dense: what the band owes a panel, summed by workgroups spread over the chip.
// Target panel q is owed  fcorr[q P + c] = sum over the source panels p = q - l (l = 1 .. Lv D + q mod D: the panels whose moves
// the mat-vec of q's group has not seen) of  sum_k G_l[q][k][c] delta_p[k];  fcorr2[q P + c] = the same sum over the sub-blocks
// of panel q ITSELF (band block 0) that lie more than HBD_NEAR before column c's (columns of sub-block ch: sub-blocks 0 ..
// ch - HBD_NEAR - 1; the nearer ones the chain applies itself).  Workgroup (tq, ch): the targets q = p0 + tq, + nsets,
// + 2 nsets, ... (nsets = Lb + 1 targets are open at any time), columns 64 ch .. 64 ch + 63; wave w = rows 16 w .. 16 w + 15 of
// every source sub-block of 64; the band's sub-blocks first, then the own panel's. The 16 Gram entries of the next sub-block are requested BEFORE the chain's changes are polled
// (dd[] is sentinel-prefilled: every 8-byte value lands whole), so that what follows the arrival of a sub-block's changes is
// 16 fused multiply-adds; after the last sub-block of panel q - 1 the four row quarters are added in order and the sum is written
// through to fcorr[] — the chain polls it with the panel's dots —, the accumulator restarts, and after the last far sub-block of
// panel q the same goes to fcorr2[].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fold_dense(chain_view v, persist_view pv, const double *__restrict__ dd, double *__restrict__ fcorr2, int nsets)
{
    __shared__ double part[4][64];
    __shared__ int s_abort;
    constexpr int P = HBD_P;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int np = pv.npanels, D = pv.D;
    const int tq = blockIdx.x >> 3, ch = blockIdx.x & 7;
    const size_t PP = (size_t)P * P, pblk = (size_t)(pv.Lg + 1) * PP;
    if (t == 0) {
        s_abort = 0;
        if (atomicAdd(pv.flags + 13, 1u) == 0u) { // (diagnostics, HB_DEBUG_STARTS: when the first fold workgroup started, beside the chain's own start in words 16 / 17)
            const unsigned long long now = wall_clock64();
            st_flag(pv.flags + 18, (unsigned)now);
            st_flag(pv.flags + 19, (unsigned)(now >> 32));
        }
    }
    __syncthreads();
    const int nfar = ch > HBD_NEAR ? ch - HBD_NEAR : 0; // sub-blocks 0 .. nfar - 1 of the target's own panel
    for (int q = pv.p0 + tq; q < np; q += nsets) {
        const int nsrc = pv.Lb > 0 ? min(pv.Lv * D + (q - pv.p0) % D, q - pv.p0) : 0; // source panels q - nsrc .. q - 1
        const int nband = nsrc * 8, nsteps = nband + nfar;
        if (nsteps == 0) continue;
        double acc = 0.0;
        int gv[16], gn[16];
        // step st: the band's sub-blocks first (panel q - l through band block l), then the target panel's own far sub-blocks (block 0)
        auto request = [&](int st, int (&g)[16]) {
            const int l = st < nband ? nsrc - (st >> 3) : 0, s = st < nband ? (st & 7) : st - nband;
            const int32_t *gp = v.gram + (size_t)q * pblk + (size_t)l * PP + (size_t)(64 * s + 16 * wave) * P + 64 * ch + lane;
#pragma unroll
            for (int i = 0; i < 16; i++) g[i] = gp[(size_t)i * P];
        };
        auto dptr = [&](int st) {
            const int l = st < nband ? nsrc - (st >> 3) : 0, s = st < nband ? (st & 7) : st - nband;
            return dd + (size_t)(q - l) * P + 64 * s + 16 * wave + (lane & 15);
        };
        request(0, gv);
        double d = ld_sc1(dptr(0));
        bool dead = false;
        for (int st = 0; st < nsteps; st++) {
            // the next step's Gram entries AND its changes are requested before this step's are looked at: a step whose changes
            // were already published costs no round trip of its own (a target's first source panels are complete when it starts)
            const int stn = min(st + 1, nsteps - 1);
            request(stn, gn);
            double dn = ld_sc1(dptr(stn));
            if (__any(HBD_SENT(d))) {
                const double *dp = dptr(st);
                const unsigned long long t0 = wall_clock64();
                unsigned looks = 0;
                for (;;) {
                    d = (hb_fresh_look(looks) && HBD_SENT(d)) ? ld_fresh(dp) : ld_sc1(dp);
                    if (!__any(HBD_SENT(d))) break;
                    const bool own = wall_clock64() - t0 > HB_TIMEOUT_TICKS;
                    if (ld_flag(pv.flags + HB_FLAG_ABORT) || own) {
                        if (lane == 0) { st_flag(pv.flags + HB_FLAG_ABORT, 1u); s_abort = 1; st_flag(pv.flags + 11, (unsigned)q + 1u); st_flag(pv.flags + 12, (unsigned)st + (own ? 0x100u : 0u)); }
                        const unsigned long long bm = __ballot(HBD_SENT(d));
                        if (bm && lane == __ffsll((long long)bm) - 1)
                            hb_abort_log(pv.flags, HB_LOG_FOLD_DD, own, (unsigned)(dp - dd), (unsigned)q | ((unsigned)st << 16), ~0ull);
                        dead = true;
                        break;
                    }
                    hb_poll_pause(looks, 1);
                    hb_long_wait(looks);
                    looks++;
                }
                if (st + 1 < nsteps) dn = ld_sc1(dptr(st + 1)); // (looked at before this step was there: likely stale)
            }
            if (dead) break;
#pragma unroll
            for (int i = 0; i < 16; i++) acc = fma((double)gv[i], readlane_f64(d, i), acc);
#pragma unroll
            for (int i = 0; i < 16; i++) gv[i] = gn[i];
            d = dn;
            if (st + 1 == nband || st + 1 == nsteps) { // the band's sum (-> fcorr[], polled at the opening) / the far sub-blocks' (-> fcorr2[])
                part[wave][lane] = acc;
                __syncthreads();
                if (wave == 0) {
                    const double tot = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
                    if (st + 1 == nband) st_sc1(&pv.fcorr[(size_t)q * P + 64 * ch + lane], tot);
                    else st_sc1(&fcorr2[(size_t)q * P + 64 * ch + lane], tot);
                }
                acc = 0.0;
                __syncthreads();
            }
        }
        __syncthreads();
        if (s_abort) return;
        if (t == 0) atomicAdd(pv.flags + 14, 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// k_warm_dense: the chain workgroup's own reads — the diagonal blocks and strips of the NEXT panels' Gram blocks, 576 KB per panel —
// pulled into ITS L2 ahead of time. One compute unit gets ~18 bytes per clock out of HBM however many loads it keeps in flight,
// but ~64 out of its XCD's L2, and with the serial pass at 28 cycles per marker the chain would otherwise wait for
// its strips. 8 x per_xcd workgroups are launched; those that did not land on the chain's XCD (it publishes HB_FLAG_XCC) leave.
// The order is static, so this is exact prefetching: rows k of panel q from the diagonal block's first column to the row's end,
// one workgroup per row residue, paced `ahead` panels in front of chain_done. A hint with no dependency: results are discarded.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_warm_dense(chain_view v, persist_view pv, int per_xcd, int ahead, int *__restrict__ sink)
{
    __shared__ int s_rank;
    constexpr int P = HBD_P;
    const int t = threadIdx.x;
    if (t == 0) {
        unsigned my;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
        my &= 15u;
        unsigned want = 0;
        const unsigned long long t0 = wall_clock64();
        while ((want = ld_flag(pv.flags + HB_FLAG_XCC)) == 0u) {
            if (ld_flag(pv.flags + HB_FLAG_ABORT) || wall_clock64() - t0 > 100000000ull) break; // (1 s: the chain never started)
            __builtin_amdgcn_s_sleep(16);
        }
        s_rank = (want == my + 1u) ? (int)(blockIdx.x >> 3) % per_xcd : -1;
    }
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) return;
    const int np = pv.npanels;
    const size_t PP = (size_t)P * P, pblk = (size_t)(pv.Lg + 1) * PP;
    int acc = 0;
    for (int q = pv.p0 + 1; q < np; q++) {
        unsigned done;
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            done = ld_flag(pv.flags + HB_FLAG_CHAIN_DONE);
            if ((int)done + ahead >= q || ld_flag(pv.flags + HB_FLAG_ABORT) || wall_clock64() - t0 > HB_TIMEOUT_TICKS) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if ((int)done >= np || ld_flag(pv.flags + HB_FLAG_ABORT) || (int)done + ahead < q) break;
        if (q <= (int)done) continue; // the chain is already past this panel
        const int32_t *gp = v.gram + (size_t)q * pblk;
        // two rows per pass (128 lanes x 16 bytes each); row k is read from column 64 (k / 64) on
        for (int k = 2 * rank + (t >> 7); k < P; k += 2 * per_xcd) {
            const int c = 4 * (t & 127);
            if (c >= (k & ~63)) {
                const int4 x = *reinterpret_cast<const int4 *>(gp + (size_t)k * P + c);
                acc += x.x ^ x.y ^ x.z ^ x.w;
            }
        }
    }
    if (acc == 0x7fffffff) *sink = acc; // (keeps the loads alive)
}
