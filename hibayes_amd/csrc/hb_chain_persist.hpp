// hb_chain_persist.hpp — k_chain_persist (+ k_hotlist): the persistent per-panel chain workgroup with its LDS row cache, opening ring and speculative rounds (BayesR; RR / A / L on small panels).
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

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
    const int4 *opn;                     // k_chain_group, certified shape: the staged opening's records (hb_ctx.opn); null: the opening loads and computes its own
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
                                                 uint8_t *__restrict__ tracker, int4 *__restrict__ opn = nullptr,
                                                 const int32_t *__restrict__ gB = nullptr, double candf = 1.0)
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
        if (opn) { // the group chain's staged opening: the candidate threshold as it compares it (hot: -1, always; filtered out: NaN, never), the filter word, gB
            const double thc = (f == -__int_as_float(0x7f800000)) ? -1.0 : candf * (double)f;
            opn[j] = make_int4(__double2loint(thc), __double2hiint(thc), __float_as_int(f), gB ? gB[j] : 0);
        }
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

