// hb_chain_panel.hpp — k_chain: the serial conditional updates of one panel, one kernel per panel (the event-ordered fall-back path and the oracle of the persistent chains).
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

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
    // round 5: the band as rank one + int16 residual (hb_ctx.gram16): G[k][j] = ga[k] * gB[j] + gram16[k][j]; null: not built
    const int16_t *gram16;
    const int32_t *ga, *gB;
    const int32_t *gcmax; // the certificate's bound (hb_build_gcert); null: no certificate
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

// v of the lane N below within the lane's row of 16 (DPP row_shr: a modifier of the move, no LDS round trip); 0 where there is none
template <int N>
__device__ __forceinline__ double dpp_row_shr_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// the two wave-wide DPP moves of a 64-lane scan, on a double: row_bcast:15 (lane 15 of a row to the next row; rows 1 and 3 take it) and
// row_bcast:31 (lane 31 to rows 2 and 3); every other lane gets +0.0
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_bcast_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum / running maximum (of non-negative values) over the 64 lanes of a wave, no LDS round trip (round 6: the certificate's
// scans above sixteen candidates were five __shfl_up and twelve __shfl_xor round trips, ~2 000 cycles of a BayesR pass)
__device__ __forceinline__ double wave_scan_incl_f64(double v)
{
    v += dpp_row_shr_f64<1>(v);
    v += dpp_row_shr_f64<2>(v);
    v += dpp_row_shr_f64<4>(v);
    v += dpp_row_shr_f64<8>(v);
    v += dpp_bcast_f64<0x142, 0xa>(v);
    v += dpp_bcast_f64<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ double wave_scan_max_f64(double v)
{
    v = fmax(v, dpp_row_shr_f64<1>(v));
    v = fmax(v, dpp_row_shr_f64<2>(v));
    v = fmax(v, dpp_row_shr_f64<4>(v));
    v = fmax(v, dpp_row_shr_f64<8>(v));
    v = fmax(v, dpp_bcast_f64<0x142, 0xa>(v));
    v = fmax(v, dpp_bcast_f64<0x143, 0xc>(v));
    return v;
}

// inclusive prefix sum over the 64 lanes of a wave without an LDS round trip per step: four row_shr steps inside the rows of 16, then the row
// totals handed on by row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3) — the gfx9 wave-wide DPP modes
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
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

