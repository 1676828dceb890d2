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
//
// ONE translation unit, by role in included files (round 5; the kernels share device globals — the wait bound, the abort log —
// and the view structs, which separate objects could only share through relocatable device code):
//   hand-offs        hb_handoff.hpp        flag block, sc1 loads / write-through stores, bounded waits, reductions
//   mat-vec          hb_matvec.hpp         k_dot, k_dotq (int8 columns)        hb_dotq2.hpp   k_dotq2 / k_dotq2r / k_dotq2m (2-bit)
//                    hb_update.hpp         the residual update rows that ride in the launches, and their dense form
//   chains           hb_pre.hpp            k_pre (thresholds, deviates)
//                    hb_chain_panel.hpp    k_chain (one kernel per panel)
//                    hb_chain_persist.hpp  k_chain_persist, k_hotlist          hb_chain_group.hpp  k_chain_group, k_fwd
//                    hb_chain_dense.hpp    k_chain_dense, k_fold_dense         hb_warm.hpp         k_gate, k_warm
//   host blocks      hb_blocks.hpp         intercept / covariates / random effects, delta pack / unpack
//                    hb_reduce.hpp         var(u), yadj.yadj, BayesL's variances, GWAS windows
//   ingest / egress  hb_stats.hpp          xpx, vx                              hb_ingest.hpp  f64 check, .bed decode, X alpha, GEBV, generator
//   summary level    hb_sbayes.hpp         SBayesD on a dense LD matrix
//   this file        sweep start (k_sweep_init, k_quant0), the launchers, graph capture, probes, thin wrappers for hb_ctx.hip
#include "hb_internal.hpp"
#include "hb_rng.hpp"
#include <type_traits>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

#define HB_INF __builtin_huge_val()

#include "hb_handoff.hpp"
#include "hb_stats.hpp"
#include "hb_update.hpp"
#include "hb_matvec.hpp"
#include "hb_dotq2.hpp"
// The persistent mat-vec is an EXPERIMENT (round 6; DESIGN section 6.0: it does not hold the pace of the launches and stalls at the wide geometry): built only
// with -DHB_WITH_MVP=1 (tools/build_variant.sh mvp "-DHB_WITH_MVP=1"; then HB_MVP=1 selects it). Its instantiations stay out of the default library on
// purpose: merely having them in this translation unit changed the inlining around the headline chain kernel, which sits at 256 registers — five spilled
// registers, 446 -> 421 sweeps/s (tools/kernel_resources.sh shows the spills per kernel).
#ifndef HB_WITH_MVP
#define HB_WITH_MVP 0
#endif
#ifndef HB_W8_CH
#define HB_W8_CH 3 /* moves per trip of the eight-panel group chain (4 spills nine registers, 3 four) */
#endif
#if HB_WITH_MVP
#include "hb_mvp.hpp"
#endif

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

// sweep start of the fixed-point path, in two steps that use the whole device (round 5: one workgroup of 1024 took 33 us over n = 50 000 —
// 49 dependent loads per thread, twice — and k_pre + k_hotlist beside it are done after 30): max |yadj| (exact in any order: the doubles
// are non-negative, so their bit patterns order like the numbers and one integer atomicMax per wave collects them; *out zeroed before),
// then the digits of every row on that exponent.
__global__ __launch_bounds__(256) void k_absmax(const double *__restrict__ r, int64_t ld, unsigned long long *__restrict__ out)
{
    double mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ld; i += (int64_t)gridDim.x * blockDim.x) mx = fmax(mx, fabs(r[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));
}

__global__ __launch_bounds__(256) void k_quant0(const double *__restrict__ r, int64_t ld, int8_t *__restrict__ rq, double *__restrict__ mb,
                                                int *__restrict__ vexp, const double *__restrict__ maxp)
{
    // (maxp: the word k_absmax left in mb[0], or — row-sharded mode — max |yadj| over ALL shards, so that every shard's digits share one exponent)
    const double m2 = *maxp;
    const int E = hb_fix_exp(m2);
    for (int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; row0 < ld; row0 += (int64_t)gridDim.x * blockDim.x * 4)
        hb_store_digits(rq, ld, row0, E, r[row0], r[row0 + 1], r[row0 + 2], r[row0 + 3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (maxp != mb) mb[0] = m2;
        vexp[0] = E;
    }
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

#include "hb_pre.hpp"
#include "hb_chain_panel.hpp"
#include "hb_chain_persist.hpp"
#include "hb_chain_group.hpp"
#include "hb_chain_dense.hpp"
#include "hb_warm.hpp"
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

#include "hb_reduce.hpp"
#include "hb_blocks.hpp"
#include "hb_ingest.hpp"
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
#define HB_GROUP_ATTR16(K1, DM, FW, CH) HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<K1, DM, FW, CH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    HB_GROUP_ATTR(1, 8, 14, 3); HB_GROUP_ATTR(1, 8, 7, 4); HB_GROUP_ATTR(1, 2, 4, 10); HB_GROUP_ATTR(1, 1, 2, 20);
    HB_GROUP_ATTR(3, 1, 2, 20); HB_GROUP_ATTR(7, 1, 2, 20);
    HB_GROUP_ATTR(3, 8, 14, 3); HB_GROUP_ATTR(3, 8, 7, 4); HB_GROUP_ATTR(3, 2, 4, 10); // round 6: BayesR (K <= 4 classes) on the group chain at every shape
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<3, 8, 7, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<3, 2, 4, 10, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<3, 4, 8, 5, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<3, 2, 2, 15, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group_fwd<1, 8, 7, 4, true, 7, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group_fwd<1, 8, 7, 4, true, 7, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_GROUP_ATTR16(1, 8, 7, 4);
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<1, 8, 7, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<1, 8, 8, HB_W8_CH, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#if HB_WITH_MVP
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<1, 8, 7, 4, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain_group<1, 2, 4, 10, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
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
    // the matrix-core kernel's shape (hb_dotq2.hpp): column tiles of 16 per wave, stages requested together, one accumulator set per scale or one
    int q2m_ct = c->q2m_ct, q2m_g = c->q2m_g;
    while (mfma && q2m_ct > 4 && ncols % (16 * q2m_ct)) q2m_ct /= 2;
    if (q2m_ct == 16 && q2m_g == 2) q2m_g = 1;
    if ((q2m_g == 0 || q2m_g == 3) && (c->ld % 512 != 0 || ncols % 64)) q2m_g = 1; // (the 512-individual stages read whole stages of digits: the padded length must be a multiple)
    if (q2m_g == 0 || q2m_g == 3) q2m_ct = 4;
    const int RS = mfma ? ((q2m_g == 0 || q2m_g == 3) ? 512 : Q2M_RS) : c->dotq2_rs;
    const int nst = (int)((c->ld + RS - 1) / RS);
    const int ncg = mfma ? ncols / (16 * q2m_ct) : ncols / (64 * cpl);
    // (a tile is at least four stages: its first stage's load latency and its closing atomics are paid per tile)
    // (the matrix-core kernel streams best with few, long tiles — its per-stage work is an eighth of the v_dot4 kernel's, so a tile's fixed
    // costs weigh more: ~800 tiles per 3584-column launch)
    int ns = std::max(1, std::min(std::max(1, nst / 4), (int)((double)(mfma && !getenv("HB_DOTQ2_TILES") ? ((q2m_g == 0 || q2m_g == 3) ? 900 : 800) : c->dotq2_tiles) / ncg + 0.5)));
    // (int32 accumulators of genotypes scaled by up to 32 — Q2_SCALED, k_dotq2m: rows x 96 x 128 < 2^31 bounds a tile at 174 000 individuals)
    ns = std::max(ns, (int)(((int64_t)nst * RS + 131071) / 131072));
    upd_view uq{};
    if (upd) uq = *upd;
    const int nupd_blk = (uq.p1 > uq.p0) ? (int)(c->ld / (uq.dense ? 64 : 256)) : 0, nfin_blk = fin_ncols > 0 ? (fin_ncols + 63) / 64 : 0;
    if (mfma && (q2m_g == 0 || q2m_g == 3) && !getenv("HB_DOTQ2_TILES")) {
        // ALL blocks of the launch resident at once (round 5, the last measurement of the round). A block of this kernel holds 37 KB of LDS: four per
        // compute unit, 128 per XCD — less the chain workgroup's compute unit and k_fwd's share of another, which sit on ONE XCD — and the
        // dispatcher deals the blocks round-robin over the eight XCDs whatever they have free. 784 tiles + 196 update + 56 finalize blocks = 1 036
        // is 130 per XCD: the XCD with the chain started its last nine tiles when its first ones ended, 8.4 us into a 9-us launch, and the launch
        // took 14.6 us in situ (tools/launch_roles.py). Fewer, longer tiles until the fullest XCD's share fits its slots: 12.6 us, 433 -> 454 sweeps/s.
        const int lds = q2m_g == 3 ? q2m512_lds<true>() : q2m512_lds<false>();
        const int per_cu = std::max(1, std::min(8, (160 * 1024) / std::max(1, lds)));
        const int cus_per_xcd = std::max(1, c->num_cus / 8);
        const int budget = 8 * (cus_per_xcd * per_cu - (per_cu + 3)); // (the fullest XCD gets ceil(blocks / 8))
        auto total = [&](int k) { const int NSk = (nst + k - 1) / k; return nupd_blk + nfin_blk + ncg * ((nst + NSk - 1) / NSk); };
        const int ns_min = std::max(1, (int)(((int64_t)nst * RS + 131071) / 131072));
        while (ns > ns_min && total(ns) > budget) ns--;
    }
    const int NS = (nst + ns - 1) / ns, nsplit = (nst + NS - 1) / NS;
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
    v.nupd = nupd_blk;
    v.nfin = nfin_blk;
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
    if (mfma) {
#define HB_Q2M_LAUNCH(CT, G, SC) hipLaunchKernelGGL((k_dotq2m<CT, G, SC>), dim3(nblk), dim3(64), (q2m_lds<CT, G>()), st, v, uq)
        const bool sc = c->q2m_sc != 0;
        if (q2m_g == 0) {
            if (sc) hipLaunchKernelGGL((k_dotq2m<4, 0, true>), dim3(nblk), dim3(64), q2m512_lds<false>(), st, v, uq);
            else hipLaunchKernelGGL((k_dotq2m<4, 0, false>), dim3(nblk), dim3(64), q2m512_lds<false>(), st, v, uq);
        } else if (q2m_g == 3) {
            if (sc) hipLaunchKernelGGL((k_dotq2m<4, 3, true>), dim3(nblk), dim3(64), q2m512_lds<true>(), st, v, uq);
            else hipLaunchKernelGGL((k_dotq2m<4, 3, false>), dim3(nblk), dim3(64), q2m512_lds<true>(), st, v, uq);
        } else if (q2m_ct == 16) { if (sc) HB_Q2M_LAUNCH(16, 1, true); else HB_Q2M_LAUNCH(16, 1, false); }
        else if (q2m_ct == 8 && q2m_g == 2) { if (sc) HB_Q2M_LAUNCH(8, 2, true); else HB_Q2M_LAUNCH(8, 2, false); }
        else if (q2m_ct == 8) { if (sc) HB_Q2M_LAUNCH(8, 1, true); else HB_Q2M_LAUNCH(8, 1, false); }
        else if (q2m_g == 2) { if (sc) HB_Q2M_LAUNCH(4, 2, true); else HB_Q2M_LAUNCH(4, 2, false); }
        else { if (sc) HB_Q2M_LAUNCH(4, 1, true); else HB_Q2M_LAUNCH(4, 1, false); }
#undef HB_Q2M_LAUNCH
    }
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
    const int qb = (int)std::max<int64_t>(1, std::min<int64_t>(64, (c->ld / 4 + 255) / 256));
    (void)hipMemsetAsync(c->mb, 0, sizeof(double), st);
    hipLaunchKernelGGL(k_absmax, dim3(qb), dim3(256), 0, st, c->r, c->ld, reinterpret_cast<unsigned long long *>(c->mb));
    hipLaunchKernelGGL(k_quant0, dim3(qb), dim3(256), 0, st, c->r, c->ld, c->rq, c->mb, c->vexp, (const double *)c->mb);
    if (c->row_reduce) { // the shards' maxima -> one exponent for all (host round trip: this is the cross-check mode)
        (void)hipStreamSynchronize(st);
        std::vector<double> slot((size_t)std::max(1, c->row_world), 0.0);
        double mxl = 0.0;
        (void)hipMemcpy(&mxl, c->mb, sizeof(double), hipMemcpyDeviceToHost);
        slot[c->row_rank] = mxl;
        if (c->row_reduce(c->row_user, slot.data(), slot.size())) { c->row_failed = true; return; }
        for (double v : slot) mxl = std::max(mxl, v);
        (void)hipMemcpy(c->scratch, &mxl, sizeof(double), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_quant0, dim3(qb), dim3(256), 0, st, c->r, c->ld, c->rq, c->mb, c->vexp, (const double *)c->scratch);
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
        hipLaunchKernelGGL(k_reduce_ru, dim3(16), dim3(64), 0, sA, c->r, c->u, c->n, c->acc, c->ru_ws, c->flags);
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
                       c->hot_list, c->thr0f, c->tracker, (c->gcert_ok && c->gB) ? c->opn : nullptr, c->gB, c->candf);
    if (fx) HB_HIP(hipStreamWaitEvent(sA, c->ev_upd[1 % c->npanels], 0));
    HB_HIP(hipEventRecord(c->ev_fork, sA));
    HB_HIP(hipStreamWaitEvent(sB, c->ev_fork, 0));
    const double xabs = std::max(std::abs((double)c->xmin), std::abs((double)c->xmax));
    chain_view cv{c->m_pad, c->P, c->nsplit, Lv, c->Lg, c->xpx, c->vx, c->g, c->tracker, c->nzrate, c->alpha_sum, c->alpha_sq,
                  c->thr, c->invv, c->sdz, c->gram, c->partial, c->dsum, c->ev_count, c->ev_idx, c->ev_delta, c->acc,
                  c->wind, c->wflag, c->dbg, fx ? c->mb : nullptr, xabs};
    const bool g16 = c->gram16_ok && c->gram16 != nullptr && kp == 1; // (the compact band: only the wide group chain and its k_fwd read it)
    if (g16) { cv.gram16 = c->gram16; cv.ga = c->ga; cv.gB = c->gB; }
    const bool cert = !g16 && c->gcert_ok && c->gcmax != nullptr; // (the wide group chain's certified violation check)
    if (cert) { cv.ga = c->ga; cv.gB = c->gB; cv.gcmax = c->gcmax; }
    const int last_panels = np - (g0 + ngroups - 1) * D;
    persist_view pv{np, D, Lv, c->L, c->Lg, pb, c->flags,
                    c->hot_slot, c->hot_list, c->thr0f, c->candf, nullptr, nullptr};
    if (cert) pv.opn = c->opn; // (k_hotlist wrote it: gcert_ok)
    // HB_CHAIN_ALONE=1 / hb_ctx_set_profiling(c, 4) — a TIMING AND COUNTER DIAGNOSTIC, results are meaningless (it needs no
    // co-resident kernels, so it is also how k_chain_persist runs under a counter-collecting profiler, tools/chain_counters.py): the mat-vec launches run first against a pre-set
    // chain_done (their update rows find empty event lists), the chain afterwards with the device to itself; the stamped span
    // (tools/chain_timeline.py with CT_ALONE=1) is then what the chain costs without the mat-vec's memory traffic beside it.
    const bool alone = c->chain_alone || getenv("HB_CHAIN_ALONE") != nullptr;
    // the point-mass models run the group-granular chain (hb_chain_group.hpp); HB_CHAIN=panel keeps the per-panel one
    // (chain_kind bit 0: BayesB / BayesC; bit 1: the dense models too — BayesR and RR / A / L at one panel per group)
    const int shape = (D <= 1 && Lv * D <= 2) ? 2 : (D <= 2 && Lv * D <= 4) ? 1 : (D <= 8 && Lv * D <= 14) ? 0 : (c->fwd_group && c->P == 512 && ((Lv == 3 && D == 7) || (Lv == 2 && D == 8))) ? 0 : -1;
    const bool sparse_model = kp == 1 && (model == 3 || model == 4);
    // round 6: BayesR with up to four classes (kp == 3) runs the group chain too wherever a launch covers more than one panel (chain_kind bit 2
    // clear; HB_CHAIN=panel keeps k_chain_persist). K1 nested thresholds per candidate instead of one; everything else — candidates, certificate
    // (it bounds the right-hand side, not the class), fold, k_fwd — is the point-mass models' path.
    const bool mix_model = kp == 3 && model == 6 && D >= 2;
    const bool group_chain = !dense && shape >= 0 && !c->chain_alone && (sparse_model ? (c->chain_kind & 1) != 0 : mix_model ? c->chain_kind != 0 : ((c->chain_kind & 2) != 0 && shape == 2));
    // k_fwd beside the wide group chain: the chain folds a move into its own group and the next (15 rows, four moves per trip),
    // a second workgroup into the group after that (HB_FWD=0: the chain does all 22 rows itself, three moves per trip)
    // round 6: also beside BayesR's two-panel groups ((2, 2): the chain folds a move into the next group's two panels, k_fwd into the two after — half of the
    // chain's fold rows leave its compute unit, and a group's ~16 moves fit ONE trip of 62 loads per lane instead of two of 60)
    const bool fwd2 = group_chain && mix_model && Lv == 2 && D == 2 && c->P == 512 && c->fwd_group && !alone && cert && !getenv("HB_FWD2_OFF");
    // round 6: eight panels per launch (Lv = 2 only, point-mass models, certified): k_chain_group<1, 8, 8, CH, CERT> + k_fwd<8, 1, 8>
    const bool wide8 = group_chain && kp == 1 && Lv == 2 && D == 8 && c->P == 512 && c->fwd_group && !alone && cert;
    const bool fwd = (group_chain && (kp == 1 || mix_model) && (Lv == 2 || Lv == 3) && D == 7 && c->P == 512 && c->fwd_group && !alone) || fwd2 || wide8;
    const bool warm_r_env_off = !(getenv("HB_WARM_G") && atoi(getenv("HB_WARM_G")) > 0);
    // (only where chain and k_fwd run as ONE kernel — the wide certified shape of BayesB / BayesC: four branches in all, what a process has queues for)
    const bool overlap = c->overlap && fx && !dense && !alone && !c->lstamp && ngroups > Lv + 2 && Lv + 1 <= 8 && c->s_fk != nullptr &&
                         fwd && !fwd2 && cert && kp == 1 && !g16 && warm_r_env_off;
    const bool merged = overlap;
    // round 6: the persistent mat-vec (hb_mvp.hpp, HB_MVP=1): 2-bit genotypes, matrix-core tiles on 512-individual stages, whole sweeps
    const bool mvp = HB_WITH_MVP && c->mvp && fx && c->layout == 2 && c->X2 && c->dotq2_kind == 2 && (c->q2m_g == 0) && c->ld % 512 == 0 && !dense && !alone && !overlap &&
                     pb == 0 && ngroups > Lv + 2 && c->rq_slots >= ngroups + 1 && c->mvp_ho != nullptr &&
                     kp == 1 && group_chain && !g16 && ((fwd && cert) || (!fwd && shape == 1)); // (the two chain shapes instantiated with memory-side looks)
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
            if (merged) { // chain + k_fwd as the two workgroups of one kernel (the overlapped launch stream)
                if (Lv == 2) hipLaunchKernelGGL((k_chain_group_fwd<1, 8, 7, 4, true, 7, 1, 8>), dim3(2), dim3(c->P), sm, st, c->d_in, cv, pv);
                else hipLaunchKernelGGL((k_chain_group_fwd<1, 8, 7, 4, true, 7, 2, 4>), dim3(2), dim3(c->P), sm, st, c->d_in, cv, pv);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return hb_fail(HB_ERR_HIP, std::string("k_chain_group_fwd launch: ") + hipGetErrorString(e));
                return HB_OK;
            }
            if (mix_model) {
                if (fwd2) hipLaunchKernelGGL((k_chain_group<3, 2, 2, 15, false, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
                else if (fwd && cert) hipLaunchKernelGGL((k_chain_group<3, 8, 7, 4, false, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
                else if (fwd) hipLaunchKernelGGL((k_chain_group<3, 8, 7, 4>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
                else if (cert && c->P == 512 && D <= 4 && Lv * D <= 8 && !(D <= 2 && Lv * D <= 4) && !getenv("HB_CERT_NARROW_OFF"))
                    hipLaunchKernelGGL((k_chain_group<3, 4, 8, 5, false, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv); // (three or four panels per launch, certified)
                else if (shape == 0) hipLaunchKernelGGL((k_chain_group<3, 8, 14, 3>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
                else if (cert && c->P == 512 && !getenv("HB_CERT_NARROW_OFF")) hipLaunchKernelGGL((k_chain_group<3, 2, 4, 10, false, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv); // (shape 1: D = 2, certified)
                else hipLaunchKernelGGL((k_chain_group<3, 2, 4, 10>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv); // (shape 1: D = 2; shape 2 needs D <= 1)
            }
#if HB_WITH_MVP
            else if (mvp && fwd) hipLaunchKernelGGL((k_chain_group<1, 8, 7, 4, false, true, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else if (mvp) hipLaunchKernelGGL((k_chain_group<1, 2, 4, 10, false, false, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
#endif
            else if (wide8) hipLaunchKernelGGL((k_chain_group<1, 8, 8, HB_W8_CH, false, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else if (fwd && g16) hipLaunchKernelGGL((k_chain_group<1, 8, 7, 4, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else if (fwd && cert) hipLaunchKernelGGL((k_chain_group<1, 8, 7, 4, false, true>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
            else if (fwd) hipLaunchKernelGGL((k_chain_group<1, 8, 7, 4>), dim3(1), dim3(c->P), sm, st, c->d_in, cv, pv);
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
    bool fold_first = false;
    const bool side_first = !alone && getenv("HB_SIDE_FIRST") && atoi(getenv("HB_SIDE_FIRST")) != 0; // (A/B: k_fwd and the warmers enqueued before the chain, as k_fold_dense is)
    if (alone) { // (the update rows poll the move counts themselves: "no moves" for every panel)
        HB_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->flags + HB_FLAG_CHAIN_DONE), 0x7ffffff0, 1, sA));
        HB_HIP(hipMemsetAsync(c->ev_count, 0, sizeof(int32_t) * (size_t)c->npanels * HB_EVS, sA));
        if (fx) HB_HIP(hipMemsetAsync(c->mb + HB_MBS, 0, sizeof(double) * ((size_t)c->npanels + 1) * HB_MBS, sA));
    }
    else {
        // round 6: k_fold_dense is enqueued BEFORE the chain and the gate. Captured after them, the graph started it ~1 ms late — the dense chain waits for its
        // first far sums at sub-block 4 of the sweep's first panel, launch 2's update rows wait for the chain: 1 ms of every 18 ms sweep
        // (tools/r6_long_launch.py, tools/r6_dense_start.py; HB_GATE=0 or HB_GRAPH=0 alone also removed it; HB_FOLD_FIRST=0: the old order)
        fold_first = dense && !(getenv("HB_FOLD_FIRST") && atoi(getenv("HB_FOLD_FIRST")) == 0);
        if (fold_first) {
            HB_HIP(hipStreamWaitEvent(c->s_upd, c->ev_fork, 0));
            hipLaunchKernelGGL(k_fold_dense, dim3(8 * (c->L + 1)), dim3(256), 0, c->s_upd, cv, pv, c->ddense, c->fcorr2, c->L + 1);
            HB_HIP(hipGetLastError());
        }
        if (!side_first) {
            if (int rc = launch_the_chain(sB)) return rc;
            // (the first mat-vec launch starts when the chain is resident; HB_GATE=0 / 1 overrides: by default only where a launch's
            // update blocks can sit on every compute unit)
            bool gate = dense || overlap || mvp; // (overlap / persistent mat-vec: tiles from the first microsecond on, no compute unit left free for the chain's 160 KB of LDS)
            if (const char *e = getenv("HB_GATE")) gate = atoi(e) != 0;
            if (gate) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sA, c->flags);
        }
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
    // the wide group chain with k_fwd beside it (BayesB / BayesC, the headline): warmers on the same fourth stream — the listed markers' Gram rows
    // for the chain's share of the band (its own group and the next: 2 D - 1 blocks) and the panels' exact per-marker data (HB_WARM_G: workgroups
    // per XCD, 0 = off)
    if (fwd && c->s_warm && warm_r == 0) {
        warm_r = fwd2 ? 4 : c->warm_g; // (BayesR's two-panel groups, round 6: 92.1 sweeps/s without, 95.9 / 97.5 / 96.5 with 2 / 4 / 8 workgroups per XCD, profiles/r06_bayesr_conv_warm.txt; the wide BayesCpi shape: no effect, round 5)
        if (const char *e = getenv("HB_WARM_G")) warm_r = std::max(0, std::min(16, atoi(e)));
    }
    if (dense && !fold_first) { // (Lb + 1 target panels are open at any time: Lb ahead for their band, the chain's own for its far sub-blocks)
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
    if (fwd && !merged) {
        HB_HIP(hipStreamWaitEvent(c->s_upd, c->ev_fork, 0));
        if (fwd2) hipLaunchKernelGGL((k_fwd<2, 1, 16>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
#if HB_WITH_MVP
        else if (mvp && Lv == 2) hipLaunchKernelGGL((k_fwd<7, 1, 8, false, true>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
        else if (mvp) hipLaunchKernelGGL((k_fwd<7, 2, 4, false, true>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
#endif
        else if (wide8) hipLaunchKernelGGL((k_fwd<8, 1, 8>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
        else if (Lv == 2 && g16) hipLaunchKernelGGL((k_fwd<7, 1, 8, true>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
        else if (Lv == 2) hipLaunchKernelGGL((k_fwd<7, 1, 8>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
        else if (g16) hipLaunchKernelGGL((k_fwd<7, 2, 4, true>), dim3(1), dim3(c->P), 0, c->s_upd, cv, pv);
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
        int ahead = fwd ? 2 * D : 2; // (measured, BayesR at n = 50k, m = 500k: off 48.3 sweeps/s, 2 panels ahead 51.2, 4 ahead 50.5, 8 ahead 50.0)
        if (const char *e = getenv("HB_WARM_AHEAD")) ahead = std::max(1, atoi(e));
        persist_view pw = pv;
        pw.Lb = fwd ? 2 * D - 1 : 1; // (BayesR: the chain folds into the next panel only; the group chain: into its own group's later panels and the next group's)
        hipLaunchKernelGGL(k_warm, dim3(8 * warm_r), dim3(256), 0, c->s_warm, pw, cv, kp, c->gram, c->P, ahead, warm_r, reinterpret_cast<int *>(c->flags + 48));
        HB_HIP(hipGetLastError());
    }
    if (side_first) {
        if (int rc = launch_the_chain(sB)) return rc;
        bool gate = dense || overlap || mvp;
        if (const char *e = getenv("HB_GATE")) gate = atoi(e) != 0;
        if (gate) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, sA, c->flags);
    }
    const bool inject = c->inject_abort_panel >= 0 && c->s_dbg && !alone;
    if (inject) { // (debug hook: a fourth branch that aborts the sweep in mid-flight)
        HB_HIP(hipStreamWaitEvent(c->s_dbg, c->ev_fork, 0));
        hipLaunchKernelGGL(k_inject_abort, dim3(1), dim3(1), 0, c->s_dbg, c->flags, (unsigned)std::min(c->inject_abort_panel, np));
        HB_HIP(hipGetLastError());
    }
    const int upd_blocks = (int)((c->ld / 4 + 255) / 256);
    // ---- Round 6: the OVERLAPPED launch stream (HB_OVERLAP=1). A 3 584-column launch lives 12 us of which its tiles run 9: the rest is the ramp of a
    // kernel that is too small for the chip, plus 1.6 us of dependent-dispatch gap — and two such launches in flight stream 21 % more
    // (9.6 against 12.2 us per launch isolated, tools/matvec_only.py with HB_TM_STREAMS=2; the int8 launches reach the measured read ceiling, 6.1 TB/s).
    // What ties launch g to launch g - 1 is only what RIDES in it — the update rows that write the residual version the next launch reads, and the finalize
    // rows of the previous launch's sums. Here both are kernels of their own: the mat-vec launches (tiles only) alternate between two streams, the residual
    // updates u(h) follow each other on a third (each still polls the chain's counts for its group on the device), the finalize kernels f(g) on a
    // fourth; t(g) waits for u(g - Lv - 1) — the version it reads — and f(g) for t(g): graph edges, no new device-side hand-off, no coherence question.
    // The residual versions live in a ring of Lv + 1 slots: while t(g) reads version g - Lv - 1 the chain has closed group g - 1 at most, so the
    // youngest version written is g - 1; u(g), which overwrites the slot t(g) reads, needs chain_done(g), i.e. the sums of t(g). Same integers, same
    // chain: the band, the correction ring and k_fwd see the geometry they always saw. ----
    if (overlap) {
        const int NBr = Lv + 1;
        auto slotn = [&](int v) { return v < 0 ? 0 : (v + 1) % NBr; };
        // (the second tile stream starts behind the gate / the sweep's start like the first: an event on sA here)
        HB_HIP(hipEventRecord(c->ev_chain[2 % c->npanels], sA));
        HB_HIP(hipStreamWaitEvent(c->s_t2, c->ev_chain[2 % c->npanels], 0));
        HB_HIP(hipStreamWaitEvent(c->s_uk, c->ev_fork, 0));
        for (int g = 0; g < ngroups; g++) {
            const int ga = g0 + g, p0 = ga * D, p1 = std::min(np, p0 + D);
            hipStream_t st = (g & 1) ? c->s_t2 : sA;
            const int v = g - Lv - 1; // the residual version the tiles read
            if (v >= 0) HB_HIP(hipStreamWaitEvent(st, c->ev_ou[v], 0));
            launch_dot(c, p0 * c->P, (p1 - p0) * c->P, slotn(v), st, true, nullptr, 0, 0, ga);
            // (the finalize kernel follows its tiles in their stream: one edge per group — u(v) -> t(v + Lv + 1) — is all the graph holds besides its three
            // chains; with an event per tile launch, finalize and update the executor's traversal of the captured graph did not return)
            launch_dotq_fin(c, p0 * c->P, (p1 - p0) * c->P, ga, c->dsum + (size_t)p0 * c->P, st);
            upd_view uq = make_upd(c, p0, p1, slotn(g - 1), slotn(g), c->flags, ga);
            hipLaunchKernelGGL(k_update, dim3(upd_blocks), dim3(256), 0, c->s_uk, c->ld, uq);
            HB_HIP(hipEventRecord(c->ev_ou[g], c->s_uk));
        }
        // join: everything back into sA
        HB_HIP(hipEventRecord(c->ev_ot[0], c->s_t2));
        HB_HIP(hipStreamWaitEvent(sA, c->ev_ot[0], 0));
        HB_HIP(hipStreamWaitEvent(sA, c->ev_ou[ngroups - 1], 0));
    }
#if HB_WITH_MVP
    if (mvp) {
        const int ncols = D * c->P, nst = (int)(c->ld / 512), ncg = ncols / 64, nupd_blk = (int)(c->ld / 256);
        // tiles per group: as a k_dotq2m launch sizes them (launch_dotq2) — all workgroups of the kernel resident at once
        int ns = std::max(1, std::min(std::max(1, nst / 4), (int)(900.0 / ncg + 0.5)));
        const int ns_min = std::max(1, (int)(((int64_t)nst * 512 + 131071) / 131072));
        ns = std::max(ns, ns_min);
        {
            const int lds = q2m512_lds<false>(), per_cu = std::max(1, std::min(8, (160 * 1024) / std::max(1, lds))), cus_per_xcd = std::max(1, c->num_cus / 8);
            const int budget = 8 * (cus_per_xcd * per_cu - (per_cu + 3));
            auto total = [&](int k) { const int NSk = (nst + k - 1) / k; return nupd_blk + ncg * ((nst + NSk - 1) / NSk); };
            while (ns > ns_min && total(ns) > budget) ns--;
        }
        const int NS = (nst + ns - 1) / ns, nsplit = (nst + NS - 1) / NS;
        mvp_view mv{};
        mv.v0.X = nullptr;
        mv.v0.X2 = reinterpret_cast<const uint8_t *>(c->X2) + (int64_t)(g0 * D) * c->P * c->ld2;
        mv.v0.ld2 = c->ld2;
        mv.v0.ld = c->ld;
        mv.v0.gexp_out = c->gexp + g0;
        mv.v0.accq = c->accq + (int64_t)(g0 * D) * c->P;
        mv.v0.accstride = c->m_pad;
        mv.v0.nstages = nst;
        mv.v0.NS = NS;
        mv.v0.ncg = ncg;
        mv.u0 = make_upd(c, 0, 0, 0, 0, c->flags, 0);
        auto envi = [](const char *k, int d) { const char *e = getenv(k); return e ? atoi(e) : d; };
        mv.ufresh = std::max(1, envi("HB_MVP_UFRESH", 4));
        mv.usleep = std::max(0, envi("HB_MVP_USLEEP", 0));
        mv.tfresh = std::max(1, envi("HB_MVP_TFRESH", 4));
        mv.tsleep = std::max(0, envi("HB_MVP_TSLEEP", 0));
        mv.ngroups = ngroups; mv.D = D; mv.np = np; mv.g0 = g0; mv.Lv = Lv;
        mv.ntile = ncg * nsplit; mv.nupd = nupd_blk; mv.nsplit = nsplit;
        mv.rqv = c->rq; mv.vexpv = c->vexp; mv.r = c->r; mv.mb = c->mb; mv.r32 = c->r32; mv.dsum = c->dsum;
        mv.ho = c->mvp_ho; mv.flags = c->flags;
        mv.rel = c->mvp_ho + ((size_t)(c->npanels + 2) * 65); // (behind the counters: 64 lines)
        mv.stamp = nullptr;
        const int nblk = mv.nupd + mv.ntile;
        if (c->lstamp && nblk <= HB_LSTAMP_BLOCKS) {
            mv.stamp = c->lstamp;
            for (int g = 0; g < ngroups; g++) { c->lstamp_nblk[g0 + g] = nblk; c->lstamp_cols[g0 + g] = (std::min(np, (g0 + g + 1) * D) - (g0 + g) * D) * c->P; }
        }
        HB_HIP(hipMemsetAsync(c->mvp_ho, 0, sizeof(unsigned) * ((size_t)(ngroups + 1) + (size_t)ngroups * 64), sA));
        HB_HIP(hipMemsetAsync(mv.rel, 0, sizeof(unsigned) * 64 * 32, sA));
        if (c->q2m_sc) hipLaunchKernelGGL((k_mvp2<true>), dim3(nblk), dim3(64), q2m512_lds<false>(), sA, mv);
        else hipLaunchKernelGGL((k_mvp2<false>), dim3(nblk), dim3(64), q2m512_lds<false>(), sA, mv);
        HB_HIP(hipGetLastError());
    }
#endif
    // Residual versions advance per mat-vec group: version h = every panel of groups <= h applied. Mat-vec launch g
    // reads version g - Lv - 1 and, in one extra grid row, carries update(h = g - Lv): version h-1 -> h, which the
    // NEXT launch reads. Two buffers ping-pong (slot = (version + 1) & 1). No third stream, no cross-stream events.
    auto slot2 = [](int v) { return v < 0 ? 0 : ((v + 1) & 1); };
    // (g, h: group indices within the range — they drive the version slots; ga, ha: the absolute ones — they address panels)
    for (int g = 0; g < ngroups && !overlap && !mvp; g++) {
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
    if (!overlap && !mvp) launch_reduce(c, (g0 + ngroups - 1) * D * c->P, last_panels * c->P, sA, g0 + ngroups - 1);
    if (alone)
        if (int rc = launch_the_chain(sA)) return rc;
    for (int h = std::max(0, ngroups - Lv); h < ngroups && !overlap && !mvp; h++) { // the updates that had no later mat-vec to ride on
        upd_view uq = make_upd(c, (g0 + h) * D, std::min(np, (g0 + h) * D + D), slot2(h - 1), slot2(h), c->flags, g0 + h);
        if (dense_upd && c->layout == 8 && D <= 2) {
            uq.dense = 1;
            hipLaunchKernelGGL(k_update_dense, dim3((unsigned)(c->ld / 64)), dim3(64), 0, sA, c->ld, uq);
        } else hipLaunchKernelGGL(k_update, dim3(upd_blocks), dim3(256), 0, sA, c->ld, uq);
    }
    HB_HIP(hipEventRecord(c->ev_chain[0], sB));
    HB_HIP(hipStreamWaitEvent(sA, c->ev_chain[0], 0));
    if (warm || (fwd && !merged) || dense || warm_dense || fwd_persist) {
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
    const int sfin = overlap ? ((ngroups - 1 + 1) % (Lv + 1)) : slot2(ngroups - 1);
    if (sfin != 0) {
        HB_HIP(hipMemcpyAsync(c->r, c->r + (size_t)sfin * c->ld, sizeof(double) * c->ld, hipMemcpyDeviceToDevice, sA));
        HB_HIP(hipMemcpyAsync(c->r32, c->r32 + (size_t)sfin * c->ld, sizeof(float) * c->ld, hipMemcpyDeviceToDevice, sA));
    }
    if (model == 5 && last) {
        hipLaunchKernelGGL(k_bayesl_post, dim3((c->m + 255) / 256), dim3(256), 0, sA, c->d_in, c->m, c->m_offset, c->seed,
                           c->vx, c->g, c->vargL, 0);
        hipLaunchKernelGGL(k_sum_vec, dim3(1), dim3(1024), 0, sA, c->vargL, c->m, c->acc + HB_ACC_SUMVARGL);
    }
    if (last) hipLaunchKernelGGL(k_reduce_ru, dim3(16), dim3(64), 0, sA, c->r, c->u, c->n, c->acc, c->ru_ws, c->flags);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

// streams and events of the overlapped launch stream: created OUTSIDE any capture
static int overlap_resources(hb_ctx *c)
{
    auto mk = [&](hipStream_t *st) { return *st ? hipSuccess : hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
    HB_HIP(mk(&c->s_t2));
    HB_HIP(mk(&c->s_uk));
    HB_HIP(mk(&c->s_fk));
    const int need = c->npanels + 1;
    while ((int)c->ev_ot.size() < need) { hipEvent_t e; HB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_ot.push_back(e); }
    while ((int)c->ev_ou.size() < need) { hipEvent_t e; HB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_ou.push_back(e); }
    return HB_OK;
}

// the persistent mat-vec's buffers (outside any capture): digit planes and exponents for every residual version of a sweep, the hand-over counters
static int mvp_resources(hb_ctx *c)
{
    const int need = c->npanels + 2;
    if (c->rq_slots < need) {
        HB_HIP(hipStreamSynchronize(c->stream));
        int8_t *rq = nullptr;
        int *vexp = nullptr;
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&rq), (size_t)c->ld * HB_ND * need));
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&vexp), sizeof(int) * need));
        HB_HIP(hipMemset(rq, 0, (size_t)c->ld * HB_ND * need));
        HB_HIP(hipMemset(vexp, 0, sizeof(int) * need));
        (void)hipFree(c->rq);
        (void)hipFree(c->vexp);
        c->rq = rq;
        c->vexp = vexp;
        c->rq_slots = need;
        c->graph_model = -1; // (the captured sweeps hold the old pointers)
    }
    if (!c->mvp_ho) HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->mvp_ho), sizeof(unsigned) * ((size_t)(c->npanels + 2) * 65 + 64 * 32)));
    return HB_OK;
}

int hb_sweep_enqueue(hb_ctx *c, const hb_sweep_in *in, bool timed)
{
    if (int rc = hbk_set_timeout(c)) return rc;
    if (c->mvp && c->pipeline && c->precise == 2 && c->layout == 2 && (c->rq_slots < c->npanels + 2 || !c->mvp_ho))
        if (int rc = mvp_resources(c)) return rc;
    if (c->overlap && c->pipeline && !c->s_fk)
        if (int rc = overlap_resources(c)) return rc;
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
    hipLaunchKernelGGL(k_reduce_ru, dim3(16), dim3(64), 0, c->stream, c->r, c->u, c->n, c->acc, c->ru_ws, c->flags);
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
    // (round 6, timing experiment only — HB_TM_STREAMS=2: the launches alternate between two streams with nothing between them, i.e. two launches in flight:
    // what an overlapped launch stream could reach; the sums are not meaningful then)
    const int nstreams = getenv("HB_TM_STREAMS") ? std::max(1, std::min(2, atoi(getenv("HB_TM_STREAMS")))) : 1;
    if (nstreams == 2) {
        HB_HIP(hipEventRecord(c->ev_fork, c->stream));
        HB_HIP(hipStreamWaitEvent(c->s_upd, c->ev_fork, 0));
    }
    for (int gi = 0; gi < ngroups; gi++) {
        const int p0 = gi * D, p1 = std::min(c->npanels, p0 + D);
        launch_dot(c, p0 * c->P, (p1 - p0) * c->P, 0, (nstreams == 2 && (gi & 1)) ? c->s_upd : c->stream, as_pipeline != 0, nullptr,
                   gi > 0 ? (gi - 1) * D * c->P : 0, gi > 0 ? D * c->P : 0, gi);
    }
    if (nstreams == 2) {
        HB_HIP(hipEventRecord(c->ev_upd[0], c->s_upd));
        HB_HIP(hipStreamWaitEvent(c->stream, c->ev_upd[0], 0));
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
