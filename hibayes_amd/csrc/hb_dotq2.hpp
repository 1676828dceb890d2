// hb_dotq2.hpp — the exact fixed-point mat-vec on 2-BIT RESIDENT genotypes (the layout directly upstream of X is
// PLINK's 2 bits per genotype, reference src/read_bed.cpp:116-167; kept resident it is a quarter of the int8 bytes).
// Included by hb_kernels.hip after k_dotq (same views, same hand-offs, same block roles).
//
// Layout. Column-major, ld2 bytes per column (ld2 = 128 * ceil(n / 512)). A 32-bit word holds 16 consecutive individuals of one
// marker, value = genotype code (0..3), individual 16 w + 4 k + b in bits [8 b + 2 k, 8 b + 2 k + 1]: then
//     (word >> 2 k) & 0x03030303
// is four BYTES holding the genotypes of individuals 16 w + 4 k + 0..3 — exactly the operand v_dot4_i32_i8 wants beside the
// dword of the residual's digit plane for the same four individuals. One shift + one and per four genotypes (the shift is free
// for k = 0) against seven dot4: the expansion happens in registers, HBM and LDS only ever see the packed form.
// (The .bed file's own bit order — 00 -> 2, 10 -> 1, 11 -> 0, 01 -> NA, individual i in bits 2 (i mod 4) of byte i / 4 — would
// need two dot4 per plane (x = 2 - hi - lo) plus a plane sum; k_pack2 re-encodes once, after the major-genotype imputation the
// reference applies (read_bed.cpp:182-230) has removed the NA code.)
//
// One wave = 64 * CPL columns x NS stages of 512 individuals; lane = column (and column + 64 for CPL = 2: the seven digit reads
// of a 16-individual chunk — wave-uniform LDS broadcasts, the kernel's LDS traffic — then serve two columns). Per stage the
// tile (8 * CPL pieces of 1 KiB: 8 columns x 512 individuals each) and the seven digit planes (4 pieces: two planes each) arrive
// by LDS-DMA, double-buffered, counted vmcnt.
#pragma once

#ifndef Q2_DIAG
#define Q2_DIAG 0 /* timing diagnostics only (wrong results): 1 = no atomics at the tile's end, 2 = no dot4 (loads + reduction only), 3 = plain stores instead of the atomics */
#endif
#define Q2_RS 512   /* individuals per stage of the default shape (the padded column length is a multiple of it) */
// Q2_SCALED (round 4): the four genotypes of a byte are masked WITHOUT shifting them down — w & 0x03030303, w & 0x0c0c0c0c,
// w & 0x30303030 leave them scaled by 1, 4, 16, and (w >> 1) & 0x60606060 by 32 (0xc0 would be a negative int8) — and each scale
// sums into its own accumulator; the four are combined exactly at the tile's end (a sum of multiples of 4^k shifts back without
// loss). Five mask operations per 16 genotypes instead of seven, and four independent dot4 chains per plane instead of one.
// The scale-32 sums bound a tile: rows x 96 x 128 < 2^31, i.e. at most 174 000 individuals per tile (the host splits taller columns).
#ifndef Q2_SCALED
#define Q2_SCALED 1
#endif
#ifndef Q2_TWO_PER_SIMD
#define Q2_TWO_PER_SIMD 1 /* k_dotq2 allocates 176 VGPRs so that two of its waves fit a SIMD, not three (A/B on one box, twice: 295.8 / 295.9 sweeps/s without and 2000 tiles, 300.7 / 299.2 with and 1600; with and 2000: 277 — the launch no longer fits the chip at once) */
#endif
#ifndef Q2_AHEAD
#define Q2_AHEAD 2 /* chunks the digit reads run ahead of the dot4 that use them: two since round 4 (a ring of three register sets, the stage fully unrolled: 24.2 against 25.7 us per launch isolated, 22.6 against 23.2 in situ); 1 = the round-3 loop */
#endif

// RS: individuals per stage (512 or 256). A 1-KiB DMA piece holds 4096 / RS columns x RS individuals of the tile, or 1024 / RS
// digit planes x RS individuals.
template <int CPL, int RS>
__device__ __forceinline__ void dotq2_tile(const dq_view &v, char *smem, int b)
{
    constexpr int XPC = 4096 / RS, LPC = 64 / XPC;          // columns per tile piece, lanes per column in it
    constexpr int NXP = 64 * CPL / XPC;                      // tile pieces per stage
    constexpr int PPP = 1024 / RS, LPP = RS / 16;            // planes per digit piece, lanes per plane in it
    constexpr int NDP = (HB_ND + PPP - 1) / PPP;             // digit pieces per stage
    constexpr int XB = NXP * HBQ_SLOT, BUF = XB + NDP * 1024, PER = NXP + NDP;
    const int lane = threadIdx.x;
    const int cg = b % v.ncg, sp = b / v.ncg;
    if (b == 0 && lane == 0) *v.gexp_out = *v.vexp_in;
    const int st0 = sp * v.NS, st1 = min(v.nstages, st0 + v.NS);
    if (st0 >= st1) return;
    const int64_t ld2 = v.ld2, ld = v.ld;
    const uint8_t *xg = v.X2 + (int64_t)cg * (64 * CPL) * ld2;
    const unsigned voff = (unsigned)((lane / LPC) * ld2 + (lane % LPC) * 16);   // piece i: columns XPC i .. XPC i + XPC - 1
    unsigned doff[NDP];                                                          // digit piece j: planes PPP j .. (clamped to the last)
#pragma unroll
    for (int j = 0; j < NDP; j++) doff[j] = (unsigned)(min(j * PPP + lane / LPP, HB_ND - 1) * ld + (lane % LPP) * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    constexpr int NSC = Q2_SCALED ? 4 : 1;
    int acc[CPL][HB_ND][NSC];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int k = 0; k < HB_ND; k++)
#pragma unroll
            for (int q = 0; q < NSC; q++) acc[c][k][q] = 0;
    // (values that ARE wave-uniform, but that hipcc may keep on the vector unit when scalar registers run short)
    auto uni_p = [](const int8_t *p) {
        const unsigned long long u = (unsigned long long)(uintptr_t)p;
        return reinterpret_cast<const int8_t *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                                            (unsigned)__builtin_amdgcn_readfirstlane((int)u)));
    };
    auto issue = [&](int st, int buf) {
        const int8_t *xs = uni_p(reinterpret_cast<const int8_t *>(xg) + (int64_t)st * (RS / 4));
        const int8_t *ds = uni_p(v.rq + (int64_t)st * RS);
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)buf * BUF));
#pragma unroll
        for (int i = 0; i < NXP; i++) hbq_dma16<true>(voff, xs + (int64_t)(XPC * i) * ld2, dst + i * HBQ_SLOT);
#pragma unroll
        for (int j = 0; j < NDP; j++) hbq_dma16<false>(doff[j], ds, dst + XB + j * 1024);
    };
    issue(st0, 0);
    int buf = 0;
    for (int st = st0; st < st1; ++st) {
        if (st + 1 < st1) {
            issue(st + 1, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PER) : "memory"); // everything but the stage just requested has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char *bp = smem + buf * BUF;
        const hb_v4i *px[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) px[c] = reinterpret_cast<const hb_v4i *>(bp + (lane / XPC + (64 / XPC) * c) * HBQ_SLOT + (lane % XPC) * (RS / 4));
        const char *pd = bp + XB;
        // software-pipelined by hand: the seven digit reads of chunk ch + 1 (and the column's next 16 bytes) are issued BEFORE
        // the 28 dot4 of chunk ch — a wave parks 78 % of its cycles otherwise (rocprofv3 SQ_WAIT_ANY), every chunk waiting out
        // its own LDS round trip behind the other waves' reads
#if Q2_AHEAD >= 2
        // (the digit reads Q2_AHEAD chunks ahead of the dot4 that use them, a ring of Q2_AHEAD + 1 register sets, the stage fully unrolled)
        hb_v4i dr[Q2_AHEAD + 1][HB_ND], xq[CPL], xqn[CPL];
#pragma unroll
        for (int a = 0; a < Q2_AHEAD; a++)
#pragma unroll
            for (int k = 0; k < HB_ND; k++) dr[a][k] = *reinterpret_cast<const hb_v4i *>(pd + k * RS + a * 16);
#pragma unroll
        for (int c = 0; c < CPL; c++) xqn[c] = px[c][0];
#pragma unroll
        for (int ch = 0; ch < RS / 16; ch++) {
            const int w = ch & 3;
            if (w == 0) {
#pragma unroll
                for (int c = 0; c < CPL; c++) xq[c] = xqn[c];
            }
            const int chn = min(ch + Q2_AHEAD, RS / 16 - 1);
#pragma unroll
            for (int k = 0; k < HB_ND; k++) dr[(ch + Q2_AHEAD) % (Q2_AHEAD + 1)][k] = *reinterpret_cast<const hb_v4i *>(pd + k * RS + chn * 16);
            if (w == 2) {
#pragma unroll
                for (int c = 0; c < CPL; c++) xqn[c] = px[c][min((ch >> 2) + 1, RS / 64 - 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
            hb_v4i d[HB_ND];
#pragma unroll
            for (int k = 0; k < HB_ND; k++) d[k] = dr[ch % (Q2_AHEAD + 1)][k];
#else
        hb_v4i dn[HB_ND], xq[CPL], xqn[CPL];
#pragma unroll
        for (int k = 0; k < HB_ND; k++) dn[k] = *reinterpret_cast<const hb_v4i *>(pd + k * RS);
#pragma unroll
        for (int c = 0; c < CPL; c++) xqn[c] = px[c][0];
#pragma unroll 4
        for (int ch = 0; ch < RS / 16; ch++) { // 16 individuals per chunk; four chunks per 16-byte read of a column
            hb_v4i d[HB_ND];
#pragma unroll
            for (int k = 0; k < HB_ND; k++) d[k] = dn[k];
            const int w = ch & 3;
            if (w == 0) {
#pragma unroll
                for (int c = 0; c < CPL; c++) xq[c] = xqn[c];
            }
            const int chn = min(ch + 1, RS / 16 - 1);
#pragma unroll
            for (int k = 0; k < HB_ND; k++) dn[k] = *reinterpret_cast<const hb_v4i *>(pd + k * RS + chn * 16);
            if (w == 3) {
#pragma unroll
                for (int c = 0; c < CPL; c++) xqn[c] = px[c][min((ch >> 2) + 1, RS / 64 - 1)];
            }
#endif
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const unsigned xw = (unsigned)(w == 0 ? xq[c].x : w == 1 ? xq[c].y : w == 2 ? xq[c].z : xq[c].w);
#if Q2_SCALED
                const int m0 = (int)(xw & 0x03030303u), m1 = (int)(xw & 0x0c0c0c0cu), m2 = (int)(xw & 0x30303030u),
                          m3 = (int)((xw >> 1) & 0x60606060u);
#pragma unroll
                for (int k = 0; k < HB_ND; k++) {
                    acc[c][k][0] = __builtin_amdgcn_sdot4(m0, d[k].x, acc[c][k][0], false);
                    acc[c][k][1] = __builtin_amdgcn_sdot4(m1, d[k].y, acc[c][k][1], false);
                    acc[c][k][2] = __builtin_amdgcn_sdot4(m2, d[k].z, acc[c][k][2], false);
                    acc[c][k][3] = __builtin_amdgcn_sdot4(m3, d[k].w, acc[c][k][3], false);
                }
#else
                const int m0 = (int)(xw & 0x03030303u), m1 = (int)((xw >> 2) & 0x03030303u), m2 = (int)((xw >> 4) & 0x03030303u),
                          m3 = (int)((xw >> 6) & 0x03030303u);
#pragma unroll
                for (int k = 0; k < HB_ND; k++) {
                    acc[c][k][0] = __builtin_amdgcn_sdot4(m0, d[k].x, acc[c][k][0], false);
                    acc[c][k][0] = __builtin_amdgcn_sdot4(m1, d[k].y, acc[c][k][0], false);
                    acc[c][k][0] = __builtin_amdgcn_sdot4(m2, d[k].z, acc[c][k][0], false);
                    acc[c][k][0] = __builtin_amdgcn_sdot4(m3, d[k].w, acc[c][k][0], false);
                }
#endif
            }
        }
        buf ^= 1;
    }
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int k = 0; k < HB_ND; k++) {
            const int tot = Q2_SCALED ? acc[c][k][0] + (acc[c][k][NSC > 1 ? 1 : 0] >> 2) + (acc[c][k][NSC > 2 ? 2 : 0] >> 4) + (acc[c][k][NSC > 3 ? 3 : 0] >> 5) : acc[c][k][0];
            if (Q2_DIAG == 1) { if (tot == 0x12345678) v.accq[0] = 1; } // (timing diagnostic: the tile without its closing atomics)
            else if (Q2_DIAG == 3) v.accq[(int64_t)k * v.accstride + cg * (64 * CPL) + c * 64 + lane] = (long long)tot; // (... plain stores in their place)
            else
            __hip_atomic_fetch_add(v.accq + (int64_t)k * v.accstride + cg * (64 * CPL) + c * 64 + lane, (long long)tot, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
}

template <int CPL, int RS>
__global__ __launch_bounds__(64) void k_dotq2(dq_view v, upd_view uq)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if Q2_TWO_PER_SIMD
    // (a register the kernel does not need, named so that its allocation passes 170: the hardware then fits two of its waves on a SIMD,
    // not three — a launch's tiles are VALU-bound, a SIMD with three of them ends half a tile time after one with two)
    asm volatile("" ::: "v175");
#endif
    unsigned long long t0 = 0;
    if (v.stamp || v.ldiag) t0 = wall_clock64();
    int b = blockIdx.x;
    if (b < v.nfin) { // (finalize blocks first: see dotq_block)
        const int col = b * 64 + threadIdx.x;
        if (col < v.fin_ncols) hbq_finalize(v.fin_acc, v.accstride, col, *v.fin_exp, v.fin_out);
    } else if (b < v.nfin + v.nupd) { // residual update of an earlier group (2-bit columns: upd_view.X2), lists staged in the tile buffers
        b -= v.nfin;
        update_rows(v.ld, uq, b, reinterpret_cast<int *>(smem), reinterpret_cast<double *>(smem + 2048), reinterpret_cast<int *>(smem + 2048 + 4096), v.ldiag ? v.ldiag + 3 : nullptr);
    } else {
        dotq2_tile<CPL, RS>(v, smem, b - v.nupd - v.nfin);
    }
    if (v.stamp && threadIdx.x == 0) {
        v.stamp[2 * (size_t)blockIdx.x] = t0;
        v.stamp[2 * (size_t)blockIdx.x + 1] = wall_clock64();
    }
    if (v.ldiag) hb_ldiag_note(v.ldiag, t0);
}
static constexpr int q2_lds(int cpl, int rs) { return 2 * ((64 * cpl / (4096 / rs)) * HBQ_SLOT + ((HB_ND + 1024 / rs - 1) / (1024 / rs)) * 1024); }

// ---------------------------------------------------------------------------------------------
// k_dotq2m: the same product on the MATRIX CORES — an A/B kernel, not the default (hb_ctx_set_matvec_kernel(c, 2) /
// HB_DOTQ2_KIND=2; BASELINE's north_star asks for coalesced loads and wavefront reductions for this mat-vec, "no MFMA").
// Once the residual is seven int8 digit planes the panel product is a skinny integer GEMM, [columns x individuals] x
// [individuals x 7], and k_dotq2 above is bound by what the VALU formulation costs per genotype: 35 vector instructions and
// 7 KiB of broadcast LDS reads per 16 individuals of 64 columns (LDS return bandwidth 17 us, v_dot4 issue 12 us per
// 3584-column launch at n = 50k, against 8 us for the launch's 44.8 MB at the HBM rate). Here one v_mfma_i32_16x16x64_i8 does
// 16 columns x 64 individuals x 16 "planes" (7 real ones) — the same exact int8 x int8 -> int32 products, summed in another
// order, so the results are the SAME INTEGERS (tests/test_gpu_kernels.py::test_two_bit_layout_*).
//   * Operands. The instruction sums A[i][kk] B[kk][j] over 64 positions kk = (lane group kb, register r, byte b); which
//     individual sits at a position is ours to choose as long as A and B agree. Lane (m, kb) of a column tile holds the 16
//     bytes = four words w_0..w_3 of column m for individuals 64 kb .. 64 kb + 63 of the 256-individual stage; sub-position k
//     of word w_r (bits [8 b + 2 k, +1]) is individual 64 kb + 16 r + 4 k + b. Masking WITHOUT shifting, w_r & (0x03030303 << 2 k),
//     leaves the genotypes of sub-position k scaled by 4^k in the four bytes of register r (k = 3: (w_r >> 1) & 0x60606060, scaled
//     by 32 — 0xC0 would be a negative int8): A_k. Its partner B_k holds the digits of the same individuals: register r =
//     dword 4 r + k of the lane's 64 digit bytes of plane n — pure register selection. Four MFMAs per column tile and stage,
//     one accumulator set per scale, combined exactly at the tile's end (sums of multiples of 4^k shift back without loss).
//   * LDS. Tile and digit planes arrive by LDS-DMA exactly as in k_dotq2 (double-buffered, counted vmcnt); the DMA lane order
//     is chosen so that every operand read is one conflict-free ds_read_b128: tile piece i (16 columns) is stored [kb][m] —
//     lane l of the wave reads slot l —, digit piece j (planes 4 j .. 4 j + 3) is stored [16-byte chunk][plane], pieces 1088
//     bytes apart so that planes 4..6 land 16 banks away from planes 0..3. 8 reads per stage against 116 in k_dotq2.
//   * Cost per stage of 64 columns x 256 individuals: 16 MFMA + 80 mask operations + 8 LDS reads, against 448 v_dot4 + 112
//     mask operations + 116 LDS reads: the launch becomes HBM-bound.
// ---------------------------------------------------------------------------------------------
#define Q2M_RS 256
#define Q2M_DSTRIDE 1088
#ifndef Q2M_NBUF
#define Q2M_NBUF 3 /* stage buffers: NBUF - 1 (super-)stages in flight ahead of the one being multiplied (a stage computes in ~0.3 us, a loaded round trip takes ~2) */
#endif
#ifndef Q2M_NODIG
#define Q2M_NODIG 0 /* TIMING DIAGNOSTIC ONLY (wrong results): dotq2m512_tile requests no digit pieces at all — the round-5 verdict's test of "the launch is bound by LDS-DMA ingest, a third of which is digits every single-wave tile re-reads" (profiles/r06_q2m_digits.txt) */
#endif
#ifndef Q2M_NT
#define Q2M_NT 1 /* the genotype tile's DMA pieces carry the non-temporal hint (k_dotq: so that the digit planes and the chain's rows stay in L2) */
#endif
static_assert(Q2M_NBUF >= 2 && Q2M_NBUF <= 6, "the counted waits of dotq2m_tile cover up to five stages in flight");
// Shape (round 5). CT = column tiles of 16 per wave (4: 64 columns, the round-4 shape; 8: 128; 16: 256) — the stage's digit planes
// (1.75 KB, re-read from L2 by every wave) then serve CT * 16 columns, and what the launch moves through the compute units'
// LDS-DMA path, the genotypes PLUS those digit bytes, is what bounds it: 63 MB per 3584-column launch at CT = 4 (44.8 + 18.4),
// 54 at CT = 8, 49 at CT = 16. G = 256-individual stages requested together (2: both halves of every 128-byte line of a column
// are asked for back to back). One accumulator set per scale (SC = true, 16 registers per column tile) or one in all (SC = false:
// the genotypes shifted down to one scale, 7 mask / shift operations per register instead of 5, 4 registers per column tile).
template <int CT, int G>
static constexpr int q2m_lds() { return Q2M_NBUF * G * (CT * HBQ_SLOT + 2 * Q2M_DSTRIDE); }

template <int CT, int G, bool SC>
__device__ __forceinline__ void dotq2m_tile(const dq_view &v, char *smem, int b)
{
    constexpr int XB = CT * HBQ_SLOT, BUF = XB + 2 * Q2M_DSTRIDE, PER = G * (CT + 2), NSC = SC ? 4 : 1;
    static_assert((Q2M_NBUF - 1) * PER <= 63, "the in-flight DMA pieces must fit the 6-bit vmcnt");
    const int lane = threadIdx.x;
    const int cg = b % v.ncg, sp = b / v.ncg;
    if (b == 0 && lane == 0) *v.gexp_out = *v.vexp_in;
    const int st0 = sp * v.NS, st1 = min(v.nstages, st0 + v.NS);
    if (st0 >= st1) return;
    const int64_t ld2 = v.ld2, ld = v.ld;
    const uint8_t *xg = v.X2 + (int64_t)cg * (16 * CT) * ld2;
    const int m = lane & 15, kb = lane >> 4;
    // DMA sources: tile piece i = columns 16 i + m, 16-byte chunk kb of the stage's 64 bytes; digit piece j = plane 4 j + (lane & 3)
    // (clamped to the last plane), 16-byte chunk lane >> 2 of the stage's 256 digit bytes
    const unsigned voff = (unsigned)(m * ld2 + kb * 16);
    unsigned doff[2];
#pragma unroll
    for (int j = 0; j < 2; j++) doff[j] = (unsigned)(min(4 * j + (lane & 3), HB_ND - 1) * ld + (lane >> 2) * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    auto uni_p = [](const int8_t *p) {
        const unsigned long long u = (unsigned long long)(uintptr_t)p;
        return reinterpret_cast<const int8_t *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                                            (unsigned)__builtin_amdgcn_readfirstlane((int)u)));
    };
    // super-stage ss = stages st0 + G ss .. + G - 1 (the last may be short: its missing stages re-request the tile's last stage, so
    // that every super-stage is exactly PER pieces and the counted waits stay exact; they are not multiplied)
    const int nss = (st1 - st0 + G - 1) / G;
    auto issue = [&](int ss, int buf) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int st = min(st0 + ss * G + g, st1 - 1);
            const int8_t *xs = uni_p(reinterpret_cast<const int8_t *>(xg) + (int64_t)st * (Q2M_RS / 4));
            const int8_t *ds = uni_p(v.rq + (int64_t)st * Q2M_RS);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(buf * G + g) * BUF));
#pragma unroll
            for (int i = 0; i < CT; i++) hbq_dma16<Q2M_NT != 0>(voff, xs + (int64_t)(16 * i) * ld2, dst + i * HBQ_SLOT);
#pragma unroll
            for (int j = 0; j < 2; j++) hbq_dma16<false>(doff[j], ds, dst + XB + j * Q2M_DSTRIDE);
        }
    };
    hb_v4i C[CT][NSC]; // [column tile][scale 4^k, k = 3: 32]
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int k = 0; k < NSC; k++) C[ct][k] = hb_v4i{0, 0, 0, 0};
    // this lane's digit reads: plane n = min(lane & 15, 6) (the output columns 7..15 of the instruction are never looked at)
    const int n = min(m, HB_ND - 1);
    const unsigned dlane = (unsigned)((n >> 2) * Q2M_DSTRIDE + (kb * 16 + (n & 3)) * 16); // + 64 r: chunk 4 kb + r
#pragma unroll
    for (int a = 0; a < Q2M_NBUF - 1; a++)
        if (a < nss) issue(a, a);
    int buf = 0;
    for (int ss = 0; ss < nss; ++ss) {
        if (ss + Q2M_NBUF - 1 < nss) issue(ss + Q2M_NBUF - 1, (buf + Q2M_NBUF - 1) % Q2M_NBUF);
        // super-stages still in flight behind this one: the counted wait lets exactly those stay outstanding
        const int ahead = min(Q2M_NBUF - 1, nss - 1 - ss);
        if (ahead >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(Q2M_NBUF > 5 ? 5 * PER : 0) : "memory");
        else if (ahead == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(Q2M_NBUF > 4 ? 4 * PER : 0) : "memory");
        else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(Q2M_NBUF > 3 ? 3 * PER : 0) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(Q2M_NBUF > 2 ? 2 * PER : 0) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (st0 + ss * G + g >= st1) break;
            const char *bp = smem + (buf * G + g) * BUF;
            hb_v4i D[4];
#pragma unroll
            for (int r = 0; r < 4; r++) D[r] = *reinterpret_cast<const hb_v4i *>(bp + XB + dlane + r * 64);
            const hb_v4i B0 = hb_v4i{D[0].x, D[1].x, D[2].x, D[3].x}, B1 = hb_v4i{D[0].y, D[1].y, D[2].y, D[3].y},
                         B2 = hb_v4i{D[0].z, D[1].z, D[2].z, D[3].z}, B3 = hb_v4i{D[0].w, D[1].w, D[2].w, D[3].w};
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                const hb_v4i w = *reinterpret_cast<const hb_v4i *>(bp + ct * HBQ_SLOT + lane * 16);
                if (SC) {
                    const hb_v4i a0 = w & 0x03030303, a1 = w & 0x0c0c0c0c, a2 = w & 0x30303030;
                    const hb_v4i a3 = hb_v4i{(int)((unsigned)w.x >> 1), (int)((unsigned)w.y >> 1), (int)((unsigned)w.z >> 1), (int)((unsigned)w.w >> 1)} & 0x60606060;
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, B0, C[ct][0], 0, 0, 0);
                    C[ct][NSC > 1 ? 1 : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, B1, C[ct][NSC > 1 ? 1 : 0], 0, 0, 0);
                    C[ct][NSC > 2 ? 2 : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, B2, C[ct][NSC > 2 ? 2 : 0], 0, 0, 0);
                    C[ct][NSC > 3 ? 3 : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, B3, C[ct][NSC > 3 ? 3 : 0], 0, 0, 0);
                } else {
                    auto shr = [](const hb_v4i &x, int sh) {
                        return hb_v4i{(int)((unsigned)x.x >> sh), (int)((unsigned)x.y >> sh), (int)((unsigned)x.z >> sh), (int)((unsigned)x.w >> sh)};
                    };
                    const hb_v4i a0 = w & 0x03030303, a1 = shr(w, 2) & 0x03030303, a2 = shr(w, 4) & 0x03030303, a3 = shr(w, 6) & 0x03030303;
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, B0, C[ct][0], 0, 0, 0);
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, B1, C[ct][0], 0, 0, 0);
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, B2, C[ct][0], 0, 0, 0);
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, B3, C[ct][0], 0, 0, 0);
                }
            }
        }
        // (the next iteration's DMA overwrites the buffer read here: every LDS read of it has returned — the MFMAs consumed them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        buf = (buf + 1 == Q2M_NBUF) ? 0 : buf + 1;
    }
    // the instruction's result layout: lane l, register r = C[row 4 (l / 16) + r][column l % 16], i.e. genotype column
    // 4 kb + r of the tile, plane m. The sums of scale 4^k are multiples of it: the shifts are exact.
    // Turned through LDS (the tile buffers are free now; one wave: no barrier) so that a lane ends up with ONE column and the
    // closing atomics of the wave each cover 64 consecutive columns — straight from the result layout an atomic instruction
    // would touch 28 separate lines (7 planes x 4 lane groups): measured 44 us per launch against 22 with a quarter of the tiles.
    int *tr = reinterpret_cast<int *>(smem); // [plane][16 CT columns]
    if (m < HB_ND) {
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const hb_v4i tot = SC ? C[ct][0] + (C[ct][NSC > 1 ? 1 : 0] >> 2) + (C[ct][NSC > 2 ? 2 : 0] >> 4) + (C[ct][NSC > 3 ? 3 : 0] >> 5) : C[ct][0];
            *reinterpret_cast<hb_v4i *>(tr + m * (16 * CT) + ct * 16 + 4 * kb) = tot;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < HB_ND; k++)
#pragma unroll
        for (int q = 0; q < CT / 4; q++) {
            const int tv = tr[k * (16 * CT) + q * 64 + lane];
            if (Q2_DIAG == 1) { if (tv == 0x12345678) v.accq[0] = 1; } // (timing diagnostic: the tile without its closing atomics)
            else if (Q2_DIAG == 3) v.accq[(int64_t)k * v.accstride + cg * (16 * CT) + q * 64 + lane] = (long long)tv; // (... with plain stores in their place)
            else
            __hip_atomic_fetch_add(v.accq + (int64_t)k * v.accstride + cg * (16 * CT) + q * 64 + lane, (long long)tv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
}

// The same product on 512-individual stages whose DMA pieces are WHOLE 128-byte lines (round 5 A/B, shape G = 0): dotq2m_tile above asks
// for a column in 64-byte halves, the second half one stage after the first; here a piece is 8 columns x 128 bytes (lane l: column l / 8,
// 16-byte chunk l % 8 — k_dotq2's mapping) and a stage is multiplied in two halves of 256 individuals. The operand read of lane (m, kb), half h,
// is chunk 4 h + kb of column m of the column tile: a 4-way bank conflict on the read (eight 128-byte rows 128 bytes apart), paid for with LDS
// cycles the kernel does not need. 64 columns per wave.
template <bool SC, bool SWZ>
__device__ __forceinline__ void dotq2m512_tile(const dq_view &v, char *smem, int b)
{
    // SWZ: the DMA lanes are dealt so that the operand reads are bank-conflict free — tile piece: lane l = column l % 8, chunk l / 8 (slot = chunk * 8 + column:
    // the 16 lanes of a read group take two runs of 128 consecutive bytes, the pieces 1152 bytes apart = 32 banks); digit piece j: lane l = plane l % 8,
    // chunk 8 j + l / 8 (slot = chunk * 8 + plane: the seven planes of a read group are neighbours). The eight lanes that share a 128-byte line are then
    // eight apart; without SWZ they are neighbours and the reads conflict four- and seven-fold.
    constexpr int XSL = SWZ ? 1152 : HBQ_SLOT;
    constexpr int NXP = 8, NDP = 4, XB = NXP * XSL, BUF = XB + NDP * 1024, PER = NXP + (Q2M_NODIG ? 0 : NDP), NSC = SC ? 4 : 1;
    static_assert((Q2M_NBUF - 1) * PER <= 63, "the in-flight DMA pieces must fit the 6-bit vmcnt");
    const int lane = threadIdx.x;
    const int cg = b % v.ncg, sp = b / v.ncg;
    if (b == 0 && lane == 0) *v.gexp_out = *v.vexp_in;
    const int st0 = sp * v.NS, st1 = min(v.nstages, st0 + v.NS);
    if (st0 >= st1) return;
    const int64_t ld2 = v.ld2, ld = v.ld;
    const uint8_t *xg = v.X2 + (int64_t)cg * 64 * ld2;
    const int m = lane & 15, kb = lane >> 4;
    const unsigned voff = SWZ ? (unsigned)((lane & 7) * ld2 + (lane >> 3) * 16) : (unsigned)((lane >> 3) * ld2 + (lane & 7) * 16); // tile piece i: 8 columns x the stage's 128 bytes
    unsigned doff[NDP];                                                                // digit piece j: planes 2 j + lane / 32 (clamped), chunk lane % 32 of the stage's 512 bytes
#pragma unroll
    for (int j = 0; j < NDP; j++)
        doff[j] = SWZ ? (unsigned)(min(lane & 7, HB_ND - 1) * ld + (8 * j + (lane >> 3)) * 16) : (unsigned)(min(2 * j + (lane >> 5), HB_ND - 1) * ld + (lane & 31) * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    auto uni_p = [](const int8_t *p) {
        const unsigned long long u = (unsigned long long)(uintptr_t)p;
        return reinterpret_cast<const int8_t *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                                            (unsigned)__builtin_amdgcn_readfirstlane((int)u)));
    };
    auto issue = [&](int st, int buf) {
        const int8_t *xs = uni_p(reinterpret_cast<const int8_t *>(xg) + (int64_t)st * 128);
        const int8_t *ds = uni_p(v.rq + (int64_t)st * 512);
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)buf * BUF));
#pragma unroll
        for (int i = 0; i < NXP; i++) hbq_dma16<Q2M_NT != 0>(voff, xs + (int64_t)(8 * i) * ld2, dst + i * XSL);
        if constexpr (!Q2M_NODIG) {
#pragma unroll
            for (int j = 0; j < NDP; j++) hbq_dma16<false>(doff[j], ds, dst + XB + j * 1024);
        }
    };
    hb_v4i C[4][NSC];
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int k = 0; k < NSC; k++) C[ct][k] = hb_v4i{0, 0, 0, 0};
    const int n = min(m, HB_ND - 1);
    // operand reads: column 16 ct + m = piece 2 ct + (m >> 3), row (m & 7) of it, chunk 4 h + kb; digits of plane n: piece n / 2, lane slot (n & 1) * 32 + 16 h + 4 kb + r
    // (+ 4 h chunks for the second half; SWZ: a chunk step is 8 slots)
    const unsigned xlane = SWZ ? (unsigned)((m >> 3) * XSL + (kb * 8 + (m & 7)) * 16) : (unsigned)((m >> 3) * XSL + ((m & 7) * 8 + kb) * 16);
    const unsigned dlane = SWZ ? (unsigned)(XB + n * 16) : (unsigned)(XB + (n >> 1) * 1024 + ((n & 1) * 32 + 4 * kb) * 16);
#pragma unroll
    for (int a = 0; a < Q2M_NBUF - 1; a++)
        if (st0 + a < st1) issue(st0 + a, a);
    int buf = 0;
    for (int st = st0; st < st1; ++st) {
        if (st + Q2M_NBUF - 1 < st1) issue(st + Q2M_NBUF - 1, (buf + Q2M_NBUF - 1) % Q2M_NBUF);
        const int ahead = min(Q2M_NBUF - 1, st1 - 1 - st);
        if (ahead >= 5) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(Q2M_NBUF > 5 ? 5 * PER : 0) : "memory");
        else if (ahead == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(Q2M_NBUF > 4 ? 4 * PER : 0) : "memory");
        else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(Q2M_NBUF > 3 ? 3 * PER : 0) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(Q2M_NBUF > 2 ? 2 * PER : 0) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char *bp = smem + buf * BUF;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            hb_v4i D[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int ch = 16 * h + 4 * kb + r; // SWZ: chunk ch of the stage = piece ch / 8, slot (ch % 8) * 8 + plane
                D[r] = SWZ ? *reinterpret_cast<const hb_v4i *>(bp + dlane + (ch >> 3) * 1024 + (ch & 7) * 128) : *reinterpret_cast<const hb_v4i *>(bp + dlane + (16 * h + r) * 16);
            }
            const hb_v4i B0 = hb_v4i{D[0].x, D[1].x, D[2].x, D[3].x}, B1 = hb_v4i{D[0].y, D[1].y, D[2].y, D[3].y},
                         B2 = hb_v4i{D[0].z, D[1].z, D[2].z, D[3].z}, B3 = hb_v4i{D[0].w, D[1].w, D[2].w, D[3].w};
#pragma unroll
            for (int ct = 0; ct < 4; ct++) {
                const hb_v4i w = *reinterpret_cast<const hb_v4i *>(bp + 2 * ct * XSL + xlane + (SWZ ? 4 * h * 128 : 4 * h * 16));
                if (SC) {
                    const hb_v4i a0 = w & 0x03030303, a1 = w & 0x0c0c0c0c, a2 = w & 0x30303030;
                    const hb_v4i a3 = hb_v4i{(int)((unsigned)w.x >> 1), (int)((unsigned)w.y >> 1), (int)((unsigned)w.z >> 1), (int)((unsigned)w.w >> 1)} & 0x60606060;
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, B0, C[ct][0], 0, 0, 0);
                    C[ct][NSC > 1 ? 1 : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, B1, C[ct][NSC > 1 ? 1 : 0], 0, 0, 0);
                    C[ct][NSC > 2 ? 2 : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, B2, C[ct][NSC > 2 ? 2 : 0], 0, 0, 0);
                    C[ct][NSC > 3 ? 3 : 0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, B3, C[ct][NSC > 3 ? 3 : 0], 0, 0, 0);
                } else {
                    auto shr = [](const hb_v4i &x, int sh) {
                        return hb_v4i{(int)((unsigned)x.x >> sh), (int)((unsigned)x.y >> sh), (int)((unsigned)x.z >> sh), (int)((unsigned)x.w >> sh)};
                    };
                    const hb_v4i a0 = w & 0x03030303, a1 = shr(w, 2) & 0x03030303, a2 = shr(w, 4) & 0x03030303, a3 = shr(w, 6) & 0x03030303;
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, B0, C[ct][0], 0, 0, 0);
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, B1, C[ct][0], 0, 0, 0);
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, B2, C[ct][0], 0, 0, 0);
                    C[ct][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, B3, C[ct][0], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        buf = (buf + 1 == Q2M_NBUF) ? 0 : buf + 1;
    }
    int *tr = reinterpret_cast<int *>(smem); // [plane][64 columns]
    if (m < HB_ND) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
            const hb_v4i tot = SC ? C[ct][0] + (C[ct][NSC > 1 ? 1 : 0] >> 2) + (C[ct][NSC > 2 ? 2 : 0] >> 4) + (C[ct][NSC > 3 ? 3 : 0] >> 5) : C[ct][0];
            *reinterpret_cast<hb_v4i *>(tr + m * 64 + ct * 16 + 4 * kb) = tot;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < HB_ND; k++)
        __hip_atomic_fetch_add(v.accq + (int64_t)k * v.accstride + cg * 64 + lane, (long long)tr[k * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool SWZ>
static constexpr int q2m512_lds() { return Q2M_NBUF * (8 * (SWZ ? 1152 : HBQ_SLOT) + 4 * 1024); }

template <int CT, int G, bool SC>
__global__ __launch_bounds__(64) void k_dotq2m(dq_view v, upd_view uq)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t0 = 0;
    if (v.stamp || v.ldiag) t0 = wall_clock64();
    int b = blockIdx.x;
    if (b < v.nfin) { // (finalize blocks first: see dotq_block)
        const int col = b * 64 + threadIdx.x;
        if (col < v.fin_ncols) hbq_finalize(v.fin_acc, v.accstride, col, *v.fin_exp, v.fin_out);
    } else if (b < v.nfin + v.nupd) {
        b -= v.nfin;
        update_rows(v.ld, uq, b, reinterpret_cast<int *>(smem), reinterpret_cast<double *>(smem + 2048), reinterpret_cast<int *>(smem + 2048 + 4096), v.ldiag ? v.ldiag + 3 : nullptr);
    } else {
        if constexpr (G == 0) dotq2m512_tile<SC, false>(v, smem, b - v.nupd - v.nfin);
        else if constexpr (G == 3) dotq2m512_tile<SC, true>(v, smem, b - v.nupd - v.nfin);
        else dotq2m_tile<CT, G, SC>(v, smem, b - v.nupd - v.nfin);
    }
    if (v.stamp && threadIdx.x == 0) {
        v.stamp[2 * (size_t)blockIdx.x] = t0;
        v.stamp[2 * (size_t)blockIdx.x + 1] = wall_clock64();
    }
    if (v.ldiag) hb_ldiag_note(v.ldiag, t0);
}

// ---------------------------------------------------------------------------------------------
// k_dotq2r: the same product with the INDIVIDUALS across the lanes and no LDS at all.
// The lane = column tile above spends its time parked: seven wave-uniform (broadcast) LDS reads per 16 individuals, each a
// queue behind every other wave's, then 28 v_dot4 (4 cycles each on gfx950, measured: tools/dot4_rate.hip) — 78 % of the wave
// cycles wait (rocprofv3 --pmc SQ_WAIT_ANY / SQ_WAVE_CYCLES), and the LDS budget of the double-buffered tile caps the waves that
// could hide it. Here a lane owns 64 consecutive individuals of a 4096-individual row block: their 7 x 64 digit bytes live in
// 112 registers for the whole tile (distinct per lane: nothing is broadcast), a column is ONE coalesced 1-KiB global load per
// wave (16 bytes = 64 individuals per lane, four columns in flight ahead), and the work between two waits is 4 x (16 mask
// operations + 112 v_dot4). What it costs is a cross-lane reduction: 28 sums (4 columns x 7 planes) per batch, folded by a
// halving butterfly (31 exchanges) so that lane v ends up with the total of sum v — about a fifth on top of the dot4 stream.
// Integer sums are order-independent: the results are the same exact integers as k_dotq's and k_dotq2's.
// ---------------------------------------------------------------------------------------------
#define Q2R_RB 4096 /* individuals per row block (64 lanes x 64) */
#define Q2R_CB 4    /* columns per batch */
#ifndef Q2R_PF
#define Q2R_PF 2    /* batches requested ahead */
#endif

__device__ __forceinline__ void dotq2r_tile(const dq_view &v, int b)
{
    const int lane = threadIdx.x;
    const int cgi = b % v.ncg, rb = b / v.ncg; // column group (v.NS columns each), row block
    if (b == 0 && lane == 0) *v.gexp_out = *v.vexp_in;
    const int64_t ld2 = v.ld2, ld = v.ld;
    const int NC = v.NS;
    const int64_t rowbyte = (int64_t)rb * (Q2R_RB / 4) + lane * 16; // this lane's 16 bytes inside a packed column
    const bool valid = rowbyte < ld2;
    const uint8_t *xcol = v.X2 + (int64_t)cgi * NC * ld2 + (valid ? rowbyte : 0);
    // the lane's digits: plane p, dword q = individuals 4 q .. 4 q + 3 of its 64 (past the column's end: whatever is there,
    // the genotypes are zero)
    int dig[HB_ND][16];
    {
        const int64_t r0 = min((int64_t)rb * Q2R_RB + lane * 64, ld - 64);
#pragma unroll
        for (int p = 0; p < HB_ND; p++) {
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
                const hb_v4i t = *reinterpret_cast<const hb_v4i *>(v.rq + (int64_t)p * ld + r0 + q4 * 16);
                dig[p][4 * q4] = t.x; dig[p][4 * q4 + 1] = t.y; dig[p][4 * q4 + 2] = t.z; dig[p][4 * q4 + 3] = t.w;
            }
        }
    }
    // columns in flight: Q2R_PF batches of Q2R_CB ahead of the one being computed (a load takes ~2 us beside the other
    // waves' streams, a batch computes in ~1 us: one batch ahead leaves every batch waiting)
    hb_v4i xr[Q2R_PF + 1][Q2R_CB];
    auto fetch = [&](int slot, int cfirst) {
#pragma unroll
        for (int c = 0; c < Q2R_CB; c++)
            xr[slot][c] = __builtin_nontemporal_load(reinterpret_cast<const hb_v4i *>(xcol + (int64_t)min(cfirst + c, NC - 1) * ld2));
    };
#pragma unroll
    for (int f = 0; f < Q2R_PF; f++) fetch(f, f * Q2R_CB);
    // (the ring is indexed statically: the batch loop is unrolled Q2R_PF + 1 times)
    for (int cbase = 0; cbase < NC; cbase += (Q2R_PF + 1) * Q2R_CB) {
#pragma unroll
      for (int u = 0; u <= Q2R_PF; u++) {
        const int c0 = cbase + u * Q2R_CB;
        if (c0 >= NC) break;
        fetch((u + Q2R_PF) % (Q2R_PF + 1), c0 + Q2R_PF * Q2R_CB); // (past the tile's end: the last column again, unused)
        hb_v4i x[Q2R_CB];
#pragma unroll
        for (int c = 0; c < Q2R_CB; c++) x[c] = xr[u][c];
        int a[32];
#pragma unroll
        for (int i = 0; i < 32; i++) a[i] = 0;
#pragma unroll
        for (int c = 0; c < Q2R_CB; c++) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                unsigned xw = (unsigned)(w == 0 ? x[c].x : w == 1 ? x[c].y : w == 2 ? x[c].z : x[c].w);
                xw = valid ? xw : 0u;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int m = (int)((xw >> (2 * k)) & 0x03030303u);
#pragma unroll
                    for (int p = 0; p < HB_ND; p++) {
                        if (Q2_DIAG == 2) { if (p == 0 && k == 0) a[c * 8 + p] += m ^ dig[p][4 * w + k]; }
                        else a[c * 8 + p] = __builtin_amdgcn_sdot4(m, dig[p][4 * w + k], a[c * 8 + p], false);
                    }
                }
            }
        }
        // halving butterfly: after the step with lane bit s the lanes with that bit clear hold the lower half of the sums,
        // the others the upper half, each added up over the pair
        int n = 16;
#pragma unroll
        for (int s = 32; s >= 2; s >>= 1) {
            const bool up = (lane & s) != 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (i < n) {
                    const int lo = a[i], hi = a[i + n];
                    const int send = up ? lo : hi, keep = up ? hi : lo;
                    a[i] = keep + __shfl_xor(send, s, 64);
                }
            }
            n >>= 1;
        }
        const int tot = a[0] + __shfl_xor(a[0], 1, 64);
        // which sum this lane holds: bit 32 of the lane picked the upper 16, bit 16 the upper 8, ... bit 2 the upper 1
        const int vi = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        const int cc = vi >> 3, pp = vi & 7;
        if (Q2_DIAG == 1) { if (tot == 0x12345678) v.accq[0] = tot; } else
        if (!(lane & 1) && pp < HB_ND)
            __hip_atomic_fetch_add(v.accq + (int64_t)pp * v.accstride + cgi * NC + c0 + cc, (long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_dotq2r(dq_view v, upd_view uq)
{
    __shared__ __attribute__((aligned(16))) char smem[2048 + 4096 + 16]; // the update rows' move lists
    unsigned long long t0 = 0;
    if (v.stamp || v.ldiag) t0 = wall_clock64();
    int b = blockIdx.x;
    if (b < v.nfin) {
        const int col = b * 64 + threadIdx.x;
        if (col < v.fin_ncols) hbq_finalize(v.fin_acc, v.accstride, col, *v.fin_exp, v.fin_out);
    } else if (b < v.nfin + v.nupd) {
        b -= v.nfin;
        update_rows(v.ld, uq, b, reinterpret_cast<int *>(smem), reinterpret_cast<double *>(smem + 2048), reinterpret_cast<int *>(smem + 2048 + 4096), v.ldiag ? v.ldiag + 3 : nullptr);
    } else {
        dotq2r_tile(v, b - v.nupd - v.nfin);
    }
    if (v.stamp && threadIdx.x == 0) {
        v.stamp[2 * (size_t)blockIdx.x] = t0;
        v.stamp[2 * (size_t)blockIdx.x + 1] = wall_clock64();
    }
    if (v.ldiag) hb_ldiag_note(v.ldiag, t0);
}

// int8 column-major -> the packed layout above; thread = one 32-bit word (16 individuals) of one column. Codes must be 0..3
// (the caller has checked min / max over X); individuals past n are zero in X already.
__global__ __launch_bounds__(256) void k_pack2(const int8_t *__restrict__ X, int64_t ld, uint32_t *__restrict__ X2, int64_t ld2w, int ncols)
{
    // (columns and word blocks share grid.x — nwb word blocks per column: grid.y stops at 65 535 and m does not)
    const int nwb = (int)((ld2w + 255) / 256);
    const int j = (int)(blockIdx.x / nwb);
    const int64_t wi = (int64_t)(blockIdx.x % nwb) * blockDim.x + threadIdx.x;
    if (wi >= ld2w || j >= ncols) return;
    unsigned out = 0;
    if (wi * 16 < ld) {
        const hb_u4 q = *reinterpret_cast<const hb_u4 *>(X + (int64_t)j * ld + wi * 16);
        const unsigned w[4] = {q.x, q.y, q.z, q.w}; // w[k]: individuals 16 wi + 4 k + 0..3, one per byte
#pragma unroll
        for (int k = 0; k < 4; k++) out |= (w[k] & 0x03030303u) << (2 * k);
    }
    X2[(int64_t)j * ld2w + wi] = out;
}

// ... and back (hb_ctx_download_genotype / a Gram rebuild once the int8 copy has been dropped)
__global__ __launch_bounds__(256) void k_unpack2(const uint32_t *__restrict__ X2, int64_t ld2w, int8_t *__restrict__ X, int64_t ld, int ncols)
{
    const int nwb = (int)((ld / 16 + 255) / 256);
    const int j = (int)(blockIdx.x / nwb);
    const int64_t wi = (int64_t)(blockIdx.x % nwb) * blockDim.x + threadIdx.x;
    if (wi * 16 >= ld || j >= ncols) return;
    const unsigned w = X2[(int64_t)j * ld2w + wi];
    hb_u4 q;
    q.x = w & 0x03030303u;
    q.y = (w >> 2) & 0x03030303u;
    q.z = (w >> 4) & 0x03030303u;
    q.w = (w >> 6) & 0x03030303u;
    *reinterpret_cast<hb_u4 *>(X + (int64_t)j * ld + wi * 16) = q;
}
