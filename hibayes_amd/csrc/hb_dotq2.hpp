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

#define Q2_RS 512   /* individuals per stage */
#define Q2_DP 4     /* digit pieces per stage */

template <int CPL>
__device__ __forceinline__ void dotq2_tile(const dq_view &v, char *smem, int b)
{
    constexpr int NXP = 8 * CPL, XB = NXP * HBQ_SLOT, BUF = XB + Q2_DP * 1024, PER = NXP + Q2_DP;
    const int lane = threadIdx.x;
    const int cg = b % v.ncg, sp = b / v.ncg;
    if (b == 0 && lane == 0) *v.gexp_out = *v.vexp_in;
    const int st0 = sp * v.NS, st1 = min(v.nstages, st0 + v.NS);
    if (st0 >= st1) return;
    const int64_t ld2 = v.ld2, ld = v.ld;
    const uint8_t *xg = v.X2 + (int64_t)cg * (64 * CPL) * ld2;
    const unsigned voff = (unsigned)((lane >> 3) * ld2 + (lane & 7) * 16);   // piece i: columns 8 i .. 8 i + 7, 128 bytes each
    const unsigned doff = (unsigned)((lane >> 5) * ld + (lane & 31) * 16);   // digit piece j < 3: planes 2 j, 2 j + 1
    const unsigned doff3 = (unsigned)((lane & 31) * 16);                      // digit piece 3: plane 6 (twice)
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    int acc[CPL][HB_ND];
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int k = 0; k < HB_ND; k++) acc[c][k] = 0;
    // (values that ARE wave-uniform, but that hipcc may keep on the vector unit when scalar registers run short)
    auto uni_p = [](const int8_t *p) {
        const unsigned long long u = (unsigned long long)(uintptr_t)p;
        return reinterpret_cast<const int8_t *>((uintptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                                            (unsigned)__builtin_amdgcn_readfirstlane((int)u)));
    };
    auto issue = [&](int st, int buf) {
        const int8_t *xs = uni_p(reinterpret_cast<const int8_t *>(xg) + (int64_t)st * (Q2_RS / 4));
        const int8_t *ds = uni_p(v.rq + (int64_t)st * Q2_RS);
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)buf * BUF));
#pragma unroll
        for (int i = 0; i < NXP; i++) hbq_dma16<true>(voff, xs + (int64_t)(8 * i) * ld2, dst + i * HBQ_SLOT);
#pragma unroll
        for (int j = 0; j < 3; j++) hbq_dma16<false>(doff, ds + (int64_t)(2 * j) * ld, dst + XB + j * 1024);
        hbq_dma16<false>(doff3, ds + (int64_t)6 * ld, dst + XB + 3 * 1024);
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
        for (int c = 0; c < CPL; c++) px[c] = reinterpret_cast<const hb_v4i *>(bp + ((lane >> 3) + 8 * c) * HBQ_SLOT + (lane & 7) * (Q2_RS / 4));
        const char *pd = bp + XB;
#pragma unroll 2
        for (int s = 0; s < Q2_RS / 64; s++) { // 64 individuals per 16-byte read of a column
            hb_v4i x[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) x[c] = px[c][s];
#pragma unroll
            for (int w = 0; w < 4; w++) { // 16 individuals per word
                hb_v4i d[HB_ND];
#pragma unroll
                for (int k = 0; k < HB_ND; k++) d[k] = *reinterpret_cast<const hb_v4i *>(pd + k * Q2_RS + (s * 4 + w) * 16);
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    const unsigned xw = (unsigned)(w == 0 ? x[c].x : w == 1 ? x[c].y : w == 2 ? x[c].z : x[c].w);
                    const int m0 = (int)(xw & 0x03030303u), m1 = (int)((xw >> 2) & 0x03030303u), m2 = (int)((xw >> 4) & 0x03030303u),
                              m3 = (int)((xw >> 6) & 0x03030303u);
#pragma unroll
                    for (int k = 0; k < HB_ND; k++) {
                        acc[c][k] = __builtin_amdgcn_sdot4(m0, d[k].x, acc[c][k], false);
                        acc[c][k] = __builtin_amdgcn_sdot4(m1, d[k].y, acc[c][k], false);
                        acc[c][k] = __builtin_amdgcn_sdot4(m2, d[k].z, acc[c][k], false);
                        acc[c][k] = __builtin_amdgcn_sdot4(m3, d[k].w, acc[c][k], false);
                    }
                }
            }
        }
        buf ^= 1;
    }
#pragma unroll
    for (int c = 0; c < CPL; c++)
#pragma unroll
        for (int k = 0; k < HB_ND; k++)
            __hip_atomic_fetch_add(v.accq + (int64_t)k * v.accstride + cg * (64 * CPL) + c * 64 + lane, (long long)acc[c][k], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
}

template <int CPL>
__global__ __launch_bounds__(64) void k_dotq2(dq_view v, upd_view uq)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t0 = 0;
    if (v.stamp) t0 = wall_clock64();
    int b = blockIdx.x;
    if (b < v.nupd) { // residual update of an earlier group (2-bit columns: upd_view.X2), lists staged in the tile buffers
        update_rows(v.ld, uq, b, reinterpret_cast<int *>(smem), reinterpret_cast<double *>(smem + 2048), reinterpret_cast<int *>(smem + 2048 + 4096));
    } else if (b < v.nupd + v.nfin) {
        const int col = (b - v.nupd) * 64 + threadIdx.x;
        if (col < v.fin_ncols) hbq_finalize(v.fin_acc, v.accstride, col, *v.fin_exp, v.fin_out);
    } else {
        dotq2_tile<CPL>(v, smem, b - v.nupd - v.nfin);
    }
    if (v.stamp && threadIdx.x == 0) {
        v.stamp[2 * (size_t)blockIdx.x] = t0;
        v.stamp[2 * (size_t)blockIdx.x + 1] = wall_clock64();
    }
}

// int8 column-major -> the packed layout above; thread = one 32-bit word (16 individuals) of one column. Codes must be 0..3
// (the caller has checked min / max over X); individuals past n are zero in X already.
__global__ __launch_bounds__(256) void k_pack2(const int8_t *__restrict__ X, int64_t ld, uint32_t *__restrict__ X2, int64_t ld2w, int ncols)
{
    const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
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
    const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (wi * 16 >= ld || j >= ncols) return;
    const unsigned w = X2[(int64_t)j * ld2w + wi];
    hb_u4 q;
    q.x = w & 0x03030303u;
    q.y = (w >> 2) & 0x03030303u;
    q.z = (w >> 4) & 0x03030303u;
    q.w = (w >> 6) & 0x03030303u;
    *reinterpret_cast<hb_u4 *>(X + (int64_t)j * ld + wi * 16) = q;
}
