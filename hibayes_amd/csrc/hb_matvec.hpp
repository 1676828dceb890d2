// hb_matvec.hpp — the panel mat-vec on int8 columns: k_dot (fp32 / fp64) and k_dotq (exact fixed point, v_dot4_i32_i8); the 2-bit kernels are hb_dotq2.hpp.
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

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

