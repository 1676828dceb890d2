// hb_gram.hip — per-panel Gram matrices G_p = X_p' X_p (int32, exact), set-up phase only.
//
// This is the one dense int8 contraction of the engine: the marker sweep uses
// G_p[k][j] = x_k . x_j to keep the running right-hand sides exact inside a panel, the same algebra
// the reference applies in Gram space in its summary-level sampler (src/SBayesD.cpp:262-266).
// X is column-major int8, so both MFMA operands are K-contiguous: every lane feeds
// v_mfma_i32_32x32x32_i8 straight from one 16-byte global load, no LDS transpose.
// One wave = one 64x64 block of the panel's P x P matrix (2x2 MFMA tiles of 32x32).
#include "hb_internal.hpp"
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Band layout: gram[(p * (L+1) + l) * P*P + row * P + col] = x_{(p-l)P + row} . x_{pP + col}, l = 0..L
// (l = 0: the panel's own Gram; l >= 1: the panels the look-ahead pipeline has not yet folded into
// the residual when panel p's mat-vec runs).
__global__ __launch_bounds__(64) void k_gram(const int8_t *__restrict__ X, int64_t ld, int P, int L,
                                             int32_t *__restrict__ gram, int p_lo)
{
    const int nb = P >> 6;                   // 64-blocks per panel side
    const int pl = blockIdx.x / (nb * nb) + p_lo * (L + 1); // (panels [p_lo, ...) of this launch: X may be a window, see hb_build_gram_impl)
    const int p = pl / (L + 1), l = pl % (L + 1);
    if (p - l < 0) return;
    const int rem = blockIdx.x % (nb * nb);
    const int bi = rem / nb, bj = rem % nb;
    const int lane = threadIdx.x;
    const int8_t *xa = X + ((int64_t)(p - l) * P + bi * 64 + (lane & 31)) * ld + 16 * (lane >> 5);
    const int8_t *xb = X + ((int64_t)p * P + bj * 64 + (lane & 31)) * ld + 16 * (lane >> 5);
    v16i acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0;
    for (int64_t kk = 0; kk < ld; kk += 32) {
        const v4i a0 = *reinterpret_cast<const v4i *>(xa + kk);
        const v4i a1 = *reinterpret_cast<const v4i *>(xa + 32 * ld + kk);
        const v4i b0 = *reinterpret_cast<const v4i *>(xb + kk);
        const v4i b1 = *reinterpret_cast<const v4i *>(xb + 32 * ld + kk);
        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc[1][1], 0, 0, 0);
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    int32_t *gp = gram + (size_t)pl * P * P;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = bi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = bj * 64 + b * 32 + (lane & 31);
                gp[(size_t)row * P + col] = acc[a][b][r];
            }
}

// The same blocks for P >= 256, one 256 x 256 tile per workgroup of 16 waves (4 x 4 wave tiles of 64 x 64): per 64-row step the
// 256 + 256 operand columns are staged through LDS once (16-byte global loads into registers one step ahead, ds_write_b128 with
// an 80-byte column stride so that the MFMA operand reads are bank-conflict free) instead of being re-read from global memory by
// every wave tile — 4x less traffic than k_gram (9.5 TB -> 1.8 TB of fetches for the 18 GB band at n=50k, m=500k).
#define HG_T 256
#define HG_KS 64
#define HG_CS 80 /* bytes per staged column: 64 + 16 */
__global__ __launch_bounds__(1024) void k_gram_tiled(const int8_t *__restrict__ X, int64_t ld, int P, int L, int32_t *__restrict__ gram, int p_lo)
{
    __shared__ __attribute__((aligned(16))) char sa[HG_T * HG_CS], sb[HG_T * HG_CS];
    const int nt = P / HG_T;
    const int pl = blockIdx.x / (nt * nt) + p_lo * (L + 1);
    const int p = pl / (L + 1), l = pl % (L + 1);
    if (p - l < 0) return;
    const int rem = blockIdx.x % (nt * nt);
    const int ti = rem / nt, tj = rem % nt;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 2, wc = wave & 3;
    // staging: thread -> (column tid / 4, 16-byte part tid % 4) of both operand tiles
    const int scol = tid >> 2, spart = tid & 3;
    const int8_t *ga = X + ((int64_t)(p - l) * P + ti * HG_T + scol) * ld + spart * 16;
    const int8_t *gb = X + ((int64_t)p * P + tj * HG_T + scol) * ld + spart * 16;
    char *wa = sa + scol * HG_CS + spart * 16, *wb = sb + scol * HG_CS + spart * 16;
    // MFMA operands: lane & 31 = column inside the 32-wide sub-tile, lane >> 5 = which 16 of the 32 k values
    const char *ra = sa + (wr * 64 + (lane & 31)) * HG_CS + (lane >> 5) * 16;
    const char *rb = sb + (wc * 64 + (lane & 31)) * HG_CS + (lane >> 5) * 16;
    v16i acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0;
    v4i na = *reinterpret_cast<const v4i *>(ga), nb = *reinterpret_cast<const v4i *>(gb);
    for (int64_t kk = 0; kk < ld; kk += HG_KS) {
        __syncthreads(); // everybody is done reading the previous step
        *reinterpret_cast<v4i *>(wa) = na;
        *reinterpret_cast<v4i *>(wb) = nb;
        __syncthreads();
        if (kk + HG_KS < ld) { // next step's operands travel while this one is multiplied
            na = *reinterpret_cast<const v4i *>(ga + kk + HG_KS);
            nb = *reinterpret_cast<const v4i *>(gb + kk + HG_KS);
        }
#pragma unroll
        for (int ks = 0; ks < HG_KS; ks += 32) {
            const v4i a0 = *reinterpret_cast<const v4i *>(ra + ks);
            const v4i a1 = *reinterpret_cast<const v4i *>(ra + 32 * HG_CS + ks);
            const v4i b0 = *reinterpret_cast<const v4i *>(rb + ks);
            const v4i b1 = *reinterpret_cast<const v4i *>(rb + 32 * HG_CS + ks);
            acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    int32_t *gp = gram + (size_t)pl * P * P;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = ti * HG_T + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = tj * HG_T + wc * 64 + b * 32 + (lane & 31);
                gp[(size_t)row * P + col] = acc[a][b][r];
            }
}

int hbk_unpack2(hb_ctx *c, int col0, int ncols, int8_t *dst);

// panels [pa, pb) of the band; Xv = where column 0 WOULD be (a window of the matrix may be all that exists: only the columns of
// panels pa - Lg .. pb - 1 are read)
static int gram_launch(hb_ctx *c, const int8_t *Xv, int pa, int pb)
{
    if (c->P >= HG_T && !getenv("HB_GRAM_UNTILED")) {
        const int nt = c->P / HG_T;
        hipLaunchKernelGGL(k_gram_tiled, dim3((unsigned)((pb - pa) * (c->Lg + 1) * nt * nt)), dim3(1024), 0, c->stream, Xv, c->ld,
                           c->P, c->Lg, c->gram, pa);
    } else {
        const int nb = c->P / 64;
        hipLaunchKernelGGL(k_gram, dim3((unsigned)((pb - pa) * (c->Lg + 1) * nb * nb)), dim3(64), 0, c->stream, Xv, c->ld, c->P,
                           c->Lg, c->gram, pa);
    }
    HB_HIP(hipGetLastError());
    return HB_OK;
}

// ---- the compact band (round 5): G = ga (x) gB + int16 residual ----
__global__ void k_g16_ab(const double *__restrict__ s1, int m_pad, double n, int32_t *__restrict__ ga, int32_t *__restrict__ gB)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m_pad) return;
    const double s = s1[j];
    ga[j] = (int32_t)rint(s / 256.0);
    gB[j] = (int32_t)rint(s * 256.0 / n);
}
// one thread per four consecutive entries of a row (the band's own order: [panel][block l][row k][column t])
__global__ __launch_bounds__(256) void k_gram16(const int32_t *__restrict__ gram, int16_t *__restrict__ g16, const int32_t *__restrict__ ga,
                                                const int32_t *__restrict__ gB, int P, int Lg, size_t nquads, int *__restrict__ flag)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nquads) return;
    const size_t idx = q * 4, PP = (size_t)P * P;
    const size_t blk = idx / PP, within = idx % PP;
    const int p = (int)(blk / (size_t)(Lg + 1)), l = (int)(blk % (size_t)(Lg + 1));
    const int k = (int)(within / P), t = (int)(within % P);
    const int4 gq = *reinterpret_cast<const int4 *>(gram + idx);
    int a = 0;
    if (p - l >= 0) a = ga[(size_t)(p - l) * P + k];
    const int4 bq = *reinterpret_cast<const int4 *>(gB + (size_t)p * P + t);
    // (64-bit: the difference of two int32-range numbers; whatever does not fit a short raises the flag and the band stays int32 only)
    const long long d0 = (long long)gq.x - (long long)a * bq.x, d1 = (long long)gq.y - (long long)a * bq.y, d2 = (long long)gq.z - (long long)a * bq.z,
                    d3 = (long long)gq.w - (long long)a * bq.w;
    const long long lo = min(min(d0, d1), min(d2, d3)), hi = max(max(d0, d1), max(d2, d3));
    if (p - l >= 0 && (lo < -32767 || hi > 32767)) *flag = 1; // (any writer: the band then stays int32 only)
    const int c0 = (int)d0, c1 = (int)d1, c2 = (int)d2, c3 = (int)d3;
    short4 o;
    o.x = (short)c0; o.y = (short)c1; o.z = (short)c2; o.w = (short)c3;
    *reinterpret_cast<short4 *>(g16 + idx) = o;
}

// cmax[k] = max over the stored band of |G[k][j] - ga[k] gB[j]|, row marker k (one 128-thread block per row of a block: P / 4 threads x 4 entries)
__global__ __launch_bounds__(128) void k_gcmax(const int32_t *__restrict__ gram, const int32_t *__restrict__ ga, const int32_t *__restrict__ gB, int P, int Lg,
                                               int32_t *__restrict__ gcmax)
{
    __shared__ int red[2];
    const size_t rowi = blockIdx.x; // [panel][block l][row k]
    const size_t blk = rowi / P;
    const int k = (int)(rowi % P);
    const int p = (int)(blk / (size_t)(Lg + 1)), l = (int)(blk % (size_t)(Lg + 1));
    if (p - l < 0) return; // (uniform per block: no such pair of panels)
    const size_t rm = (size_t)(p - l) * P + k;
    const int a = ga[rm];
    int mx = 0;
    for (int t4 = threadIdx.x * 4; t4 < P; t4 += blockDim.x * 4) {
        const int4 gq = *reinterpret_cast<const int4 *>(gram + rowi * P + t4);
        const int4 bq = *reinterpret_cast<const int4 *>(gB + (size_t)p * P + t4);
        // (the diagonal, x_k . x_k, is no pair: a marker's own entry never reaches another marker's right-hand side)
        // (advisor finding, round 5: in int32 the difference can overflow for int8-coded genotypes near the Gram's own range check — 64-bit, clamped:
        // a bound of INT32_MAX makes the certificate prove nothing for that marker, which is safe)
        auto cabs = [](int g, int aa, int b) { const long long d = (long long)g - (long long)aa * (long long)b; const long long m = d < 0 ? -d : d; return (int)(m > 2147483647ll ? 2147483647ll : m); };
        const int c0 = (l == 0 && t4 == k) ? 0 : cabs(gq.x, a, bq.x), c1 = (l == 0 && t4 + 1 == k) ? 0 : cabs(gq.y, a, bq.y);
        const int c2 = (l == 0 && t4 + 2 == k) ? 0 : cabs(gq.z, a, bq.z), c3 = (l == 0 && t4 + 3 == k) ? 0 : cabs(gq.w, a, bq.w);
        mx = max(mx, max(max(c0, c1), max(c2, c3)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(gcmax + rm, max(red[0], blockDim.x > 64 ? red[1] : 0));
}

// after the int32 band is in place: the arrays of the group chain's certificate (hb_chain_group.hpp, CERT): G[k][j] = ga[k] gB[j] + c[k][j] with
// |c[k][j]| <= gcmax[k] for every stored pair — exact integers, whatever the genotype coding
int hb_build_gcert(hb_ctx *c)
{
    c->gcert_ok = false;
    if (!c->gcert_on || c->P % 4 != 0 || c->row_reduce) return HB_OK; // (row-sharded mode: the local blocks are partial sums)
    if (!c->ga) {
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->ga), sizeof(int32_t) * (size_t)c->m_pad));
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->gB), sizeof(int32_t) * (size_t)c->m_pad));
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->gcmax), sizeof(int32_t) * (size_t)c->m_pad));
        HB_HIP(hipMalloc(reinterpret_cast<void **>(&c->g16_flag), sizeof(int)));
    }
    hipLaunchKernelGGL(k_g16_ab, dim3((c->m_pad + 255) / 256), dim3(256), 0, c->stream, c->s1, c->m_pad, (double)c->n, c->ga, c->gB);
    HB_HIP(hipMemsetAsync(c->gcmax, 0, sizeof(int32_t) * (size_t)c->m_pad, c->stream));
    const size_t nrows = (size_t)c->npanels * (size_t)(c->Lg + 1) * (size_t)c->P;
    hipLaunchKernelGGL(k_gcmax, dim3((unsigned)nrows), dim3(128), 0, c->stream, c->gram, c->ga, c->gB, c->P, c->Lg, c->gcmax);
    HB_HIP(hipGetLastError());
    c->gcert_ok = true;
    return HB_OK;
}

// ... and the compact copy, if every residual fits an int16 (non-negative genotype codes only: the column
// sums are then the scale of every product)
int hb_build_gram16(hb_ctx *c)
{
    c->gram16_ok = false;
    if (!c->gram16_on || c->xmin < 0 || c->P % 4 != 0 || c->row_reduce) return HB_OK; // (row-sharded mode: the local blocks are partial sums)
    const size_t need = (size_t)c->m_pad * (size_t)c->P * (size_t)(c->Lg + 1);
    if (need > c->gram16_cap) {
        if (c->gram16) { (void)hipFree(c->gram16); c->gram16 = nullptr; }
        c->gram16_cap = 0;
        if (hipMalloc(reinterpret_cast<void **>(&c->gram16), need * sizeof(int16_t)) != hipSuccess) { // (no room: the int32 band serves)
            (void)hipGetLastError();
            c->gram16 = nullptr;
            return HB_OK;
        }
        c->gram16_cap = need;
    }
    if (!c->gcert_ok) return HB_OK; // (ga / gB come from hb_build_gcert)
    HB_HIP(hipMemsetAsync(c->g16_flag, 0, sizeof(int), c->stream));
    const size_t nquads = need / 4;
    hipLaunchKernelGGL(k_gram16, dim3((unsigned)((nquads + 255) / 256)), dim3(256), 0, c->stream, c->gram, c->gram16, c->ga, c->gB, c->P, c->Lg,
                       nquads, c->g16_flag);
    HB_HIP(hipGetLastError());
    int flag = 0;
    HB_HIP(hipMemcpyAsync(&flag, c->g16_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    c->gram16_ok = flag == 0;
    return HB_OK;
}

int hb_build_gram_impl(hb_ctx *c)
{
    if (c->X) return gram_launch(c, c->X, 0, c->npanels);
    // Only the 2-bit resident layout exists (the int8 copy was dropped, hb_ctx_set_layout(c, 2, 0)): the columns are unpacked a
    // window at a time into a scratch buffer — the panels of one chunk plus the Lg panels before them that their band blocks reach
    // back to —, never the whole matrix (25 GB at n = 50k, m = 500k: the capacity regime the 2-bit layout exists for).
    if (!c->X2) return hb_fail(HB_ERR_INVALID, "hb_ctx_build_gram: no genotypes on the device");
    const size_t colbytes = (size_t)c->ld * c->P;
    int chunk = (int)std::max<size_t>(1, ((size_t)1 << 31) / colbytes); // ~2 GB of panels per window
    chunk = std::min(chunk, c->npanels);
    int8_t *win = nullptr;
    HB_HIP(hipMalloc(reinterpret_cast<void **>(&win), colbytes * (size_t)(chunk + c->Lg)));
    int rc = HB_OK;
    for (int pa = 0; pa < c->npanels && rc == HB_OK; pa += chunk) {
        const int pb = std::min(c->npanels, pa + chunk), first = std::max(0, pa - c->Lg);
        rc = hbk_unpack2(c, first * c->P, (pb - first) * c->P, win);
        if (rc == HB_OK) rc = gram_launch(c, win - (int64_t)first * (int64_t)colbytes, pa, pb);
        if (rc == HB_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = hb_fail(HB_ERR_HIP, "hb_ctx_build_gram: window build failed");
    }
    (void)hipFree(win);
    return rc;
}
