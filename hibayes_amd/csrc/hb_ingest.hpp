// hb_ingest.hpp — data paths either side of X: f64 -> int8 check, .bed decode on the device, X * alpha, the GEBV sample matrix, the synthetic generator.
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

// ---------------------------------------------------------------------------------------------
// data paths upstream of X: f64 -> int8 check, .bed decode, synthetic generator
// ---------------------------------------------------------------------------------------------
__global__ void k_f64_to_i8(const double *__restrict__ src, int64_t lds, int n, int ncols,
                            int8_t *__restrict__ dst, int64_t ldd, int *__restrict__ bad)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * ncols) return;
    const int c = (int)(idx / n), i = (int)(idx % n);
    const double v = src[(int64_t)c * lds + i];
    const double rv = rint(v);
    if (!(rv == v) || rv < -127.0 || rv > 127.0) { atomicExch(bad, 1); return; }
    dst[(int64_t)c * ldd + i] = (int8_t)rv;
}

// PLINK .bed SNP-major: byte (i>>2) of SNP j, bits 2*(i&3); map 00->2, 01->NA, 10->1, 11->0
// (reference src/read_bed.cpp:116-120).  One workgroup per SNP: count genotypes over ALL nind
// individuals (the reference imputes before ibrm() subsets rows, :182-230), then write the
// selected rows.
__global__ __launch_bounds__(256) void k_bed_decode(const uint8_t *__restrict__ bed, int64_t bpc, int nind,
                                                    const int32_t *__restrict__ rows, int n, int8_t *__restrict__ dst,
                                                    int64_t ldd)
{
    __shared__ long long red[4];
    const int j = blockIdx.x;
    const uint8_t *p = bed + (int64_t)j * bpc;
    long long c0 = 0, c1 = 0, c2 = 0, cm = 0;
    for (int i = threadIdx.x; i < nind; i += blockDim.x) {
        const int code = (p[i >> 2] >> (2 * (i & 3))) & 3;
        c2 += (code == 0);
        cm += (code == 1);
        c1 += (code == 2);
        c0 += (code == 3);
    }
    c0 = block_sum(c0, red);
    c1 = block_sum(c1, red);
    c2 = block_sum(c2, red);
    cm = block_sum(cm, red);
    int8_t major = 0;
    long long best = 0;
    if (c0 > best) { best = c0; major = 0; }
    if (c1 > best) { best = c1; major = 1; }
    if (c2 > best) { best = c2; major = 2; }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int src = rows ? rows[i] : i;
        const int code = (p[src >> 2] >> (2 * (src & 3))) & 3;
        const int8_t gg = code == 0 ? 2 : code == 2 ? 1 : code == 3 ? 0 : major;
        dst[(int64_t)j * ldd + i] = gg;
    }
    (void)cm;
}

// out[row] = sum_j x[row][j] alpha[j]  (e -= X*alpha, reference src/Bayes.cpp:971); block = 1024 rows x 256 columns
__global__ __launch_bounds__(256) void k_xalpha(const int8_t *__restrict__ X, int64_t ld, const uint32_t *__restrict__ X2, int64_t ld2w, int m_pad,
                                                const double *__restrict__ alpha, double *__restrict__ out)
{
    const int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (row0 >= ld) return;
    const int j0 = blockIdx.y * 256, j1 = min(m_pad, j0 + 256);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int j = j0; j < j1; j++) {
        const double al = alpha[j];
        if (al == 0.0) continue;
        const int w = hb_ld4(X, ld, X2, ld2w, j, row0);
        a0 = fma((double)(int8_t)(w), al, a0);
        a1 = fma((double)(int8_t)(w >> 8), al, a1);
        a2 = fma((double)(int8_t)(w >> 16), al, a2);
        a3 = fma((double)(int8_t)(w >> 24), al, a3);
    }
    if (a0 != 0.0) atomicAdd(out + row0, a0);
    if (a1 != 0.0) atomicAdd(out + row0 + 1, a1);
    if (a2 != 0.0) atomicAdd(out + row0 + 2, a2);
    if (a3 != 0.0) atomicAdd(out + row0 + 3, a3);
}

// out[rec][row] = sum_e x[row][idx[e]] * val[e][rec] for 8 sample records at once: MCMCsamples$g = M %*% MCMCsamples$alpha,
// reference R/bayes.r:303-305. The host hands over only the columns where any of the 8 records is non-zero (the
// point-mass models keep ~0.1-5 % of the markers in the model), so the work is n x nnz x 8 instead of n x m x 8.
// thread = 4 rows x 8 records (32 fp64 accumulators, no atomics); the column index and its 8 effects are wave-uniform.
#define HB_XM_RB 8
__global__ __launch_bounds__(256) void k_xmat(const int8_t *__restrict__ X, int64_t ld, const uint32_t *__restrict__ X2, int64_t ld2w, const int *__restrict__ idx,
                                              const double *__restrict__ val, int nnz, double *__restrict__ out, int64_t ldo)
{
    const int64_t row0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (row0 >= ld) return;
    double acc[4][HB_XM_RB];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < HB_XM_RB; r++) acc[a][r] = 0.0;
    for (int e0 = 0; e0 < nnz; e0 += 4) {
        int w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = hb_ld4(X, ld, X2, ld2w, idx[min(e0 + k, nnz - 1)], row0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (e0 + k < nnz) { // uniform
                const double *v = val + (size_t)(e0 + k) * HB_XM_RB;
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const double x = (double)(int8_t)(w[k] >> (8 * a));
#pragma unroll
                    for (int r = 0; r < HB_XM_RB; r++) acc[a][r] = fma(x, v[r], acc[a][r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < HB_XM_RB; r++)
#pragma unroll
        for (int a = 0; a < 4; a++) out[(int64_t)r * ldo + row0 + a] = acc[a][r];
}

int hbk_xmat(hb_ctx *c, const int *didx, const double *dval, int nnz, double *dout)
{
    hipLaunchKernelGGL(k_xmat, dim3((unsigned)((c->ld / 4 + 255) / 256)), dim3(256), 0, c->stream, c->X, c->ld, c->layout == 2 ? c->X2 : nullptr, c->ld2 / 4, didx, dval, nnz, dout, c->ld);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

// synthetic genotypes, SURVEY §8(d): p_j ~ U(0.05, 0.5), x ~ Binomial(2, p_j); thread = 4 rows
__global__ __launch_bounds__(256) void k_generate(int8_t *__restrict__ X, int64_t ld, int n, int m, int64_t m_offset,
                                                  uint64_t seed, int mono_every)
{
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= ld) return;
    for (int j = blockIdx.y; j < m; j += gridDim.y) {
    const uint64_t gj = (uint64_t)(m_offset + j);
    const uint64_t sub = hb_sub(HB_PURPOSE_DATA, gj);
    const double pj = 0.05 + 0.45 * hb_uniform_blk(seed, sub, 0xFFFFFFFFFFull);
    const unsigned thr16 = (unsigned)(pj * 65536.0);
    const bool mono = mono_every > 0 && (gj % (uint64_t)mono_every) == (uint64_t)(mono_every - 1);
    const uint4 w = hb_block(seed, sub, (uint64_t)i4);
    const unsigned ws[4] = {w.x, w.y, w.z, w.w};
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        unsigned x = ((ws[k] & 0xffffu) < thr16) + ((ws[k] >> 16) < thr16);
        if (mono || i4 * 4 + k >= n) x = 0;
        out |= x << (8 * k);
    }
    *reinterpret_cast<unsigned *>(X + (int64_t)j * ld + i4 * 4) = out;
    }
}


// ---------------------------------------------------------------------------------------------
// measurement helper (SURVEY §8 d: "fraction of a measured streaming-read kernel on the same buffer"): what the chip delivers when
// the resident genotypes are only READ — 16 bytes per lane and load, eight loads in flight per lane, non-temporal, a grid that fills
// every compute unit eight times over. The OR of everything read is stored only if it has a value no genotype word combination can
// give under the mask below, so the loads cannot be dropped and nothing is written.
// ---------------------------------------------------------------------------------------------
typedef unsigned hb_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_read(const hb_u32x4 *__restrict__ src, int64_t n16, unsigned *__restrict__ sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    hb_u32x4 a = {0, 0, 0, 0};
    for (; i + 7 * stride < n16; i += 8 * stride) {
        hb_u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; k++) a |= v[k];
    }
    for (; i < n16; i += stride) {
        a |= __builtin_nontemporal_load(src + i);
    }
    const unsigned o = a.x | a.y | a.z | a.w;
    if ((o & 0x80808080u) == 0x80808080u && (o & 0x7f7f7f7fu) == 0x12345678u) *sink = o; // (never true for genotype data; keeps the loads)
}

int hbk_time_stream_read(hb_ctx *c, int reps, double *avg_ms, int64_t *bytes)
{
    HB_HIP(hipSetDevice(c->device));
    const bool two = c->layout == 2 && c->X2;
    const void *p = two ? (const void *)c->X2 : (const void *)c->X;
    const int64_t nb = (two ? c->ld2 : c->ld) * (int64_t)c->m_pad;
    if (!p || nb < 16) return hb_fail(HB_ERR_INVALID, "hb_ctx_time_stream_read: no resident genotypes");
    hipEvent_t e0, e1;
    HB_HIP(hipEventCreate(&e0));
    HB_HIP(hipEventCreate(&e1));
    unsigned *sink = reinterpret_cast<unsigned *>(c->scratch);
    const dim3 grid((unsigned)(c->num_cus * 8)), blk(256);
    hipLaunchKernelGGL(k_stream_read, grid, blk, 0, c->stream, reinterpret_cast<const hb_u32x4 *>(p), nb / 16, sink);
    HB_HIP(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_stream_read, grid, blk, 0, c->stream, reinterpret_cast<const hb_u32x4 *>(p), nb / 16, sink);
    HB_HIP(hipEventRecord(e1, c->stream));
    HB_HIP(hipStreamSynchronize(c->stream));
    HB_HIP(hipGetLastError());
    float ms = 0;
    HB_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
    *bytes = nb / 16 * 16;
    return HB_OK;
}
