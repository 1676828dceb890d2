// hb_gram.hip — per-panel Gram matrices G_p = X_p' X_p (int32, exact), set-up phase only.
//
// This is the one dense int8 contraction of the engine: the marker sweep uses
// G_p[k][j] = x_k . x_j to keep the running right-hand sides exact inside a panel, the same algebra
// the reference applies in Gram space in its summary-level sampler (src/SBayesD.cpp:262-266).
// X is column-major int8, so both MFMA operands are K-contiguous: every lane feeds
// v_mfma_i32_32x32x32_i8 straight from one 16-byte global load, no LDS transpose.
// One wave = one 64x64 block of the panel's P x P matrix (2x2 MFMA tiles of 32x32).
#include "hb_internal.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Band layout: gram[(p * (L+1) + l) * P*P + row * P + col] = x_{(p-l)P + row} . x_{pP + col}, l = 0..L
// (l = 0: the panel's own Gram; l >= 1: the panels the look-ahead pipeline has not yet folded into
// the residual when panel p's mat-vec runs).
__global__ __launch_bounds__(64) void k_gram(const int8_t *__restrict__ X, int64_t ld, int P, int L,
                                             int32_t *__restrict__ gram)
{
    const int nb = P >> 6;                   // 64-blocks per panel side
    const int pl = blockIdx.x / (nb * nb);
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

int hb_build_gram_impl(hb_ctx *c)
{
    const int nb = c->P / 64;
    hipLaunchKernelGGL(k_gram, dim3((unsigned)(c->npanels * (c->L + 1) * nb * nb)), dim3(64), 0, c->stream, c->X, c->ld, c->P,
                       c->L, c->gram);
    HB_HIP(hipGetLastError());
    return HB_OK;
}
