// hb_blocks.hpp — the host blocks that share yadj (src/Bayes.cpp:479-516) as device kernels: intercept shift, covariates, random-effect levels; the sharded run's delta pack / unpack.
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once

// ---------------------------------------------------------------------------------------------
// helpers for the host blocks sharing yadj (reference src/Bayes.cpp:479-516)
// ---------------------------------------------------------------------------------------------
__global__ void k_shift(double *__restrict__ r, float *__restrict__ r32, int n, double a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = r[i] + a;
    r[i] = v;
    r32[i] = (float)v;
}

__global__ void k_axpy(double *__restrict__ r, float *__restrict__ r32, const double *__restrict__ x, int n, double a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = fma(a, x[i], r[i]);
    r[i] = v;
    r32[i] = (float)v;
}

__global__ __launch_bounds__(1024) void k_dot_vec(const double *__restrict__ x, const double *__restrict__ y, int n,
                                                  double *__restrict__ out)
{
    __shared__ double red[16];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s = fma(x[i], y[i], s);
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out = s;
}

// Z_t' yadj: per-level sums; one workgroup, LDS-free atomics on a zeroed buffer
__global__ void k_level_sums(const double *__restrict__ r, const int32_t *__restrict__ zid, int n,
                             double *__restrict__ sums)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAdd(&sums[zid[i]], r[i]);
}

__global__ void k_level_axpy(double *__restrict__ r, float *__restrict__ r32, const int32_t *__restrict__ zid, int n,
                             const double *__restrict__ delta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = r[i] + delta[zid[i]];
    r[i] = v;
    r32[i] = (float)v;
}

// ---- the covariate and random-effect blocks of one iteration entirely on the device (reference src/Bayes.cpp:484-516) ----
// The host pre-draws the deviates in the reference's order (they do not depend on the data) and passes them in; nothing
// comes back until the iteration's single fetch. One workgroup each: n is a few hundred KB.
__global__ __launch_bounds__(1024) void k_cov_step(double *__restrict__ r, float *__restrict__ r32, const double *__restrict__ ci, int n,
                                                   double v, double vare, double z, double *__restrict__ beta_i)
{
    __shared__ double red[16];
    double s = 0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) s = fma(ci[k], r[k], s);
    s = block_sum(s, red);                                   // rhs = C_i . yadj            (:487)
    const double old = *beta_i;
    const double rhs = s + v * old;                          // (:488)
    const double gi = rhs / v + sqrt(vare / v) * z;          // norm_sample(rhs / v, sqrt(vare / v))  (:489)
    const double d = old - gi;                               // (:490)
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) {      // daxpy (:491)
        const double a = fma(d, ci[k], r[k]);
        r[k] = a;
        r32[k] = (float)a;
    }
    if (threadIdx.x == 0) *beta_i = gi;
}

__global__ __launch_bounds__(1024) void k_lev_step(double *__restrict__ r, float *__restrict__ r32, const int32_t *__restrict__ zid, int n,
                                                   int qr, const double *__restrict__ zz, double *__restrict__ estR,
                                                   const double *__restrict__ z, double *__restrict__ work, double vare,
                                                   double *__restrict__ vrtmp, double *__restrict__ vr, double s2r_dfr, double chis)
{
    __shared__ double red[16];
    for (int q = threadIdx.x; q < qr; q += blockDim.x) st_sc1(work + q, 0.0);
    __threadfence();
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) atomicAdd(&work[zid[k]], r[k]);     // Z' yadj            (:501)
    __threadfence();
    __syncthreads();
    const double lam = vare / *vrtmp;
    double ss = 0, sm = 0;
    for (int q = threadIdx.x; q < qr; q += blockDim.x) {
        const double rhs = ld_sc1(work + q) + zz[q] * estR[q];                            // + ZZ estR          (:502)
        const double l = zz[q] + lam;                                                     // (:504)
        const double en = rhs / l + sqrt(vare / l) * z[q];                                // (:505)
        st_sc1(work + q, estR[q] - en);                                                   // what yadj moves by (:508-510)
        estR[q] = en;
        ss = fma(en, en, ss);
        sm += en;
    }
    ss = block_sum(ss, red);
    sm = block_sum(sm, red);
    const double mean = sm / qr;
    double a2 = 0, a3 = 0;
    for (int q = threadIdx.x; q < qr; q += blockDim.x) { // arma::var, two-pass, N - 1 (:513)
        const double d = mean - estR[q];
        a2 = fma(d, d, a2);
        a3 += d;
    }
    a2 = block_sum(a2, red);
    a3 = block_sum(a3, red);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        *vrtmp = (ss + s2r_dfr) / chis;                                                   // (:512)
        *vr = qr > 1 ? (a2 - a3 * a3 / qr) / (qr - 1) : 0.0;
    }
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double a = r[k] + ld_sc1(work + zid[k]);
        r[k] = a;
        r32[k] = (float)a;
    }
}

int hbk_cov_step(hb_ctx *c, int i, double v, double vare, double z, double *beta_i)
{
    hipLaunchKernelGGL(k_cov_step, dim3(1), dim3(1024), 0, c->stream, c->r, c->r32, c->Cmat + (size_t)i * c->n, c->n, v, vare, z, beta_i);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

int hbk_lev_step(hb_ctx *c, int term, int q0, int qr, const double *zz, double *estR, const double *z, double vare, double *vrtmp,
                 double *vr, double s2r_dfr, double chis)
{
    hipLaunchKernelGGL(k_lev_step, dim3(1), dim3(1024), 0, c->stream, c->r, c->r32, c->zid + (size_t)term * c->n, c->n, qr, zz + q0,
                       estR + q0, z + q0, c->lev_buf, vare, vrtmp, vr, s2r_dfr, chis);
    HB_HIP(hipGetLastError());
    return HB_OK;
}

__global__ void k_to_f32(const double *__restrict__ r, float *__restrict__ r32, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r32[i] = (float)r[i];
}

// multi-GPU exchange: pack (yadj - yadj_start, u - u_start) and unpack the summed deltas
__global__ void k_delta_pack(const double *__restrict__ r, const double *__restrict__ u,
                             const double *__restrict__ r0, const double *__restrict__ u0, int n,
                             double *__restrict__ buf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    buf[i] = r[i] - r0[i]; // (u moved by exactly the negative: u = X g, yadj = y - ... - X g)
}

__global__ void k_delta_unpack(double *__restrict__ r, double *__restrict__ u, float *__restrict__ r32,
                               const double *__restrict__ r0, const double *__restrict__ u0, int n,
                               const double *__restrict__ buf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = r0[i] + buf[i];
    r[i] = a;
    r32[i] = (float)a;
    u[i] = u0[i] - buf[i];
}

