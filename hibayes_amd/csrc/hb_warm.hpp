// hb_warm.hpp — k_gate and k_warm: the chain workgroup made resident first, and its Gram rows pulled into its XCD's L2 ahead of it.
// Part of the one translation unit hb_kernels.hip (the kernels share device globals and the views defined before them);
// included there in this order, not compiled on its own.
#pragma once


// k_gate: one lane on the mat-vec stream, ahead of the sweep's first launch, that waits until the chain workgroup is resident
// (it publishes HB_FLAG_XCC as its first act). The chain needs a compute unit with ALL of its LDS free; it is launched first,
// but the graph's branches start together, and once mat-vec blocks have touched every compute unit it only gets one when a
// compute unit drains completely — which never happens where a launch's update blocks, one per 64 rows, sit on every compute
// unit waiting for the chain (measured: the sweep then times out; with fewer update blocks than compute units the late start
// went unnoticed). While this lane waits the chip is empty, so the chain starts at once.
__global__ void k_gate(unsigned *flags)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (ld_flag(flags + HB_FLAG_XCC) == 0u) {
        if (ld_flag(flags + HB_FLAG_ABORT)) return;
        if (wall_clock64() - t0 > HB_TIMEOUT_TICKS) { st_flag(flags + HB_FLAG_ABORT, 1u); return; }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---------------------------------------------------------------------------------------------
// k_warm: the chain workgroup's memory traffic, pulled into ITS L2 ahead of time by other compute units.
// One compute unit gets ~18 bytes per clock out of HBM however many loads it keeps in flight (its miss queue is the limit;
// tools/rowfetch_bench.hip: 113 cycles per 2-KiB row), but 64 bytes per clock out of its XCD's L2 (31 cycles per row). What
// the chain will read is known a sweep ahead for every marker on a panel's hot list (k_hotlist: the markers in the model —
// certain to move — and the likely entries): the Gram row that goes into the row cache and the band rows its move folds
// forward. The workgroups of this kernel that landed on the chain's XCD (workgroups are dealt round-robin over the 8 XCDs;
// the chain publishes its own) read exactly those rows, `ahead` panels in front of the chain's published progress, and
// throw the data away. It is a hint: nothing waits for it, nothing depends on it, a late or missing row is only slower.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_warm(persist_view pv, chain_view v, int K1, const int32_t *__restrict__ gram, int P, int ahead,
                                              int per_xcd, int *__restrict__ sink)
{
    __shared__ int s_rank;
    const int t = threadIdx.x;
    if (t == 0) {
        unsigned my;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my));
        my &= 15u;
        unsigned want = 0;
        const unsigned long long t0 = wall_clock64();
        while ((want = ld_flag(pv.flags + HB_FLAG_XCC)) == 0u) {
            if (ld_flag(pv.flags + HB_FLAG_ABORT) || ld_flag(pv.flags + HB_FLAG_CHAIN_DONE) >= (unsigned)pv.npanels ||
                wall_clock64() - t0 > 100000000ull) break; // (1 s: the chain never started)
            __builtin_amdgcn_s_sleep(16);
        }
        s_rank = (want == my + 1u) ? (int)(blockIdx.x >> 3) % per_xcd : -1;
    }
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) return;
    const int np = pv.npanels, Lb = pv.Lb, Lg = pv.Lg;
    const size_t PP = (size_t)P * P, step = (size_t)(Lg + 2) * PP;
    const int quarter = P >> 2;            // int4 lanes per row
    const int rows_per_pass = 256 / quarter; // rows one instruction of this workgroup covers
    int acc = 0;
    for (int q = pv.p0; q < np; q++) {
        // pace: at most `ahead` panels in front of the chain's published progress
        unsigned done;
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            done = ld_flag(pv.flags + HB_FLAG_CHAIN_DONE);
            if ((int)done + ahead >= q || ld_flag(pv.flags + HB_FLAG_ABORT) || wall_clock64() - t0 > HB_TIMEOUT_TICKS) break;
            __builtin_amdgcn_s_sleep(32);
        }
        if ((int)done >= np || ld_flag(pv.flags + HB_FLAG_ABORT) || (int)done + ahead < q) break;
        if (q < (int)done) continue; // the chain is already past this panel
        if (rank == (q % per_xcd)) {
            // the panel's exact per-marker data (what its candidates fetch at the opening): 8 P bytes per array
            const size_t j0 = (size_t)q * P;
            for (int i = t * 2; i < P; i += 512) { // 16 bytes per lane
                const double2 a = *reinterpret_cast<const double2 *>(v.g + j0 + i), b = *reinterpret_cast<const double2 *>(v.xpx + j0 + i);
                acc += (int)(a.x + a.y + b.x + b.y);
                for (int c = 0; c < K1; c++) {
                    const double2 x = *reinterpret_cast<const double2 *>(v.thr + (size_t)c * v.m_pad + j0 + i);
                    const double2 y = *reinterpret_cast<const double2 *>(v.invv + (size_t)c * v.m_pad + j0 + i);
                    const double2 z = *reinterpret_cast<const double2 *>(v.sdz + (size_t)c * v.m_pad + j0 + i);
                    acc += (int)(x.x + y.y + z.x);
                }
            }
            for (int i = t * 4; i < P; i += 1024) acc += reinterpret_cast<const int4 *>(pv.slot_of + j0 + i)->x;
            if (v.ga && v.gcmax) // (the certified group chain's candidates fetch these with the rest)
                for (int i = t * 4; i < P; i += 1024) acc += reinterpret_cast<const int4 *>(v.ga + j0 + i)->x ^ reinterpret_cast<const int4 *>(v.gcmax + j0 + i)->x;
        }
        const int *hl = pv.hotpack + (size_t)q * HB_HS;
        const int cnt = max(hl[0], hl[1]); // (with and without a slot in the row cache)
        const int lmax = min(Lb, np - 1 - q);
        const int nitem = cnt * (1 + lmax);
        const int32_t *gp = gram + (size_t)q * (Lg + 1) * PP;
        const int32_t *fwd = gram + ((size_t)(q + 1) * (Lg + 1) + 1) * PP;
        const int sub = t / quarter, col = (t - sub * quarter) * 4;
        for (int it = rank * rows_per_pass + sub; it < nitem; it += per_xcd * rows_per_pass) {
            const int mi = it / (1 + lmax), l = it - mi * (1 + lmax);
            const int k = hl[4 + mi];
            const int32_t *src = (l == 0 ? gp : fwd + (size_t)(l - 1) * step) + (size_t)k * P + col;
            const int4 x = *reinterpret_cast<const int4 *>(src);
            acc += x.x ^ x.y ^ x.z ^ x.w;
        }
    }
    if (acc == 0x5a5a5a5a) sink[0] = acc; // (keeps the loads)
}

