// hb_mvp.hpp — round 6: the sweep's panel mat-vec as ONE persistent kernel (2-bit resident genotypes, matrix-core tiles).
// Part of the one translation unit hb_kernels.hip; included after hb_dotq2.hpp and hb_update.hpp.
//
// Why. A 3 584-column launch of k_dotq2m lives 12 us of which its tiles run 9: the rest is the ramp of a kernel that is too small for the chip, plus
// 1.6 us of dependent-dispatch gap — 140 times a sweep. Two launches in flight stream 21 % more (profiles/r06_overlap.txt), but this runtime cannot
// run the extra graph branches that takes. Here nothing is launched per group: NT tile workgroups and NU update workgroups stay resident for the whole
// sweep and walk its mat-vec groups.
//   * tile workgroup b (column group cg, row split sp — the mapping of a k_dotq2m launch) runs dotq2m512_tile for group g as soon as the residual
//     version it reads, g - Lv - 1, is complete; the last of a column group's tiles to finish (a ticket) reduces the group's digit-plane sums
//     to dsum[] — the finalize rows of the next launch, without the next launch;
//   * update workgroup u owns rows [256 u, 256 u + 256) in EVERY version: it applies group h's moves as soon as the chain has published them (the counts,
//     the bound and the lists are polled as before) and then signs version h off. Its own rows it reads back itself; the digit planes every tile
//     workgroup reads are written THROUGH to memory, into a buffer of their own per version: no cache on any XCD can hold a line of a version
//     before that version is written, and the kernel boundary at the sweep's start has dropped whatever the last sweep left.
// The integers summed, the order of the moves and the chain are the launches': the same sweep bit for bit.
#pragma once

struct mvp_view {
    dq_view v0;        // the first group's tile view (X2 at the range's first column, accq / gexp at its first group); a group is D * P columns on
    upd_view u0;       // the update view's constants (genotypes, move lists, flags, u)
    int ngroups, D, np, g0, Lv;   // groups of the range, panels per group, the range's end panel, absolute index of its first group, look-ahead
    int ufresh, usleep;           // the update workgroups' waits: the same two numbers (update_rows<true>)
    int tfresh, tsleep;           // a tile workgroup's wait for its version: every tfresh-th look at the memory side, tsleep naps of 64 x 64 cycles per look
    int ntile, nupd, nsplit;      // workgroups: [0, nupd) update, [nupd, nupd + ntile) tiles; tiles per column group
    int8_t *rqv;       // digit planes by version: slot s = version s - 1 (slot 0: the sweep's start), HB_ND * ld bytes each
    int *vexpv;        // ... and their exponents
    double *r, *mb;    // residual slots (two: ping-pong as under the launches), the chain's bounds
    float *r32;
    double *dsum;
    unsigned *ho;      // [0, ngroups]: update workgroups that signed version s off; then ngroups x 64 finalize tickets
    unsigned *rel;     // 64 cache lines (32 words apart): the newest complete version slot, announced by the last update workgroup to sign it off
    unsigned *flags;
    unsigned long long *stamp; // optional in-situ stamps: [group][HB_LSTAMP_BLOCKS][2]
};

template <bool SC>
__global__ __launch_bounds__(64) void k_mvp2(mvp_view m)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t ld = m.v0.ld;
    const int P = m.u0.P;
    if (b < m.nupd) {
        // ---- an update workgroup: its 256 rows, version after version ----
        for (int h = 0; h < m.ngroups; h++) {
            unsigned long long t0 = 0;
            if (m.stamp) t0 = wall_clock64();
            upd_view uq = m.u0;
            const int ga = m.g0 + h;
            uq.p0 = ga * m.D;
            uq.p1 = min(m.np, uq.p0 + m.D);
            const int sin = h == 0 ? 0 : (h & 1), sout = (h + 1) & 1; // (version h - 1 -> h: the launches' ping-pong)
            uq.r_in = m.r + (size_t)sin * ld;
            uq.r = m.r + (size_t)sout * ld;
            uq.r32 = m.r32 + (size_t)sout * ld;
            uq.rq = m.rqv + (size_t)(h + 1) * HB_ND * ld;
            uq.mbv = m.mb + (size_t)(1 + ga) * HB_MBS;
            uq.vexp_out = m.vexpv + h + 1;
            update_rows<true>(ld, uq, b, reinterpret_cast<int *>(smem), reinterpret_cast<double *>(smem + 2048), reinterpret_cast<int *>(smem + 2048 + 4096), nullptr, m.ufresh, m.usleep);
            if (ld_flag(m.flags + HB_FLAG_ABORT)) return;
            // (the written-through planes have reached memory when the counter is 0; the tickets are RELAXED on purpose: an acquire / release at agent scope
            // is a write-back and an INVALIDATION of the XCD's whole L2 — issued by 900 workgroups 140 times a sweep it kept every L2 busy dropping its lines,
            // and the chain workgroup's loads did not come back within the 100 ms time-out)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // sign the version off; the LAST update workgroup to do so announces it — on 64 cache lines of its own, so that the ~700 tile workgroups that
            // are (always: the chain sets the pace) waiting for a version poll 64 different lines at a dozen waves each instead of one word at 700:
            // polled at one address, the memory channel behind it served nothing else and the chain workgroup's own loads starved (every sweep timed out)
            unsigned tk = 0;
            if (lane == 0) tk = __hip_atomic_fetch_add(m.ho + h + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tk = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
            if (tk == (unsigned)m.nupd - 1u) st_flag(m.rel + (size_t)lane * 32, (unsigned)(h + 1));
            if (m.stamp && lane == 0) {
                unsigned long long *st = m.stamp + ((size_t)(ga + m.Lv) * HB_LSTAMP_BLOCKS + (size_t)b) * 2; // (where the launch that carried this update kept it)
                if (h + m.Lv < m.ngroups) { st[0] = t0; st[1] = wall_clock64(); }
            }
        }
        return;
    }
    // ---- a tile workgroup ----
    const int bt = b - m.nupd;
    unsigned *fin = m.ho + m.ngroups + 1;
    for (int g = 0; g < m.ngroups; g++) {
        const int ga = m.g0 + g;
        const int p0 = ga * m.D, p1 = min(m.np, p0 + m.D);
        const int ncg = (p1 - p0) * P / 64;
        if (bt >= ncg * m.nsplit) continue; // (a short last group)
        const int v = g - m.Lv - 1, s = v < 0 ? 0 : v + 1;
        if (v >= 0) { // the version this group's product is taken with: complete? (versions complete in order: the announced number only grows)
            const unsigned *rl = m.rel + (size_t)(bt & 63) * 32;
            unsigned seen = ld_flag(rl);
            if (seen < (unsigned)s) {
                const unsigned long long t0 = wall_clock64();
                for (unsigned looks = 0;; looks++) {
                    seen = ld_poll_flag(rl, looks, (unsigned)m.tfresh);
                    if (seen >= (unsigned)s) break;
                    if (ld_flag(m.flags + HB_FLAG_ABORT) || wall_clock64() - t0 > HB_TIMEOUT_TICKS) {
                        if (lane == 0) { st_flag(m.flags + HB_FLAG_ABORT, 1u); if ((bt & 63) == 0) hb_abort_log(m.flags, HB_LOG_WAIT_GE, true, 0xB00u + (unsigned)s, (unsigned)g, ((unsigned long long)ld_flag(m.flags + HB_FLAG_CHAIN_DONE) << 32) | seen); }
                        return;
                    }
                    __builtin_amdgcn_s_sleep(16);
                    for (int z = 0; z < m.tsleep; z++) __builtin_amdgcn_s_sleep(64);
                }
            }
        }
        unsigned long long t0 = 0;
        if (m.stamp) t0 = wall_clock64();
        dq_view vg = m.v0;
        vg.X2 = m.v0.X2 + (int64_t)g * m.D * P * m.v0.ld2;
        vg.rq = m.rqv + (size_t)s * HB_ND * ld;
        vg.vexp_in = m.vexpv + s;
        vg.gexp_out = m.v0.gexp_out + g;
        vg.accq = m.v0.accq + (int64_t)g * m.D * P;
        vg.ncg = ncg;
        dotq2m512_tile<SC, false>(vg, smem, bt);
        // the column group's sums are complete when its last tile's atomics are: that tile turns them into dots
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int cg = bt % ncg;
        unsigned tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(fin + (size_t)g * 64 + cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
        if (tk == (unsigned)m.nsplit - 1u) {
            const int E = ld_sc1(m.vexpv + s);
            const long long *acc = vg.accq;
            const int col = cg * 64 + lane;
            double a = 0.0;
#pragma unroll
            for (int k = HB_ND - 1; k >= 0; k--)
                a = fma(a, 256.0, (double)__hip_atomic_load(acc + (int64_t)k * m.v0.accstride + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            st_sc1(m.dsum + (size_t)p0 * P + col, ldexp(a, -E));
        }
        if (m.stamp && lane == 0) {
            unsigned long long *st = m.stamp + ((size_t)ga * HB_LSTAMP_BLOCKS + (size_t)b) * 2;
            st[0] = t0;
            st[1] = wall_clock64();
        }
    }
}
