"""Inside the pipeline's mat-vec launches: when do the update rows, the finalize blocks and the tiles of a launch end, and how long is
the gap to the next launch? Block stamps (hb_ctx_set_profiling bit 3) of stationary BayesCpi sweeps at n = 50k, m = 500k.
   python tools/launch_roles.py [bits] [Lv]"""
import ctypes as ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
import bench as B

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Lv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
MODEL = os.environ.get("HB_ROLES_MODEL", "BayesCpi")   # BayesR: one panel per launch, int8 columns (python tools/launch_roles.py 8 2)
D = int(os.environ.get("HB_ROLES_D", "1" if MODEL == "BayesR" else "7"))  # (BayesRR / A / L: HB_ROLES_D=2, python tools/launch_roles.py 8 2)
n, m = 50000, 500000
L = H.lib()
L.hb_ctx_debug_launch_stamps.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]
from hibayes_amd._lib import BayesArgs, check
with H.Context(n, m, seed=20240901) as c:
    c.generate(20240901, 1000)
    y = B.synth_phenotype(c, n, m, 0, m, 20240901, None, MODEL)
    c.set_pipeline(1, Lv, D)
    c.build_gram()
    c.set_adaptive(True)
    if bits == 2:
        c.set_layout(2, keep_int8=False)
    a = BayesArgs()
    a.n, a.m = n, m
    yv = np.ascontiguousarray(y); a.y = yv.ctypes.data
    a.model = MODEL.encode()
    Pi_, fold_ = B.prior(MODEL)
    pv = np.array(Pi_); a.Pi, a.n_pi = pv.ctypes.data, pv.size
    if fold_ is not None:
        fv = np.array(fold_, dtype=float); a.fold, a.n_fold = fv.ctypes.data, fv.size
    a.niter, a.nburn, a.thin = 340, 0, 5
    a.seed, a.precise, a.ctx = 20240901, 2, c.h
    run = ct.c_void_p(); check(L.hb_run_create(ct.byref(a), ct.byref(run)))
    fin = ct.c_int32()
    check(L.hb_run_step(run, 300, ct.byref(fin)))
    c.set_profiling(8)
    check(L.hb_run_step(run, 3, ct.byref(fin)))
    s = {"n_events": -1}
    st = c.matvec_stamps()
    npan = (m + c.panel - 1) // c.panel
    ng = (npan + D - 1) // D
    nupd = (n + 255) // 256
    nfin = D * c.panel // 64
    buf = np.zeros(2 * 4608, dtype=np.uint64)
    nb = ct.c_int()
    rows = []
    for g in range(ng):
        H._lib.check(L.hb_ctx_debug_launch_stamps(c.h, g, buf.ctypes.data, 4608, ct.byref(nb)))
        k = nb.value
        if k < nupd + nfin + 100 or g < 4:
            continue
        a = buf[:2 * k].reshape(k, 2).astype(np.int64)
        ok = a[:, 0] > 0
        s0 = a[ok, 0].min()
        fi, up, ti = a[:nfin][ok[:nfin]], a[nfin:nfin + nupd][ok[nfin:nfin + nupd]], a[nupd + nfin:][ok[nupd + nfin:]]  # (block roles by index: finalize, update, tiles)
        if len(rows) == 20:  # one launch in detail: when do its tiles start?
            ts = np.sort(ti[:, 0] - s0) * 1e-2
            print("  launch %d in detail: %d tiles; start times (us) p50 %.2f p90 %.2f p95 %.2f p98 %.2f max %.2f; tiles that start later than 2 us: %d; update blocks start p50 %.2f max %.2f, end p50 %.2f" % (
                g, len(ts), np.percentile(ts, 50), np.percentile(ts, 90), np.percentile(ts, 95), np.percentile(ts, 98), ts.max(), int((ts > 2.0).sum()),
                np.percentile(up[:, 0] - s0, 50) * 1e-2, (up[:, 0] - s0).max() * 1e-2, np.percentile(up[:, 1] - s0, 50) * 1e-2))
            late = np.argsort(ti[:, 0])[-12:]
            print("    the twelve latest tiles: index in the launch %s start %s" % ((late + nupd + nfin).tolist(), np.round((ti[late, 0] - s0) * 1e-2, 2).tolist()))
        rows.append((s0, up[:, 1].max() - s0 if len(up) else 0, (up[:, 1] - up[:, 0]).mean() if len(up) else 0, fi[:, 1].max() - s0 if len(fi) else 0,
                     ti[:, 1].max() - s0, np.percentile(ti[:, 1] - s0, 50), (ti[:, 1] - ti[:, 0]).mean(), a[ok, 1].max(), ti[:, 0].max() - s0))
    r = np.array(rows, dtype=np.float64)
    gap = r[1:, 0] - r[:-1, 7]
    us = 1e-2  # 100 MHz ticks -> us
    print("%s, " % MODEL + "bits %d, (Lv, D) = (%d, %d): %d full launches, %.2f us each in situ, moves per sweep %d" % (bits, Lv, D, len(r), st["avg_ms"] * 1e3, s["n_events"]))
    print("  update rows: last one ends %.2f us after the launch's first block starts (a block lives %.2f us)" % (r[:, 1].mean() * us, r[:, 2].mean() * us))
    print("  finalize blocks: last one ends at %.2f us" % (r[:, 3].mean() * us))
    print("  tiles: last one STARTS at %.2f us, half of them have ended at %.2f us, last one ends at %.2f us (a tile lives %.2f us)" % (
        r[:, 8].mean() * us, r[:, 5].mean() * us, r[:, 4].mean() * us, r[:, 6].mean() * us))
    print("  launch ends at %.2f us; gap to the next launch's first block %.2f us (p90 %.2f)" % ((r[:, 7] - r[:, 0]).mean() * us, gap.mean() * us, np.percentile(gap, 90) * us))
    late = r[:, 1] > r[:, 4]
    print("  launches whose update rows end after their last tile: %d of %d (by %.2f us on average)" % (late.sum(), len(r), ((r[late, 1] - r[late, 4]).mean() * us) if late.any() else 0))
    if os.environ.get("HB_DEBUG_ABORT"):
        L.hb_ctx_debug_ldiag.argtypes = [ct.c_void_p, ct.c_void_p]
        ld = np.zeros((npan + 2) * 4, dtype=np.uint64)
        H._lib.check(L.hb_ctx_debug_ldiag(c.h, ld.ctypes.data))
        w = ld.reshape(-1, 4)[4:ng - 2, 3]
        w = w[w != 0]
        ph = np.stack([(w >> np.uint64(16 * k)) & np.uint64(0xffff) for k in range(4)], axis=1).astype(np.float64) * us
        print("  update block 64 of each launch, phases in us (mean / p90): poll counts+bound %.2f / %.2f | move lists %.2f / %.2f | columns+sums %.2f / %.2f | stores %.2f / %.2f" % (
            ph[:, 0].mean(), np.percentile(ph[:, 0], 90), ph[:, 1].mean(), np.percentile(ph[:, 1], 90), ph[:, 2].mean(), np.percentile(ph[:, 2], 90), ph[:, 3].mean(), np.percentile(ph[:, 3], 90)))
