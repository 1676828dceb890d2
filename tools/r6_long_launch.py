"""Which mat-vec launch of a sweep is the long one (bench.py: in_situ.max_ms of the all-move legs is ~1.5-2 ms against a 36 us average), and what
does it wait for? Block stamps of every launch of one sweep.  python tools/r6_long_launch.py [MODEL] [burn]"""
import ctypes as ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
import bench as B
from hibayes_amd._lib import BayesArgs, check

MODEL = sys.argv[1] if len(sys.argv) > 1 else "BayesRR"
burn = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, m = 50000, 500000
L = H.lib()
L.hb_ctx_debug_launch_stamps.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]
with H.Context(n, m, seed=20240901) as c:
    c.generate(20240901, 1000)
    y = B.synth_phenotype(c, n, m, 0, m, 20240901, None, MODEL)
    geo = B.PIPELINE[MODEL]
    c.set_pipeline(*geo)
    c.build_gram()
    D = geo[2]
    a = BayesArgs()
    a.n, a.m = n, m
    yv = np.ascontiguousarray(y); a.y = yv.ctypes.data
    a.model = MODEL.encode()
    Pi_, fold_ = B.prior(MODEL)
    pv = np.array(Pi_); a.Pi, a.n_pi = pv.ctypes.data, pv.size
    if fold_ is not None:
        fv = np.array(fold_, dtype=float); a.fold, a.n_fold = fv.ctypes.data, fv.size
    a.niter, a.nburn, a.thin = burn + 10, 0, 5
    a.seed, a.precise, a.ctx = 20240901, 2, c.h
    run = ct.c_void_p(); check(L.hb_run_create(ct.byref(a), ct.byref(run)))
    fin = ct.c_int32()
    check(L.hb_run_step(run, burn, ct.byref(fin)))
    c.set_profiling(8)
    check(L.hb_run_step(run, 2, ct.byref(fin)))
    for rep in range(3):
        check(L.hb_run_step(run, 1, ct.byref(fin)))
        npan = (m + c.panel - 1) // c.panel
        ng = (npan + D - 1) // D
        nupd = (n + 63) // 64 if MODEL in ("BayesRR", "BayesA", "BayesL") else (n + 255) // 256
        buf = np.zeros(2 * 4608, dtype=np.uint64)
        nb = ct.c_int()
        rows = []
        for g in range(ng):
            check(L.hb_ctx_debug_launch_stamps(c.h, g, buf.ctypes.data, 4608, ct.byref(nb)))
            k = nb.value
            if k <= 0:
                continue
            s = buf[:2 * k].reshape(k, 2).astype(np.int64)
            ok = s[:, 0] > 0
            if not ok.any():
                continue
            s0, e1 = s[ok, 0].min(), s[ok, 1].max()
            dur = s[ok, 1] - s[ok, 0]
            long_blocks = np.flatnonzero(ok)[np.argsort(dur)[-3:]]
            rows.append((g, s0, e1, k, long_blocks.tolist(), np.sort(dur)[-3:].tolist(), np.median(dur)))
        r = sorted(rows, key=lambda x: x[2] - x[1])
        t0 = min(x[1] for x in rows)
        print("%s sweep %d: %d launches stamped, median duration %.1f us; the five longest:" % (MODEL, rep, len(rows), np.median([x[2] - x[1] for x in rows]) * 1e-2))
        for g, s0, e1, k, lb, ld, md in r[-5:]:
            print("   launch %4d of %d: starts %.1f us into the sweep's launches, lasts %.1f us; %d blocks, median block %.1f us; its three longest blocks: indices %s, %.1f / %.1f / %.1f us" % (
                g, ng, (s0 - t0) * 1e-2, (e1 - s0) * 1e-2, k, md * 1e-2, lb, ld[0] * 1e-2, ld[1] * 1e-2, ld[2] * 1e-2))
        gaps = sorted(((rows[i + 1][1] - rows[i][2]) * 1e-2, rows[i + 1][0]) for i in range(len(rows) - 1))
        print("   the three longest gaps between a launch's end and the next one's first block: %s" % [("%.1f us before launch %d" % g) for g in gaps[-3:]])
    L.hb_run_destroy(run)
