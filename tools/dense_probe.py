"""Development aid: where a BayesRR sweep at n = 50k, m = 500k, panel 512 spends its time. In-situ stamps of the mat-vec launches
(update rows riding in them) and, on the -DHB_STAMPS=1 build, the chain workgroup's step stamps (wave 0's clock after the opening
and after each of the eight barriers of a panel).   python tools/dense_probe.py [Lv,D] [m] [model]"""
import os, sys, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
import bench as B
from hibayes_amd._lib import check

geo = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,1").split(","))
n, m = int(os.environ.get('PN', '50000')), int(sys.argv[2]) if len(sys.argv) > 2 else 500000
model = sys.argv[3] if len(sys.argv) > 3 else "BayesRR"
import time
with H.Context(n, m, panel=512, seed=20240901) as c:
    c.generate(20240901, 1000)
    y = B.synth_phenotype(c, n, m, 0, m, 20240901, None, "BayesCpi")
    c.set_pipeline(1, *geo)
    c.build_gram()
    vare, varg = 0.5, 0.5 / (0.5 * m)
    c.set_effects(np.zeros(m), np.zeros(m, dtype=np.uint8))
    c.set_residual(y - y.mean(), np.zeros(n))
    kw = dict(logpi=(0.0, 0.0), lam=1.0, lam2=1.0, s2varg_df=varg * 4.0 * 0.5)
    for it in range(4):
        c.sweep(model, it, vare, varg, **kw)
    t0 = time.time()
    for it in range(4, 14):
        s = c.sweep(model, it, vare, varg, **kw)
    print("%s geo %s: %.2f ms per sweep, moves %.0f" % (model, geo, (time.time() - t0) * 100, s["n_events"]), flush=True)
    c.set_profiling(8)
    for it in range(14, 17):
        s = c.sweep(model, it, vare, varg, **kw)
        st = c.matvec_stamps()
    print("in situ: %.2f us per launch (min %.2f max %.2f), stream span %.3f ms" % (st["avg_ms"] * 1e3, st["min_ms"] * 1e3, st["max_ms"] * 1e3, st["span_ms"]), flush=True)
    L = c.L
    L.hb_ctx_debug_launch_stamps.argtypes = [ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p]
    buf = np.zeros(2 * 4608, dtype=np.uint64); nb = ct.c_int()
    nfin = c.panel // 64; nupd = ((n + 255) // 256 * 256) // (256 if os.environ.get("HB_DENSE_UPD") else 64)
    for g in (40, 41, 42):
        check(L.hb_ctx_debug_launch_stamps(c.h, g, buf.ctypes.data, 4608, ct.byref(nb)))
        k = nb.value; a = buf[:2 * k].reshape(k, 2).astype(np.int64); s0 = a[:, 0].min()
        fi, up, ti = a[:nfin], a[nfin:nfin + nupd], a[nfin + nupd:]
        f = lambda x: "start %.1f..%.1f end %.1f..%.1f (life %.1f)" % ((x[:, 0].min() - s0) / 100, (x[:, 0].max() - s0) / 100, (x[:, 1].min() - s0) / 100, (x[:, 1].max() - s0) / 100, (x[:, 1] - x[:, 0]).mean() / 100)
        print("launch %d: %d blocks, us from its first block: finalize %s | update (%d) %s | tiles (%d) %s" % (g, k, f(fi), len(up), f(up), len(ti), f(ti)), flush=True)
        if g == 41:
            qs = [0, 10, 25, 50, 75, 90, 100]
            print("   start-time quantiles (us) %s: update %s | tiles %s" % (qs, np.round(np.percentile((up[:, 0] - s0) / 100, qs), 1), np.round(np.percentile((ti[:, 0] - s0) / 100, qs), 1)))
            print("   end-time quantiles (us): update %s | tiles %s" % (np.round(np.percentile((up[:, 1] - s0) / 100, qs), 1), np.round(np.percentile((ti[:, 1] - s0) / 100, qs), 1)))
    c.set_profiling(0)
    ms, nl, nc = c.time_matvec(reps=3)
    print("isolated replay (no update rows): %.2f us per launch" % (ms * 1e3))
    if os.environ.get("STAMPS"):
        P = c.panel; npan = (m + P - 1) // P
        c.set_profiling(2)
        for it in range(17, 20):
            c.sweep(model, it, vare, varg, **kw)
        st = np.zeros((npan, 32), dtype=np.int64)
        c.L.hb_ctx_debug_stamps.argtypes = [ct.c_void_p, ct.c_void_p]; check(c.L.hb_ctx_debug_stamps(c.h, st.ctypes.data))
        a = st[20:-5]
        per = np.diff(st[20:-4, 11])
        print("per panel %d cycles; poll (11->0) %d; steps after the opening: %s; end(9->1) %d" % (
            per.mean(), (a[:, 0] - a[:, 11]).mean(), [int((a[:, 2 + s] - (a[:, 1 + s] if s else a[:, 0])).mean()) for s in range(8)], (a[:, 1] - a[:, 9]).mean()))
        w12 = a[:, 12] > 1; print('panels whose dot (lane 0) was there at the first look: %d of %d' % ((a[:, 12] == 1).sum(), len(a)))
        print("panels that polled: %d of %d; lane 0's dot seen valid %d cycles after the panel's top (then the wait is for fcorr / the other lanes)" % (w12.sum(), len(a), (a[w12, 12] - a[w12, 11]).mean() if w12.any() else 0))
        ref = a[:, 3]  # after barrier 1 = start of step 2
        print("step 2, cycles after its start, by wave: apply done %s; at the barrier %s; barrier released %d" % (
            [int((a[:, 24 + w] - ref).mean()) for w in range(8)], [int((a[:, 16 + w] - ref).mean()) for w in range(8)], (a[:, 4] - ref).mean()))
        print("step 2, its serial wave (wave 2), cycles after the step's start: enters %d | strips applied %d | serial pass starts %d | ends %d | at the barrier %d (a stamp costs ~400 cycles itself)" % (
            (a[:, 13] - ref).mean(), (a[:, 26] - ref).mean(), (a[:, 14] - ref).mean(), (a[:, 15] - ref).mean(), (a[:, 18] - ref).mean()))
        print("poll percentiles", np.percentile(a[:, 0] - a[:, 11], [10, 50, 90]).astype(int), "panel percentiles", np.percentile(per, [10, 50, 90]).astype(int))
