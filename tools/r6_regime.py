"""sweeps/s against moves per sweep, from a cold start, per geometry — where the narrow per-panel chain and the group chain cross (BayesR).
   python tools/r6_regime.py MODEL "lv,d lv,d ..." total chunk"""
import os, sys, time, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import BayesArgs, check, RunInfo
import bench as B

model = sys.argv[1]
geos = [tuple(int(x) for x in g.split(",")) for g in sys.argv[2].split()]
total, chunk = int(sys.argv[3]), int(sys.argv[4])
n, m = 50000, 500000
L = H.lib()
ctx = H.Context(n, m, seed=20240901)
ctx.generate(20240901, 1000)
y = B.synth_phenotype(ctx, n, m, 0, m, 20240901, None, model)
Pi, fold = B.prior(model)
ctx.set_pipeline(1, max(g[0] for g in geos), max(g[1] for g in geos))
ctx.build_gram()
if os.environ.get("R6_BITS") == "2":
    ctx.set_layout(2, keep_int8=False)  # (the headline's layout: k_dotq2m)
for lv, d in geos:
    ctx.set_pipeline(1, lv, d)
    a = BayesArgs()
    a.n, a.m = n, m
    yv = np.ascontiguousarray(y); a.y = yv.ctypes.data
    a.model = model.encode()
    pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
    if fold is not None:
        fv = np.array(fold); a.fold, a.n_fold = fv.ctypes.data, fv.size
    a.niter, a.nburn, a.thin = total + 8, 0, 5
    a.seed, a.precise, a.ctx = 20240901, 2, ctx.h
    run = ct.c_void_p(); check(L.hb_run_create(ct.byref(a), ct.byref(run)))
    fin = ct.c_int32()
    prev = RunInfo(); check(L.hb_run_state(run, ct.byref(prev)))
    out = []
    for s0 in range(0, total, chunk):
        t0 = time.perf_counter()
        check(L.hb_run_step(run, chunk, ct.byref(fin)))
        dt = time.perf_counter() - t0
        cur = RunInfo(); check(L.hb_run_state(run, ct.byref(cur)))
        mv = (cur.mean_events * cur.iter - prev.mean_events * prev.iter) / chunk
        out.append("%d:%.0f mv %.1f/s" % (s0 + chunk, mv, chunk / dt))
        prev = cur
    L.hb_run_destroy(run)
    print("%s geometry (%d,%d) pipeline %s: " % (model, lv, d, ctx.pipeline()) + " | ".join(out), flush=True)
