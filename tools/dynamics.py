"""Development aid: NumNZSnp / changed markers per sweep over time for the synthetic bench data."""
import sys, os, ctypes as ct, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import check, BayesArgs, RunInfo
import bench
n, m = int(sys.argv[1]), int(sys.argv[2]); model = sys.argv[3]; nsw = int(sys.argv[4]); every = int(sys.argv[5])
panel = int(sys.argv[6]) if len(sys.argv) > 6 else 0
c = H.Context(n, m, panel=panel); c.set_pipeline(*bench.PIPELINE.get(model, (1, 1, 1))); c.generate(20240901, 1000)
y = bench.synth_phenotype(c, n, m, 0, m, 20240901, None, model)
Pi, fold = ([0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2]) if model == "BayesR" else ([0.95, 0.05], None)
a = BayesArgs(); a.n, a.m = n, m; yv = np.ascontiguousarray(y); a.y = yv.ctypes.data; a.model = model.encode()
pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
if fold: fv = np.array(fold, dtype=float); a.fold, a.n_fold = fv.ctypes.data, fv.size
a.niter, a.nburn, a.thin = nsw + 5, 0, 5; a.seed = 1; a.ctx = c.h
run = ct.c_void_p(); check(c.L.hb_run_create(ct.byref(a), ct.byref(run)))
fin = ct.c_int32(); info = RunInfo(); prev_ev = 0.0; prev_it = 0; prev_ms = 0.0
for it in range(0, nsw, every):
    t0 = time.time(); check(c.L.hb_run_step(run, every, ct.byref(fin))); dt = time.time() - t0
    check(c.L.hb_run_state(run, ct.byref(info)))
    tot = info.mean_events * info.iter
    totm = info.mean_misses * info.iter
    print("redo(mean) %.1f" % info.mean_redo, end=" "); print("iter %4d nnz %7d events/sweep %8.1f misses %7.1f pi %s vara %.3f vare %.3f  %.2f ms/sweep" % (
        info.iter, info.nnz, (tot - prev_ev) / (info.iter - prev_it), (totm - prev_ms) / (info.iter - prev_it),
        np.round(info.pi[:len(Pi)], 4), info.vara, info.vare, dt / every * 1e3), flush=True)
    prev_ev, prev_it, prev_ms = tot, info.iter, totm
