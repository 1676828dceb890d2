"""Development aid: long runs for stability (no device-side time-outs, sane posterior) — config 2 (BayesCpi n=10k m=100k, 5000 iterations)
and a few thousand sweeps at config-3 size."""
import sys, os, time, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import check, BayesArgs, RunInfo, BayesOut
import bench

def run(n, m, model, niter, nburn, seed=20240901):
    c = H.Context(n, m); c.set_pipeline(*bench.PIPELINE.get(model, (1, 1, 1))); c.generate(seed, 1000)
    y = bench.synth_phenotype(c, n, m, 0, m, seed, None, model)
    Pi, fold = bench.prior(model)
    a = BayesArgs(); a.n, a.m = n, m; yv = np.ascontiguousarray(y); a.y = yv.ctypes.data; a.model = model.encode()
    pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
    if fold is not None: fv = np.array(fold, dtype=float); a.fold, a.n_fold = fv.ctypes.data, fv.size
    a.niter, a.nburn, a.thin = niter, nburn, 5; a.seed = 1; a.ctx = c.h
    run = ct.c_void_p(); check(c.L.hb_run_create(ct.byref(a), ct.byref(run)))
    fin = ct.c_int32(); info = RunInfo()
    t0 = time.time(); done = 0
    while done < niter:
        k = min(500, niter - done); check(c.L.hb_run_step(run, k, ct.byref(fin))); done += k
        check(c.L.hb_run_state(run, ct.byref(info)))
        print("  %s n=%d m=%d iter %5d: %.2f ms/sweep so far, nnz %d, vara %.3f vare %.3f redo/sweep %.1f" % (
            model, n, m, info.iter, (time.time() - t0) / done * 1e3, info.nnz, info.vara, info.vare, info.mean_redo), flush=True)
    r, u = c.get_residual()
    print("  done: h2-ish %.3f, max|r+u-(y-mu)| %.2e" % (info.vara / (info.vara + info.vare), np.max(np.abs(r + u - (y - info.mu)))), flush=True)
    c.L.hb_run_destroy(run); c.close()

run(10000, 100000, "BayesCpi", 5000, 2500)
run(50000, 500000, "BayesCpi", 3000, 1500)
run(50000, 500000, "BayesR", 600, 300)
