"""Development aid: long runs for stability (no device-side time-outs, sane posterior) — config 2 (BayesCpi n=10k m=100k, 5000 iterations)
and a few thousand sweeps at config-3 size; `soak.py dense`: BayesRR / A / L at config-3 size."""
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

if len(sys.argv) > 1 and sys.argv[1] == "dense":  # the models in which every marker moves (k_chain_dense, k_fold_dense)
    nfail = 0
    cases = (("BayesRR", 50000, 500000, 3000), ("BayesA", 50000, 500000, 1500), ("BayesL", 50000, 500000, 1500),
             ("BayesRR", 20000, 100000 + 300, 3000))  # (the last: ragged last panel)
    if len(sys.argv) > 2 and sys.argv[2] == "rr":
        cases = (("BayesRR", 50000, 500000, 3000),) * 3
    for model, n, m, it in cases:
        try:
            run(n, m, model, it, 500)
        except Exception as e:  # a device-side time-out: say so and go on, the count is the result
            nfail += 1
            print("  FAILED %s n=%d m=%d: %s" % (model, n, m, e), flush=True)
    print("dense soak: %d of %d runs failed" % (nfail, len(cases)), flush=True)
    sys.exit(1 if nfail else 0)
else:
    run(10000, 100000, "BayesCpi", 5000, 2500)
    run(50000, 500000, "BayesCpi", 3000, 1500)
    run(50000, 500000, "BayesR", 600, 300)
