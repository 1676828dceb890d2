"""Development aid: config-5-shaped sanity run (n = 200k individuals, BayesB), invariants only."""
import sys, os, ctypes as ct, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import check, BayesArgs, RunInfo
import bench
n, m = 200000, 20480
model = sys.argv[1] if len(sys.argv) > 1 else "BayesB"
c = H.Context(n, m); c.set_pipeline(*bench.PIPELINE.get(model, (1, 1, 1))); c.generate(7, 1000)
y = bench.synth_phenotype(c, n, m, 0, m, 7, None, model)
a = BayesArgs(); a.n, a.m = n, m; yv = np.ascontiguousarray(y); a.y = yv.ctypes.data; a.model = model.encode()
pv = np.array([0.95, 0.05]); a.Pi, a.n_pi = pv.ctypes.data, pv.size
a.niter, a.nburn, a.thin = 65, 0, 5; a.seed = 1; a.ctx = c.h
run = ct.c_void_p(); check(c.L.hb_run_create(ct.byref(a), ct.byref(run)))
fin = ct.c_int32(); info = RunInfo()
t0 = time.time(); check(c.L.hb_run_step(run, 60, ct.byref(fin))); dt = time.time() - t0
check(c.L.hb_run_state(run, ct.byref(info)))
r, u = c.get_residual(); g = c.get_effects()[0]
xg = np.zeros(n); check(c.L.hb_ctx_matvec(c.h, np.ascontiguousarray(g).ctypes.data, xg.ctypes.data))
print("n=%d m=%d %s: %.2f ms/sweep, nnz %d, moves/sweep %.0f, max|u - Xg| = %.3e, max|r + u - (y - mu)| = %.3e" % (
    n, m, model, dt / 60 * 1e3, info.nnz, info.mean_events, np.max(np.abs(u - xg)), np.max(np.abs(r + u - (y - info.mu)))))
