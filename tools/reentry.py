"""Development probe: of the markers entering the model in a sweep, how many were in it during the last K sweeps?"""
import sys, os, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import check, BayesArgs
import bench
n, m, model, nsw = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
c = H.Context(n, m); c.generate(20240901, 1000)
y = bench.synth_phenotype(c, n, m, 0, m, 20240901, None, model)
Pi, fold = bench.prior(model)
a = BayesArgs(); a.n, a.m = n, m; yv = np.ascontiguousarray(y); a.y = yv.ctypes.data; a.model = model.encode()
pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
if fold: fv = np.array(fold, dtype=float); a.fold, a.n_fold = fv.ctypes.data, fv.size
a.niter, a.nburn, a.thin = nsw + 5, 0, 5; a.seed = 1; a.ctx = c.h
run = ct.c_void_p(); check(c.L.hb_run_create(ct.byref(a), ct.byref(run)))
fin = ct.c_int32()
hist = []
for it in range(nsw):
    check(c.L.hb_run_step(run, 1, ct.byref(fin)))
    g, trk, _ = c.get_effects()
    hist.append(trk != 0)
    if it >= 20 and it % 10 == 0:
        cur, prev = hist[-1], hist[-2]
        entered = cur & ~prev
        res = []
        for K in (2, 4, 8, 16):
            recent = np.zeros(m, bool)
            for h in hist[-1 - K:-1]: recent |= h
            res.append((K, round((entered & recent).sum() / max(1, entered.sum()), 3), int(recent.sum())))
        print("iter", it, "nnz", int(cur.sum()), "entered", int(entered.sum()), "left", int((prev & ~cur).sum()), "reentry frac by K (K, frac, |recent|):", res)
