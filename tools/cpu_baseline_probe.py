"""Development aid: the CPU baseline's threaded form on its own — oracle/hb_oracle.c on n x m doubles, one sweep-rate per thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
n, m = 50000, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(1)
p = rng.uniform(0.05, 0.5, m)
X = np.empty((n, m), order="F")
for j in range(0, m, 500):
    pj = p[j:j + 500]
    X[:, j:j + 500] = (rng.random((n, pj.size)) < pj).astype(np.float64) + (rng.random((n, pj.size)) < pj)
beta = np.zeros(m); beta[:4] = rng.normal(0, 1, 4)
y = X @ beta + rng.normal(0, 1, n)
for model in ("BayesCpi", "BayesRR"):
    for thr in (1, 2, 4, 8, 16, 32, 64, 128):
        r = O.bayes(y, X, model, [0.95, 0.05], niter=4, nburn=3, thin=1, threads=thr, seed=3)
        print("%s threads %3d: %.3f s per sweep of %d markers -> %.4f sweeps/s at m = 500k" % (model, thr, r["loop_seconds"] / 4, m, 4 / r["loop_seconds"] * m / 5e5), flush=True)
