"""Development aid: BayesRR / A / L at panel 512 (k_chain_dense + k_fold_dense) draw for draw against the oracle.
usage: dense_check.py [m] [geometries "Lv,D ..."]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from oracle import oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_depth import geno, pheno, _compare
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8192 + 100
geos = [tuple(int(v) for v in g.split(",")) for g in (sys.argv[2] if len(sys.argv) > 2 else "2,1 1,1 2,2 1,2 3,1").split()]
rng = np.random.default_rng(11)
n = 2048
X = geno(rng, n, m); y = pheno(rng, X)
bad = 0
for model in ("BayesRR", "BayesA", "BayesL"):
    kw = dict(niter=6, nburn=2, thin=2, seed=31337)
    ref = O.bayes(y, X, model, [0.95, 0.05], rng=O.RNG_PHILOX, store_alpha=True, **kw)
    for geo in geos:
        with H.Context(n, m, panel=512, seed=31337) as c:
            c.upload(X); c.set_pipeline(1, *geo)
            r = H.Bayes(y, None, model, [0.95, 0.05], verbose=False, ctx=c, **kw)
        try:
            _compare(r, ref, tol=1e-6 if model == "BayesL" else 1e-9)
            a, b = r["MCMCsamples"]["alpha"], ref["s_alpha"]
            print(model, geo, "OK  max rel diff %.2e  events %.0f" % (np.max(np.abs(a - b) / (np.abs(b) + 1e-300)), r["timing"]["mean_events"]), flush=True)
        except AssertionError as e:
            bad += 1
            print(model, geo, "FAIL", str(e)[:600], flush=True)
sys.exit(1 if bad else 0)
