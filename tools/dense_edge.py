import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import hibayes_amd as H
from oracle import oracle as O
from test_gpu_depth import geno, pheno, _compare
rng = np.random.default_rng(3)
for n, m in ((100, 300), (333, 600), (2048, 1536), (700, 5000)):
    X = geno(rng, n, m); y = pheno(rng, X, ncausal=20)
    for model in ("BayesRR", "BayesL"):
        kw = dict(niter=5, nburn=1, thin=2, seed=99)
        ref = O.bayes(y, X, model, [0.95, 0.05], rng=O.RNG_PHILOX, store_alpha=True, **kw)
        for bits in (8, 2):
            r = H.Bayes(y, X, model, [0.95, 0.05], verbose=False, panel=512, genotype_bits=bits, **kw)
            _compare(r, ref, tol=1e-6 if model == "BayesL" else 1e-9)
            print(n, m, model, bits, "OK", flush=True)
