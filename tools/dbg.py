import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, hibayes_amd as H
g = np.load("/root/repo/tests/golden/small_all_models_philox.npz")
for panel in (64,):
    r = H.Bayes(g["y"], g["X"], "BayesCpi", [0.95, 0.05], niter=16, nburn=6, thin=2, seed=424242, verbose=False, precise=True, panel=panel)
    print({k: r[k] for k in r if k.startswith("mean_") or k in ("NumNZSnp",)})
    a, b = r["MCMCsamples"]["alpha"], g["BayesCpi_alpha"]
    print(os.environ.get("HB_GRAPH"), os.environ.get("HB_CANDF"), os.environ.get("HB_PIPELINE"), panel, "nnz gpu", (a != 0).sum(0), "nnz ref", (b != 0).sum(0), flush=True)
