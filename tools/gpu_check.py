"""Quick on-GPU diagnostic (not a test): kernel-level checks against numpy and the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from oracle import oracle as O

rng = np.random.default_rng(1)
n, m = 1000, 1500
p = rng.uniform(0.05, 0.5, m)
X = ((rng.random((n, m)) < p).astype(np.int8) + (rng.random((n, m)) < p).astype(np.int8))
X[:, 7] = 2
X = np.asfortranarray(X)
c = H.Context(n, m, panel=256, precise=True)
c.upload(X)
xpx, vx, sumvx, nvar0 = c.marker_stats()
Xd = X.astype(np.float64)
print("xpx exact", np.array_equal(xpx, (Xd**2).sum(0)), "vx err", np.max(np.abs(vx - Xd.var(0, ddof=1))), "nvar0", nvar0)
print("download equal", np.array_equal(c.download(), X))
t = c.build_gram(); print("gram s", t)
P = c.panel
ok = True
for pidx in range((m + P - 1) // P):
    G = c.gram(pidx)
    cols = X[:, pidx*P:(pidx+1)*P].astype(np.int64)
    ref = cols.T @ cols
    k = cols.shape[1]
    ok &= np.array_equal(G[:k, :k], ref)
print("gram exact", ok)
r = rng.normal(0, 1, n)
c.set_residual(r, np.zeros(n))
d = c.dot()
ref = Xd.T @ r
print("dot precise relerr", np.max(np.abs(d - ref)) / np.max(np.abs(ref)))
c2 = H.Context(n, m, panel=128, precise=False); c2.upload(X); c2.set_residual(r, np.zeros(n))
d2 = c2.dot(); print("dot f32 relerr", np.max(np.abs(d2 - ref)) / np.max(np.abs(ref)))
c.close(); c2.close()

beta = np.zeros(m); beta[rng.choice(m, 15, replace=False)] = rng.normal(0, 1, 15)
y = Xd @ beta + rng.normal(0, 1.5, n)
for model, Pi, fold in [("BayesCpi", [0.95, 0.05], None), ("BayesC", [0.9, 0.1], None), ("BayesRR", [0.95, 0.05], None),
                        ("BayesA", [0.95, 0.05], None), ("BayesBpi", [0.95, 0.05], None), ("BayesL", [0.95, 0.05], None),
                        ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2])]:
    kw = dict(model=model, Pi=Pi, fold=fold, niter=12, nburn=4, thin=2, seed=99)
    t0 = time.time()
    try:
        got = H.Bayes(y, X, verbose=False, precise=True, **kw)
    except Exception as e:
        print(model, "FAILED", e); continue
    t1 = time.time()
    ref = O.bayes(y, X, rng=O.RNG_PHILOX, store_alpha=True, **kw)
    a, b = got["MCMCsamples"]["alpha"], ref["s_alpha"]
    err = np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))
    print("%-9s alpha relerr %.2e  Vg %.6g/%.6g Ve %.6g/%.6g pi %s/%s pipdiff %.2e  gpu %.2fs ev %.1f" % (
        model, err, got["Vg"], ref["Vg"], got["Ve"], ref["Ve"], np.round(got["pi"], 5), np.round(ref["pi"], 5),
        np.max(np.abs(got["pip"] - ref["pip"])), t1 - t0, got["timing"]["mean_events"]))
