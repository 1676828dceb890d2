"""Development aid: HB_CERT=0 against 1 on the shape of tests/test_gpu_depth.py — where and by how much do the two runs differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_depth import geno, pheno
rng = np.random.default_rng(20250929)
n, m = 2048, 32768
X = geno(rng, n, m)
y = pheno(rng, X)
rng = np.random.default_rng(12)
g0 = np.where(rng.random(m) < 0.05, rng.normal(0, 0.03, m), 0.0)
out = []
for on in ("0", "1"):
    os.environ["HB_CERT"] = on
    res = []
    with H.Context(n, m, panel=512, seed=99) as c:
        c.upload(X)
        c.set_pipeline(1, 3, 7)
        c.build_gram()
        c.set_adaptive(True)
        c.set_layout(2, keep_int8=False)
        res.append(H.Bayes(y, None, "BayesCpi", [0.95, 0.05], verbose=False, ctx=c, niter=int(sys.argv[1]) if len(sys.argv) > 1 else 60, nburn=0, thin=1, seed=99))
        c.set_adaptive(False)
        c.set_pipeline(1, 3, 7)
        res.append(H.Bayes(y, None, "BayesCpi", [0.95, 0.05], verbose=False, ctx=c, niter=6, nburn=0, thin=1, seed=98, g_init=g0))
    out.append(res)
for a, b in zip(out[0], out[1]):
    A, B = a["MCMCsamples"]["alpha"], b["MCMCsamples"]["alpha"]
    print("records", A.shape[1], "events", a["timing"]["mean_events"], b["timing"]["mean_events"])
    for r in range(A.shape[1]):
        d = np.abs(A[:, r] - B[:, r])
        pat = ((A[:, r] != 0) != (B[:, r] != 0)).sum()
        if d.max() > 0 or pat:
            j = int(d.argmax())
            print("  record %d: %d entries differ, max |diff| %.3e at marker %d (%.6g vs %.6g), inclusion pattern differs in %d" % (r, (d > 0).sum(), d.max(), j, A[j, r], B[j, r], pat))
            break
    else:
        print("  identical")
