"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/hb_oracle.c).

The reference has no test-suite and cannot be built or run here (no R / Rcpp / Armadillo), so
these vectors come from the oracle, whose pins are listed in oracle/hb_oracle.h.  They freeze the
oracle's behaviour so that (a) an accidental change of the oracle is caught on CPU and (b) the GPU
path is compared with committed numbers, not only with a live oracle.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import hibayes_amd as H  # noqa: E402  (host-side loaders only; no GPU needed)

HERE = os.path.dirname(os.path.abspath(__file__))

MODELS = [("BayesCpi", [0.95, 0.05], None), ("BayesC", [0.9, 0.1], None), ("BayesRR", [0.95, 0.05], None),
          ("BayesA", [0.95, 0.05], None), ("BayesBpi", [0.95, 0.05], None), ("BayesB", [0.9, 0.1], None),
          ("BayesL", [0.95, 0.05], None), ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2])]


def demo_slice():
    d = os.path.join(HERE, "demo", "demo")
    pl = H.read_plink(d)
    phe = H.read_table(d + ".phe")
    ids = [r[1] for r in pl["fam"]]
    pos = {v: i for i, v in enumerate(phe["id"])}
    rows = [i for i, v in enumerate(ids) if v in pos and phe["T1"][pos[v]] is not None]
    y = np.array([float(phe["T1"][pos[ids[i]]]) for i in rows])
    return y, np.asfortranarray(pl["geno"][rows, :]), rows, phe, ids, pos


def small_case(seed=11, n=400, m=700):
    rng = np.random.default_rng(seed)
    p = rng.uniform(0.05, 0.5, m)
    X = (rng.random((n, m)) < p).astype(np.int8) + (rng.random((n, m)) < p).astype(np.int8)
    X[:, 3] = 0
    X[:, 130] = 2
    beta = np.zeros(m)
    beta[rng.choice(m, 12, replace=False)] = rng.normal(0, 0.8, 12)
    y = X @ beta + rng.normal(0, 1.2, n)
    return np.asfortranarray(X), y


def sbayes_demo():
    """sumstat = the MAF, BETA, SE, NMISS columns of inst/extdata/demo.ma (R/sbayes.r:207); ldm = the variance-covariance matrix
    of the 600 demo genotypes with the 1 / n normalisation of ldmat() (src/tXXmat.cpp:165-180)."""
    d = os.path.join(HERE, "demo", "demo")
    G = H.read_plink(d)["geno"].astype(np.float64)
    ld = np.cov(G, rowvar=False, ddof=0)
    rows = [l.split() for l in open(d + ".ma")][1:]
    f = lambda x: float(x) if x != "NA" else np.nan
    return np.array([[f(r[3]), f(r[4]), f(r[5]), f(r[7])] for r in rows]), ld


def main():
    # 1. demo data, the roxygen example of ibrm (R/bayes.r:93-94): BayesCpi, 2000/1200/5
    y, M, rows, phe, ids, pos = demo_slice()
    r = O.bayes(y, M, "BayesCpi", [0.95, 0.05], niter=2000, nburn=1200, thin=5, rng=O.RNG_PHILOX, seed=666666,
                trace_iter=0)
    np.savez(os.path.join(HERE, "demo_bayescpi_philox.npz"),
             scal=np.array([r["Vg"], r["Ve"], r["h2"], r["mu"]]), pi=r["pi"], alpha=r["alpha"], pip=r["pip"],
             init=np.array([r["vary"], r["sumvx"], r["nvar0"], r["varg0"], r["s2varg"], r["vara0"], r["s2vara"],
                            r["vare0"], r["lambda2_0"], r["rate0"]]),
             xpx=r["xpx"], vx=r["vx"], trace_rhs=r["trace_rhs"][:64], trace_cls=r["trace_cls"][:64],
             trace_g=r["trace_g"][:64], s_Vg=r["s_Vg"], s_h2=r["s_h2"])
    # 2. demo data with covariates and random effects (README.md:130-133 formula), short chain
    season = [phe["season"][pos[ids[i]]] for i in rows]
    bwt = np.array([float(phe["bwt"][pos[ids[i]]]) for i in rows])
    lev = sorted(set(season))
    C = np.column_stack([[1.0 if s == l else 0.0 for s in season] for l in lev[1:]] + [bwt])
    R = np.array([[phe["loc"][pos[ids[i]]], phe["dam"][pos[ids[i]]]] for i in rows], dtype=object)
    r2 = O.bayes(y, M, "BayesCpi", [0.98, 0.02], Cmat=C, R=R, niter=300, nburn=100, thin=5, rng=O.RNG_PHILOX,
                 seed=666666)
    np.savez(os.path.join(HERE, "demo_full_formula_philox.npz"),
             scal=np.array([r2["Vg"], r2["Ve"], r2["h2"], r2["mu"]]), pi=r2["pi"], alpha=r2["alpha"], pip=r2["pip"],
             beta=r2["beta"], Vr=r2["Vr"], r=r2["r"], e=r2["e"], C=C, R=R.astype(str))
    # 3. every model on a small random case, few sweeps, draw-for-draw
    X, ys = small_case()
    out = {"X": X, "y": ys}
    for model, Pi, fold in MODELS:
        rr = O.bayes(ys, X, model, Pi, fold=fold, niter=16, nburn=6, thin=2, rng=O.RNG_PHILOX, seed=424242,
                     store_alpha=True)
        out[model + "_alpha"] = rr["s_alpha"]
        out[model + "_scal"] = np.array([rr["Vg"], rr["Ve"], rr["h2"], rr["mu"]])
        out[model + "_pi"] = rr["pi"]
        out[model + "_pip"] = rr["pip"]
    np.savez_compressed(os.path.join(HERE, "small_all_models_philox.npz"), **out)
    # 4. summary-level sampler (SBayesD) on the demo COJO file + the LD matrix of the demo genotypes, every model, few sweeps
    ss, ld = sbayes_demo()
    out = {}
    for model, Pi, fold in MODELS:
        rr = O.sbayes(ss, ld, model, Pi, fold=fold, niter=12, nburn=4, thin=2, rng=O.RNG_PHILOX, seed=2468, store_alpha=True)
        out[model + "_alpha"] = rr["s_alpha"]
        out[model + "_scal"] = np.array([rr["Vg"], rr["Ve"], rr["h2"]])
        out[model + "_pi"] = rr["pi"]
        out[model + "_pip"] = rr["pip"]
    np.savez_compressed(os.path.join(HERE, "sbayes_demo_philox.npz"), **out)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
