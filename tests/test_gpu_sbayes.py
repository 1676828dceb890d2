"""the summary-level sampler on a dense LD matrix (hb_sbayes_run; reference src/SBayesD.cpp:251-470) on the MI355X,
draw for draw against the oracle under the same Philox counters, and against the committed golden vectors."""
import os

import numpy as np
import pytest

import hibayes_amd as H
from oracle import oracle as O
from test_oracle_sbayes import MODELS, sdemo  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _compare(r, ref, tol=1e-9):
    a, b = r["MCMCsamples"]["alpha"], ref["s_alpha"]
    assert np.array_equal(a != 0, b != 0), "inclusion pattern differs in %d entries" % int(((a != 0) != (b != 0)).sum())
    np.testing.assert_allclose(a, b, rtol=tol, atol=1e-13 if tol < 1e-7 else 1e-8)   # (effects here are of order 1)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"]], [ref["Vg"], ref["Ve"], ref["h2"]], rtol=tol)
    np.testing.assert_allclose(r["pi"], ref["pi"], rtol=tol, atol=1e-14)
    np.testing.assert_allclose(r["MCMCsamples"]["pi"], ref["s_pi"], rtol=tol, atol=1e-14)
    np.testing.assert_allclose(r["pip"], ref["pip"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["r_hat"], ref["r_hat"], rtol=0, atol=1e-7 * max(1.0, np.abs(ref["r_hat"]).max()))
    assert r["n"] == ref["n"] and r["count_y"] == ref["count_y"] and r["nzct"] == ref["nzct"]


@pytest.mark.parametrize("model,Pi,fold", MODELS)
def test_demo_draw_for_draw_against_golden_and_live_oracle(sdemo, model, Pi, fold):
    ss, ld = sdemo["ss"], sdemo["ld"]          # m = 1000 (15.6 blocks of 64), 50 markers without statistics
    g = np.load(os.path.join(G, "sbayes_demo_philox.npz"))
    tol = 1e-6 if model == "BayesL" else 1e-9  # (1 / inverse-Gaussian(|g|) amplifies last-bit differences, as in the individual-level path)
    r = H.SBayesD(ss, ld, model, Pi, fold=fold, niter=12, nburn=4, thin=2, seed=2468, verbose=False)
    np.testing.assert_allclose(r["MCMCsamples"]["alpha"], g[model + "_alpha"], rtol=tol, atol=1e-13 if tol < 1e-7 else 1e-8)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"]], g[model + "_scal"], rtol=tol)
    np.testing.assert_allclose(r["pip"], g[model + "_pip"], rtol=0, atol=1e-12)
    kw = dict(fold=fold, niter=60, nburn=20, thin=4, seed=97)
    ref = O.sbayes(ss, ld, model, Pi, rng=O.RNG_PHILOX, store_alpha=True, **kw)
    _compare(H.SBayesD(ss, ld, model, Pi, verbose=False, **kw), ref, tol)


def test_ragged_size_windows_unsorted_fold_and_sbrm_defaults():
    rng = np.random.default_rng(5)
    n, m = 400, 203                                        # 3 blocks and a tail of 11 markers
    p = rng.uniform(0.1, 0.5, m)
    X = ((rng.random((n, m)) < p).astype(float) + (rng.random((n, m)) < p).astype(float))
    ld = np.cov(X, rowvar=False, ddof=0)
    beta = np.zeros(m)
    beta[rng.choice(m, 10, replace=False)] = rng.normal(0, 1, 10)
    y = X @ beta + rng.normal(0, 1.0, n)
    Xc = X - X.mean(0)
    b = (Xc * (y - y.mean())[:, None]).sum(0) / (Xc ** 2).sum(0)
    se = np.sqrt(((y - y.mean()) ** 2).sum() / (n - 2) / (Xc ** 2).sum(0))
    ss = np.column_stack([p, b, se, np.full(m, float(n))])
    ss[17, 1] = np.nan
    wind = (np.arange(m) // 20 + 1).astype(np.uint32)
    kw = dict(niter=80, nburn=30, thin=5, seed=11)
    ref = O.sbayes(ss, ld, "BayesCpi", [0.9, 0.1], windindx=wind, rng=O.RNG_PHILOX, store_alpha=True, **kw)
    r = H.SBayesD(ss, ld, "BayesCpi", [0.9, 0.1], windindx=wind, verbose=False, **kw)
    _compare(r, ref)
    np.testing.assert_allclose(r["gwas"], ref["gwas"], rtol=0, atol=1e-12)
    # BayesR with `fold` out of order: the chain of the sorted classes, pi reported in the caller's order (see hb_run.hip)
    Pi, fold = [0.875, 0.0625, 0.03125, 0.03125], [0, 1e-2, 1e-4, 1e-3]
    order = [0] + sorted(range(1, 4), key=lambda k: fold[k])
    ref = O.sbayes(ss, ld, "BayesR", [Pi[k] for k in order], fold=[fold[k] for k in order], rng=O.RNG_PHILOX, store_alpha=True, **kw)
    r = H.SBayesD(ss, ld, "BayesR", Pi, fold=fold, verbose=False, **kw)
    np.testing.assert_allclose(r["MCMCsamples"]["alpha"], ref["s_alpha"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(np.asarray(r["pi"])[order], ref["pi"], rtol=1e-9)
    # sbrm(): the defaults of R/sbayes.r:186-203 and the 8-column COJO table
    full = np.column_stack([np.zeros((m, 3)), ss[:, 0], ss[:, 1], ss[:, 2], np.zeros(m), ss[:, 3]])
    f = H.sbrm(full, ld, method="BayesCpi", niter=40, nburn=10, verbose=False)
    assert f["n_records"] == 6 and f["model"] == "Summary level Bayesian model fit by [BayesCpi]" and np.isfinite(f["h2"])


@pytest.mark.parametrize("model,Pi,fold", [("BayesCpi", [0.7, 0.3], None), ("BayesB", [0.5, 0.5], None),
                                           ("BayesR", [0.6, 0.2, 0.15, 0.05], [0, 1e-3, 1e-2, 1e-1]), ("BayesRR", [0.95, 0.05], None)])
def test_groups_of_512_with_many_candidates_under_strong_ld(model, Pi, fold):
    """Several k_sb_group launches with more than 64 candidates each (several rounds per group, markers pushed over their threshold
    by an earlier move of the same round) and a ragged last group: blocks of 32 markers in strong LD, a third of them in the model."""
    rng = np.random.default_rng(31)
    n, m = 600, 1700                                       # 3 groups of 512 and a tail of 164
    p = np.repeat(rng.uniform(0.1, 0.5, (m + 31) // 32), 32)[:m]
    X = np.empty((n, m))
    for j in range(m):
        fresh = (rng.random(n) < p[j]).astype(float) + (rng.random(n) < p[j])
        X[:, j] = fresh if j % 32 == 0 else np.where(rng.random(n) < 0.92, X[:, j - 1], fresh)
    ld = np.cov(X, rowvar=False, ddof=0)
    beta = np.zeros(m)
    causal = rng.choice(m, 120, replace=False)
    beta[causal] = rng.normal(0, 0.5, causal.size)
    y = X @ beta + rng.normal(0, 1.0, n)
    Xc = X - X.mean(0)
    xx = (Xc ** 2).sum(0)
    b = (Xc * (y - y.mean())[:, None]).sum(0) / xx
    se = np.sqrt(((y - y.mean()) ** 2).sum() / (n - 2) / xx)
    ss = np.column_stack([X.mean(0) / 2, b, se, np.full(m, float(n))])
    kw = dict(fold=fold, niter=12, nburn=4, thin=2, seed=77)
    ref = O.sbayes(ss, ld, model, Pi, rng=O.RNG_PHILOX, store_alpha=True, **kw)
    r = H.SBayesD(ss, ld, model, Pi, verbose=False, **kw)
    assert (ref["s_alpha"][:, -1] != 0).sum() > 200       # (the regime the test is about)
    _compare(r, ref, 1e-8)
