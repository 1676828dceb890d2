"""Parity of the GPU sampler with the CPU oracle, through the C ABI (hb_bayes_run):
Level 1  draw-for-draw: same Philox counters => identical inclusion flags and effects;
Level 2  statistical: blocked GPU chain (Philox; every mat-vec arithmetic: exact fixed point, fp64 FMA, fp32 image) vs the
         sequential R-stream oracle, several seeds;
plus size-independent invariants at the BASELINE sizes."""
import os

import numpy as np
import pytest

import hibayes_amd as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MODELS = [("BayesCpi", [0.95, 0.05], None), ("BayesC", [0.9, 0.1], None), ("BayesRR", [0.95, 0.05], None),
          ("BayesA", [0.95, 0.05], None), ("BayesBpi", [0.95, 0.05], None), ("BayesB", [0.9, 0.1], None),
          ("BayesL", [0.95, 0.05], None), ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2])]


@pytest.mark.parametrize("model,Pi,fold", MODELS)
@pytest.mark.parametrize("panel", [0, 64, 512])
@pytest.mark.parametrize("precise", [1, 2])   # 1: fp64 FMA mat-vec; 2: exact fixed-point digits (the default of Bayes())
def test_draw_for_draw_against_golden(model, Pi, fold, panel, precise):
    g = np.load(os.path.join(G, "small_all_models_philox.npz"))
    r = H.Bayes(g["y"], g["X"], model, Pi, fold=fold, niter=16, nburn=6, thin=2, seed=424242, verbose=False,
                precise=precise, panel=panel)
    tol = 1e-6 if model == "BayesL" else 1e-9   # BayesL's inverse-Gaussian draw amplifies last-bit differences
    a, b = r["MCMCsamples"]["alpha"], g[model + "_alpha"]
    assert np.array_equal(a != 0, b != 0)                                # identical inclusion pattern
    np.testing.assert_allclose(a, b, rtol=tol, atol=1e-12)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"], r["mu"]], g[model + "_scal"], rtol=tol)
    np.testing.assert_allclose(r["pi"], g[model + "_pi"], rtol=tol, atol=1e-14)
    np.testing.assert_allclose(r["pip"], g[model + "_pip"], rtol=0, atol=1e-12)
    assert r["alpha"][3] == 0 and r["alpha"][130] == 0                   # monomorphic markers skipped (vx == 0)


def test_demo_example_draw_for_draw(demo):
    # the ibrm() roxygen example (R/bayes.r:93-94) on the demo data, BayesCpi 2000/1200/5
    g = np.load(os.path.join(G, "demo_bayescpi_philox.npz"))
    r = H.Bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], niter=2000, nburn=1200, thin=5, seed=666666,
                verbose=False, precise=True)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"], r["mu"]], g["scal"], rtol=1e-7)
    np.testing.assert_allclose(r["alpha"], g["alpha"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(r["pip"], g["pip"], rtol=0, atol=1e-12)
    assert r["nzct"] == 800 and r["n_records"] == 160
    # PVE (reference README.md:188): vx * alpha^2 / var(y) — by-product of the same outputs
    pve = g["vx"] * r["alpha"] ** 2 / demo["y"].var(ddof=1)
    np.testing.assert_allclose(pve, g["vx"] * g["alpha"] ** 2 / demo["y"].var(ddof=1), rtol=1e-5, atol=1e-12)


def test_ibrm_full_formula_with_covariates_and_random_effects(demo):
    # README.md:130-133: T1 ~ season + bwt + (1 | loc) + (1 | dam)
    g = np.load(os.path.join(G, "demo_full_formula_philox.npz"), allow_pickle=True)
    pl, phe = demo["plink"], demo["phe"]
    fit = H.ibrm("T1 ~ season + bwt + (1 | loc) + (1 | dam)", data=phe, M=pl["geno"], M_id=demo["ids"],
                 method="BayesCpi", Pi=[0.98, 0.02], niter=300, nburn=100, thin=5, seed=666666, verbose=False,
                 precise=True)
    assert fit["beta_names"] == ["seasonSpring", "seasonSummer", "seasonWinter", "bwt"]
    np.testing.assert_allclose(fit["beta"], g["beta"], rtol=1e-7)
    np.testing.assert_allclose(fit["Vr"], g["Vr"], rtol=1e-7)
    np.testing.assert_allclose(fit["r"]["Estimation"], g["r"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(fit["alpha"], g["alpha"], rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(fit["e"]["e"], g["e"], rtol=1e-7, atol=1e-8)
    assert len(fit["g"]["gebv"]) == 600 and len(fit["e"]["id"]) == 300
    np.testing.assert_allclose(fit["g"]["gebv"], pl["geno"].astype(float) @ fit["alpha"], rtol=1e-10, atol=1e-12)
    # MCMCsamples$g = M %*% MCMCsamples$alpha over all 600 genotyped individuals, gebv = its row means (R/bayes.r:303-308)
    gs = fit["MCMCsamples"]["g"]
    assert gs.shape == (600, 40)
    np.testing.assert_allclose(gs, pl["geno"].astype(float) @ fit["MCMCsamples"]["alpha"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fit["g"]["gebv"], gs.mean(axis=1), rtol=0, atol=0)


def test_readme_fit_on_the_gpu_inside_the_printed_posterior(demo):
    """reference README.md:130-172 (real hibayes, R's RNG): Vg 52.10 +- 13.08, h2 0.357 +- 0.081, pi1 0.927 +- 0.039,
    Ve 30.77 +- 6.32, Vr(loc) 8.10 +- 4.79, Vr(dam) 54.29 +- 10.10, fixed effects -21.919 -11.484 -11.576 2.399.
    tests/test_oracle_sampler.py reproduces that printout digit for digit with the oracle on R's stream; the GPU draws
    from Philox, so its chains are other draws from the same posterior: each of 4 seeds must land inside the printed
    mean +- 2 posterior SD (a chain's own Monte-Carlo error is ~SD/10), their average inside +- 1 SD, and the
    data-determined fixed effects within 0.5 of the printed values."""
    pl, phe = demo["plink"], demo["phe"]
    fits = [H.ibrm("T1 ~ season + bwt + (1 | loc) + (1 | dam)", data=phe, M=pl["geno"], M_id=demo["ids"], method="BayesCpi",
                   Pi=[0.98, 0.02], niter=20000, nburn=16000, thin=5, seed=s, verbose=False, store_alpha=False)
            for s in (666666, 1, 2, 3)]
    band = {"Vg": (52.10097, 13.084), "h2": (0.35748, 0.081), "Ve": (30.77, 6.323)}
    for k, (mean, sd) in band.items():
        v = np.array([f[k] for f in fits])
        assert (np.abs(v - mean) < 2 * sd).all() and abs(v.mean() - mean) < sd, (k, v)
    pi1 = np.array([f["pi"][0] for f in fits])
    assert (np.abs(pi1 - 0.92683) < 2 * 0.039).all() and abs(pi1.mean() - 0.92683) < 0.039
    vr = np.array([f["Vr"] for f in fits])
    assert (np.abs(vr - [8.10, 54.29]) < 2 * np.array([4.785, 10.096])).all()
    beta = np.array([f["beta"] for f in fits])
    assert (np.abs(beta - [-21.919, -11.484, -11.576, 2.399]) < 0.5).all(), beta
    assert fits[0]["n_records"] == 800 and len(fits[0]["g"]["gebv"]) == 600


def test_gwas_windows_wppa(demo):
    chrom = np.array([int(c) for c in demo["plink"]["map"]["Chr"]])
    wind = H.cutwind_by_num(chrom, demo["plink"]["map"]["Pos"], 50)
    kw = dict(niter=200, nburn=100, thin=5, seed=7)
    r = H.Bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], windindx=wind, verbose=False, precise=True, **kw)
    ref = O.bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], windindx=wind, rng=O.RNG_PHILOX, **kw)
    np.testing.assert_allclose(r["gwas"], ref["gwas"], rtol=0, atol=1e-12)
    assert r["gwas"].size == wind.max()


def test_error_texts_through_the_c_abi(demo):
    y, M = demo["y"], demo["M"]
    cases = [
        (dict(Pi=[0.5, 0.6]), "sum of Pi should be 1.", 1),
        (dict(Pi=[1.0, 0.0]), "all markers have no effect size.", 1),
        (dict(model="BayesR", Pi=[0.9, 0.05, 0.05]), "'fold' should be provided for BayesR model.", 1),
        (dict(Pi=[0.95, 0.05], dfvg=2.0), "dfvg should not be less than 2.", 1),
        (dict(Pi=[0.95, 0.05], niter=5, nburn=10), "Number of total iteration ('niter') shold be larger than burn-in ('nburn').", 1),
        (dict(model="BSLMM", Pi=[0.95, 0.05]), "BSLMM (Ki/Kival) is not part of the GPU path", 4),
        (dict(model="BayesR", Pi=[0.9, 0.05, 0.05], fold=[0, 1e-2, 1e-2]), "BayesR on the GPU path needs distinct 'fold' values for the non-null classes", 4),
    ]
    for kw, msg, status in cases:
        a = dict(model="BayesCpi", niter=4, nburn=2, thin=1, verbose=False)
        a.update(kw)
        with pytest.raises(H.HibayesError) as ei:
            H.Bayes(y, M, a.pop("model"), a.pop("Pi"), **a)
        assert str(ei.value) == msg and ei.value.status == status
    yy = y.copy()
    yy[0] = np.nan
    with pytest.raises(H.HibayesError, match="NAs are not allowed in y."):
        H.Bayes(yy, M, "BayesCpi", [0.95, 0.05], niter=4, nburn=2, thin=1, verbose=False)


@pytest.mark.parametrize("model,Pi,fold,precise", [("BayesCpi", [0.95, 0.05], None, 2), ("BayesCpi", [0.95, 0.05], None, 0),
                                                    ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], 2),
                                                    ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], 0)])
def test_statistical_parity_vs_r_stream_oracle(demo, model, Pi, fold, precise):
    """Level 2 (SURVEY.md §8 c): GPU = blocked chain, Philox, panel mat-vec in the default exact fixed point (precise = 2) and
    in the fp32 image of the residual (precise = 0: the mode keeps a multi-seed statistical test of its own); oracle = sequential
    chain, fp64, R's Mersenne-Twister stream. Tolerance: |difference of means over 4 seeds| <
    max(1 % relative, 3 x Monte-Carlo standard error from the across-seed spread)."""
    kw = dict(niter=3000, nburn=1000, thin=5)
    gpu = [H.Bayes(demo["y"], demo["M"], model, Pi, fold=fold, seed=s, verbose=False, store_alpha=False, precise=precise, **kw) for s in (11, 12, 13, 14)]
    ora = [O.bayes(demo["y"], demo["M"], model, Pi, fold=fold, rng=O.RNG_R, seed=s, **kw) for s in (21, 22, 23, 24)]

    def close(a, b, what):
        # with 4 + 4 chains the standardised difference is t-distributed with ~6 df (P(|t| > 3) = 2.4 %,
        # P(|t| > 6) = 0.1 %): per-marker vectors may exceed 3 SE in a few percent of entries by chance
        ma, mb = np.mean(a, axis=0), np.mean(b, axis=0)
        se = np.sqrt(np.var(a, axis=0, ddof=1) / len(a) + np.var(b, axis=0, ddof=1) / len(b))
        d = np.abs(ma - mb)
        bad3 = d > np.maximum(0.01 * np.abs(mb), 3.0 * se)
        bad6 = d > np.maximum(0.01 * np.abs(mb), 6.0 * se)
        if d.size == 1:
            assert not bad6.any() and (not bad3.any() or d[0] < 0.05 * abs(mb[0])), "%s: %r vs %r" % (what, ma, mb)
        else:
            assert bad3.mean() <= 0.05 and bad6.mean() <= 0.005, "%s: %d / %d of %d outside 3 / 6 SE" % (
                what, bad3.sum(), bad6.sum(), d.size)

    for k in ("Vg", "Ve", "h2", "mu"):
        close(np.array([[r[k]] for r in gpu]), np.array([[r[k]] for r in ora]), k)
    close(np.array([r["pi"] for r in gpu]), np.array([r["pi"] for r in ora]), "pi")
    close(np.array([r["pip"] for r in gpu]), np.array([r["pip"] for r in ora]), "pip")
    close(np.array([r["alpha"] for r in gpu]), np.array([r["alpha"] for r in ora]), "alpha")
    vy = demo["y"].var(ddof=1)
    vx = demo["M"].astype(float).var(0, ddof=1)
    close(np.array([vx * r["alpha"] ** 2 / vy for r in gpu]), np.array([vx * r["alpha"] ** 2 / vy for r in ora]), "PVE")


def test_fast_and_precise_matvec_agree_on_a_short_chain(demo):
    kw = dict(niter=40, nburn=20, thin=2, seed=5, verbose=False)
    a = H.Bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], precise=True, **kw)
    b = H.Bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], precise=False, **kw)
    same = (a["MCMCsamples"]["alpha"] != 0) == (b["MCMCsamples"]["alpha"] != 0)
    assert same.mean() > 0.999          # fp32 mat-vec may flip a decision sitting within 1e-6 of its threshold
    assert abs(a["Vg"] - b["Vg"]) < 0.02 * abs(a["Vg"])


def test_preloaded_context_serves_several_models(demo):
    import ctypes as ct
    from hibayes_amd._lib import BayesArgs, BayesOut, check
    with H.Context(300, 1000) as c:
        c.upload(demo["M"])
        out = {}
        for model in ("BayesCpi", "BayesRR"):
            a = BayesArgs()
            y = np.ascontiguousarray(demo["y"])
            Pi = np.array([0.95, 0.05])
            a.n, a.m, a.y, a.model = 300, 1000, y.ctypes.data, model.encode()
            a.Pi, a.n_pi, a.niter, a.nburn, a.thin, a.seed, a.ctx = Pi.ctypes.data, 2, 30, 10, 2, 3, c.h
            alpha = np.zeros(1000)
            o = BayesOut()
            o.alpha = alpha.ctypes.data
            check(c.L.hb_bayes_run(ct.byref(a), ct.byref(o)))
            out[model] = alpha
        ref = O.bayes(demo["y"], demo["M"], "BayesRR", [0.95, 0.05], niter=30, nburn=10, thin=2, rng=O.RNG_PHILOX, seed=3)
        # BayesRR through the same context; dot-product rounding differs (fp32 fast path) -> loose tolerance
        np.testing.assert_allclose(out["BayesRR"], ref["alpha"], rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("model", ["BayesCpi", "BayesR"])
def test_invariants_at_baseline_size(model):
    """n = 50k, m = 500k (BASELINE.json metric size): size-independent properties of the sweep.
    After any number of sweeps yadj + u == y - mean(y) shifted by the intercept moves, u == X g,
    class counts add up to the polymorphic markers, and every stored effect of a monomorphic marker is 0."""
    n, m = 50000, 500000
    rng = np.random.default_rng(0)
    with H.Context(n, m, seed=99) as c:
        c.generate(20240901, mono_every=1000)
        xpx, vx, sumvx, nvar0 = c.marker_stats()
        assert nvar0 >= m // 1000
        y0 = rng.normal(0, 1, n)
        c.set_residual(y0, np.zeros(n))
        vare, varg = 0.5, 0.5 / (0.05 * sumvx)
        if model == "BayesR":
            logpi, fold = np.log([0.95, 0.02, 0.02, 0.01]), [0, 1e-4, 1e-3, 1e-2]
        else:
            logpi, fold = np.log([0.95, 0.05]), [0, 0]
        tot = 0
        for it in range(3):
            s = c.sweep(model, it, vare, varg, logpi=logpi, fold=fold)
            assert s["class_count"].sum() == m - nvar0
            tot += s["n_events"]
        assert tot > 0
        r, u = c.get_residual()
        g, trk, _ = c.get_effects()
        np.testing.assert_allclose(r + u, y0, rtol=0, atol=1e-9)           # yadj = y - X g at every sweep boundary
        xg = np.zeros(n)
        H._lib.check(c.L.hb_ctx_matvec(c.h, g.ctypes.data, xg.ctypes.data))
        np.testing.assert_allclose(u, xg, rtol=0, atol=1e-8)
        assert not g[vx == 0].any() and not trk[vx == 0].any()
        assert np.array_equal(trk != 0, g != 0)
        assert s["sum_r2"] == pytest.approx((r * r).sum(), rel=1e-10)
        assert s["var_u"] == pytest.approx(u.var(ddof=1), rel=1e-9)


@pytest.mark.parametrize("K", [2, 3, 6, 8])
def test_bayesr_class_counts_draw_for_draw_against_the_oracle(K):
    # BayesR with other numbers of mixture classes than the default four (the chain kernel is instantiated for 1, <=3 and
    # <=7 non-null classes): the oracle is the checker, computed here under the same Philox counters
    g = np.load(os.path.join(G, "small_all_models_philox.npz"))
    fold = [0.0] + [10.0 ** e for e in np.linspace(-4, -1.5, K - 1)]
    Pi = [1.0 - (K - 1) / 16.0] + [1.0 / 16.0] * (K - 1)   # dyadic: the reference compares sum(Pi) with 1 exactly
    kw = dict(fold=fold, niter=16, nburn=6, thin=2, seed=424242)
    ref = O.bayes(g["y"], g["X"], "BayesR", Pi, rng=O.RNG_PHILOX, store_alpha=True, **kw)
    for panel in (64, 512):
        r = H.Bayes(g["y"], g["X"], "BayesR", Pi, verbose=False, precise=True, panel=panel, **kw)
        a, b = r["MCMCsamples"]["alpha"], ref["s_alpha"]
        assert np.array_equal(a != 0, b != 0)
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r["pi"], ref["pi"], rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"], r["mu"]], [ref["Vg"], ref["Ve"], ref["h2"], ref["mu"]], rtol=1e-9)


def test_bayesr_fold_in_any_order_is_the_chain_of_the_sorted_classes():
    """The reference takes `fold` in any order (src/Bayes.cpp:743-815) and picks the class by a cumulative walk over the classes
    as given (:773-781). The device needs the non-null classes by increasing variance (nested thresholds on q): hb_run sorts
    them — the run is the reference's chain for the sorted classes, draw for draw (the oracle on the sorted problem is the
    checker), and pi / MCMCsamples$pi come back in the caller's order. Same posterior: a mixture does not depend on how its
    components are numbered."""
    g = np.load(os.path.join(G, "small_all_models_philox.npz"))
    for Pi, fold in (([0.9375, 0.015625, 0.03125, 0.015625], [0, 1e-2, 1e-3, 1e-4]), ([0.875, 0.0625, 0.03125, 0.03125], [0, 1e-3, 1e-2, 1e-4])):
        order = [0] + sorted(range(1, len(fold)), key=lambda k: fold[k])          # internal class -> caller's class
        kw = dict(niter=16, nburn=6, thin=2, seed=424242)
        ref = O.bayes(g["y"], g["X"], "BayesR", [Pi[k] for k in order], fold=[fold[k] for k in order], rng=O.RNG_PHILOX, store_alpha=True, **kw)
        for panel in (64, 512):
            r = H.Bayes(g["y"], g["X"], "BayesR", Pi, fold=fold, verbose=False, panel=panel, **kw)
            a, b = r["MCMCsamples"]["alpha"], ref["s_alpha"]
            assert np.array_equal(a != 0, b != 0)
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(np.asarray(r["pi"])[order], ref["pi"], rtol=1e-9, atol=1e-14)
            np.testing.assert_allclose(np.asarray(r["MCMCsamples"]["pi"])[order, :], ref["s_pi"], rtol=1e-9, atol=1e-14)
            np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"], r["mu"]], [ref["Vg"], ref["Ve"], ref["h2"], ref["mu"]], rtol=1e-9)
            np.testing.assert_allclose(r["pip"], ref["pip"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("n,m", [(37, 5), (257, 65), (64, 513)])
def test_ragged_tiny_shapes_draw_for_draw_against_the_oracle(n, m):
    # fewer markers than a panel, row counts that are not a multiple of anything, a panel boundary at m - 1, an all-constant column
    rng = np.random.default_rng(n * 1000 + m)
    p = rng.uniform(0.1, 0.5, m)
    X = ((rng.random((n, m)) < p).astype(np.int8) + (rng.random((n, m)) < p).astype(np.int8))
    X[:, m // 2] = 1
    y = X[:, : min(m, 3)].astype(float) @ np.array([0.8, -0.5, 0.3])[: min(m, 3)] + rng.normal(0, 1, n)
    for model, Pi, fold in (("BayesCpi", [0.75, 0.25], None), ("BayesR", [0.5, 0.25, 0.125, 0.125], [0, 1e-3, 1e-2, 1e-1])):
        kw = dict(fold=fold, niter=12, nburn=4, thin=2, seed=7)
        ref = O.bayes(y, X, model, Pi, rng=O.RNG_PHILOX, store_alpha=True, **kw)
        r = H.Bayes(y, np.asfortranarray(X), model, Pi, verbose=False, precise=True, **kw)
        a, b = r["MCMCsamples"]["alpha"], ref["s_alpha"]
        assert np.array_equal(a != 0, b != 0)
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
        assert (a[m // 2] == 0).all()
        np.testing.assert_allclose([r["Vg"], r["Ve"], r["mu"]], [ref["Vg"], ref["Ve"], ref["mu"]], rtol=1e-9)


def test_serialising_environment_falls_back_to_the_per_panel_kernels():
    """The persistent pipeline needs kernels on two streams to be co-resident; where they are not (AMD_SERIALIZE_KERNEL,
    HIP_LAUNCH_BLOCKING, a counter-collecting profiler) the context must notice at creation and run the event-ordered
    per-panel kernels instead of stalling into its 3 s timeout — with the same results."""
    import subprocess
    import sys
    code = (
        "import numpy as np, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import hibayes_amd as H\n"
        "g = np.load(%r)\n"
        "with H.Context(g['X'].shape[0], g['X'].shape[1], precise=2) as c:\n"
        "    c.upload(g['X']); c.set_pipeline(1, 2, 7)\n"
        "    print('NOTE', c.pipeline_note()); print('PIPE', c.pipeline()[0])\n"
        "    r = H.Bayes(g['y'], None, 'BayesCpi', [0.95, 0.05], niter=16, nburn=6, thin=2, seed=424242, verbose=False, ctx=c)\n"
        "    print('ERR', float(np.abs(r['MCMCsamples']['alpha'] - g['BayesCpi_alpha']).max()))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(G, "small_all_models_philox.npz"))
    for var in ("HB_FORCE_SERIAL", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING"):
        env = dict(os.environ)
        env[var] = "3" if var == "AMD_SERIALIZE_KERNEL" else "1"
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
        assert var in lines["NOTE"] and lines["PIPE"] == "0", out.stdout
        assert float(lines["ERR"]) < 1e-9
        assert "per-panel kernels" in out.stderr
    # and without any of them the pipeline is on
    with H.Context(64, 64) as c:
        assert c.pipeline_note() is None and c.pipeline()[0] == 1


def test_in_library_rccl_exchange_world_1_is_the_plain_chain(demo):
    """The RCCL path on the one GPU this box has: a one-rank communicator still packs the residual delta, runs
    ncclAllReduce on the sweep stream and unpacks — the chain must be the unsharded one up to the rounding of
    yadj = yadj0 + (yadj - yadj0) (same decisions, effects to 1e-9)."""
    from hibayes_amd.dist import RcclComm
    comm = RcclComm(0, 1, 0)
    assert comm.L.hb_comm_world(comm.handle) == 1
    comm.selftest()           # hb_comm_selftest: one all-reduce with a known answer (what bench.py runs under its deadline)
    kw = dict(niter=60, nburn=20, thin=5, seed=99, verbose=False)
    a = H.Bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], comm=comm, **kw)
    b = H.Bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], **kw)
    assert np.array_equal(a["MCMCsamples"]["alpha"] != 0, b["MCMCsamples"]["alpha"] != 0)
    np.testing.assert_allclose(a["MCMCsamples"]["alpha"], b["MCMCsamples"]["alpha"], rtol=1e-9, atol=1e-13)
    assert np.array_equal(a["pip"], b["pip"])
    for k in ("mu", "Ve", "Vg", "h2"):
        assert a[k] == pytest.approx(b[k], rel=1e-9)
    np.testing.assert_allclose(a["e"], b["e"], rtol=0, atol=1e-9)
    # with windows and a second model through the same communicator
    wind = H.cutwind_by_num(np.array([int(c) for c in demo["plink"]["map"]["Chr"]]), demo["plink"]["map"]["Pos"], 50)
    a = H.Bayes(demo["y"], demo["M"], "BayesR", [0.95, 0.02, 0.02, 0.01], fold=[0, 1e-4, 1e-3, 1e-2], windindx=wind, comm=comm, **kw)
    b = H.Bayes(demo["y"], demo["M"], "BayesR", [0.95, 0.02, 0.02, 0.01], fold=[0, 1e-4, 1e-3, 1e-2], windindx=wind, **kw)
    np.testing.assert_allclose(a["alpha"], b["alpha"], rtol=1e-9, atol=1e-13)
    assert np.array_equal(a["gwas"], b["gwas"])
    comm.close()
