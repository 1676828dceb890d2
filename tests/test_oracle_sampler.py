"""The CPU oracle against every known answer the reference offers for this path
and against the committed golden vectors."""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_bed_decode_matches_readme_corner(demo):
    # reference README.md:78-86: dim 600 x 1000, geno[1:4, 1:5]
    raw = open(demo["prefix"] + ".bed", "rb").read()
    g = O.decode_bed(raw, 600, 1000)
    assert g.shape == (600, 1000)
    assert g[:4, :5].tolist() == [[2, 1, 1, 1, 0], [1, 0, 1, 1, 0], [0, 2, 0, 0, 0], [1, 1, 1, 1, 0]]
    assert (g >= 0).all() and (g <= 2).all()


def test_bed_decode_missing_and_imputation():
    # 5 individuals, 2 SNPs; SNP0 codes: 00 01 10 11 00 -> 2 NA 1 0 2 ; major genotype 2 fills the NA
    b0 = (0b00) | (0b01 << 2) | (0b10 << 4) | (0b11 << 6)
    raw = bytes([0x6C, 0x1B, 0x01, b0, 0b00, 0xFF, 0b11])
    g = O.decode_bed(raw, 5, 2, impute=False)
    assert g[:, 0].tolist() == [2, -128, 1, 0, 2]
    assert g[:, 1].tolist() == [0, 0, 0, 0, 0]
    g = O.decode_bed(raw, 5, 2, impute=True)
    assert g[:, 0].tolist() == [2, 2, 1, 0, 2]


def test_init_state_facts_on_demo(demo):
    # derived from the formulas at src/Bayes.cpp:310-363
    y, M = demo["y"], demo["M"]
    assert y.size == 300 and M.shape == (300, 1000)
    r = O.bayes(y, M, "BayesCpi", [0.95, 0.05], niter=2, nburn=0, thin=1)
    assert r["vary"] == pytest.approx(215.2144812894398, rel=1e-13)
    assert r["sumvx"] == pytest.approx(294.93311036789305, rel=1e-13)
    assert r["nvar0"] == 50
    assert r["xpx"][:5].tolist() == [483, 101, 209, 464, 65]
    assert np.flatnonzero(r["vx"] == 0)[:5].tolist() == [57, 80, 100, 149, 151]
    assert r["vara0"] == pytest.approx(53.80362032235995, rel=1e-13)
    assert r["s2vara"] == pytest.approx(26.901810161179974, rel=1e-13)
    assert r["varg0"] == pytest.approx(3.648530356950866, rel=1e-13)
    assert r["s2varg"] == pytest.approx(1.824265178475433, rel=1e-13)
    assert r["vare0"] == pytest.approx(107.6072406447199, rel=1e-13)
    assert r["lambda2_0"] == pytest.approx(589.8662207357861, rel=1e-13)
    assert r["rate0"] == pytest.approx(1.6952996541361921e-4, rel=1e-12)


def test_demo_golden_vectors(demo):
    g = np.load(os.path.join(G, "demo_bayescpi_philox.npz"))
    r = O.bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], niter=2000, nburn=1200, thin=5,
                rng=O.RNG_PHILOX, seed=666666, trace_iter=0)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"], r["mu"]], g["scal"], rtol=1e-10)
    np.testing.assert_allclose(r["alpha"], g["alpha"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(r["pip"], g["pip"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["trace_rhs"][:64], g["trace_rhs"], rtol=1e-11)
    assert np.array_equal(r["trace_cls"][:64], g["trace_cls"])
    # sanity band of reference README.md:159-167 (older version, other formula): finite, plausible
    assert 0.1 < r["h2"] < 0.7 and 0.5 < r["pi"][0] < 1 and r["pip"].max() < 1 and r["pip"].min() >= 0
    assert r["nzct"] == 800 and r["n_records"] == 160


def test_r_stream_and_philox_agree_statistically(demo):
    # two different generators, same sampler: posterior summaries agree within Monte-Carlo error
    a = [O.bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], niter=1500, nburn=500, thin=5, rng=O.RNG_R, seed=s)
         for s in (1, 2, 3)]
    b = [O.bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], niter=1500, nburn=500, thin=5, rng=O.RNG_PHILOX, seed=s)
         for s in (1, 2, 3)]
    for k in ("h2", "Ve", "mu"):
        ma, mb = np.mean([r[k] for r in a]), np.mean([r[k] for r in b])
        sd = np.std([r[k] for r in a + b]) + 1e-12
        assert abs(ma - mb) < 4 * sd / np.sqrt(3) + 0.05 * abs(ma)


@pytest.mark.parametrize("model,Pi,fold", [
    ("BayesCpi", [0.95, 0.05], None), ("BayesC", [0.9, 0.1], None), ("BayesRR", [0.95, 0.05], None),
    ("BayesA", [0.95, 0.05], None), ("BayesBpi", [0.95, 0.05], None), ("BayesB", [0.9, 0.1], None),
    ("BayesL", [0.95, 0.05], None), ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2])])
def test_small_case_golden_all_models(model, Pi, fold):
    g = np.load(os.path.join(G, "small_all_models_philox.npz"))
    r = O.bayes(g["y"], g["X"], model, Pi, fold=fold, niter=16, nburn=6, thin=2, rng=O.RNG_PHILOX, seed=424242,
                store_alpha=True)
    np.testing.assert_allclose(r["s_alpha"], g[model + "_alpha"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"], r["mu"]], g[model + "_scal"], rtol=1e-10)
    # invariants of the algorithm: monomorphic markers never move, fixed-pi models return Pi, PIP in [0,1)
    assert r["alpha"][3] == 0 and r["alpha"][130] == 0
    if model in ("BayesRR", "BayesA", "BayesL"):
        assert r["pi"].tolist() == [0, 1] and (r["pip"] == 1).all()
    elif model in ("BayesB", "BayesC"):
        assert r["pi"].tolist() == Pi
    else:
        assert abs(r["pi"].sum() - 1) < 1e-12
    assert (r["pip"] >= 0).all() and (r["pip"] <= 1).all()


def test_double_and_int8_layouts_give_identical_chains():
    g = np.load(os.path.join(G, "small_all_models_philox.npz"))
    a = O.bayes(g["y"], g["X"], "BayesCpi", [0.95, 0.05], niter=10, nburn=2, thin=2, store_alpha=True)
    b = O.bayes(g["y"], g["X"].astype(np.float64), "BayesCpi", [0.95, 0.05], niter=10, nburn=2, thin=2, store_alpha=True)
    np.testing.assert_allclose(a["s_alpha"], b["s_alpha"], rtol=1e-12, atol=1e-15)


def test_residual_identity_after_run():
    # e = y - mu - X alpha (src/Bayes.cpp:942, :971) and g = final-iteration u (:1023)
    g = np.load(os.path.join(G, "small_all_models_philox.npz"))
    r = O.bayes(g["y"], g["X"], "BayesCpi", [0.95, 0.05], niter=20, nburn=10, thin=2)
    e = g["y"] - r["mu"] - g["X"].astype(float) @ r["alpha"]
    np.testing.assert_allclose(r["e"], e, rtol=1e-10, atol=1e-10)


def test_validation_messages_are_the_references():
    g = np.load(os.path.join(G, "small_all_models_philox.npz"))
    X, y = g["X"], g["y"]
    cases = [
        (dict(Pi=[0.5, 0.6]), "sum of Pi should be 1."),
        (dict(Pi=[1.0, 0.0]), "all markers have no effect size."),
        (dict(Pi=[1.5, -0.5]), "elements of Pi should be at the range of [0, 1]"),
        (dict(model="BayesR", Pi=[0.9, 0.05, 0.05]), "'fold' should be provided for BayesR model."),
        (dict(model="BayesR", Pi=[0.9, 0.05, 0.05], fold=[0, 1]), "length of Pi and fold not equals."),
        (dict(Pi=[0.9, 0.05, 0.05], fold=[0, 1, 2]), "length of Pi should be 2, the first value is the proportion of non-effect markers."),
        (dict(Pi=[0.95, 0.05], dfvg=2.0), "dfvg should not be less than 2."),
        (dict(Pi=[0.95, 0.05], niter=5, nburn=10), "Number of total iteration ('niter') shold be larger than burn-in ('nburn')."),
    ]
    for kw, msg in cases:
        a = dict(model="BayesCpi", niter=4, nburn=2, thin=1)
        a.update(kw)
        with pytest.raises(RuntimeError) as ei:
            O.bayes(y, X, a.pop("model"), a.pop("Pi"), **a)
        assert str(ei.value) == msg
    yy = y.copy()
    yy[3] = np.nan
    with pytest.raises(RuntimeError, match="NAs are not allowed in y."):
        O.bayes(yy, X, "BayesCpi", [0.95, 0.05], niter=4, nburn=2, thin=1)
    # sum(Pi) != 1 is an exact compare (src/Bayes.cpp:101); the shipped BayesR default passes it
    O.bayes(y, X, "BayesR", [0.95, 0.02, 0.02, 0.01], fold=[0, 1e-4, 1e-3, 1e-2], niter=2, nburn=0, thin=1)


def test_full_formula_fixture(demo):
    g = np.load(os.path.join(G, "demo_full_formula_philox.npz"), allow_pickle=True)
    r = O.bayes(demo["y"], demo["M"], "BayesCpi", [0.98, 0.02], Cmat=g["C"], R=g["R"], niter=300, nburn=100, thin=5,
                rng=O.RNG_PHILOX, seed=666666)
    np.testing.assert_allclose(r["beta"], g["beta"], rtol=1e-9)
    np.testing.assert_allclose(r["Vr"], g["Vr"], rtol=1e-9)
    np.testing.assert_allclose(r["r"], g["r"], rtol=1e-8, atol=1e-10)
    assert r["n_levels"] == g["r"].size


def test_oracle_reproduces_the_fit_printed_in_the_reference_readme(demo):
    """THE pin of the sampler restatement. reference README.md:130-172 prints summary() of
        ibrm(T1 ~ season + bwt + (1 | loc) + (1 | dam), ..., method = "BayesCpi", Pi = c(0.98, 0.02),
             niter = 20000, nburn = 16000, thin = 5, seed = 666666)
    run by real hibayes under R. The oracle in R-stream mode (set.seed()'s Mersenne-Twister, inversion normals,
    Ahrens-Dieter rgamma / exp_rand: oracle/hbo_rng.c) consumes the same stream draw for draw, so after 20 000
    iterations of the whole loop (intercept, 4 covariates, 2 random-effect terms, 1000-marker BayesCpi sweep,
    variance and pi draws: src/Bayes.cpp:477-917) every printed digit of that summary must come out — a single
    misplaced draw or a mis-restated formula anywhere in the loop would decorrelate the chain completely."""
    g = np.load(os.path.join(G, "demo_full_formula_philox.npz"), allow_pickle=True)
    r = O.bayes(demo["y"], demo["M"], "BayesCpi", [0.98, 0.02], Cmat=g["C"], R=g["R"], niter=20000, nburn=16000,
                thin=5, rng=O.RNG_R, seed=666666, store_alpha=True)
    sd = lambda a, **k: np.std(a, ddof=1, **k)
    assert r["n_records"] == 800
    # Genetic random effects (README.md:162-166): Estimate, SD
    assert round(r["Vg"], 5) == 52.10097 and round(sd(r["s_Vg"]), 3) == 13.084
    assert round(r["h2"], 5) == 0.35748 and round(sd(r["s_h2"]), 3) == 0.081
    assert [round(v, 5) for v in r["pi"]] == [0.92683, 0.07317]
    assert [round(v, 3) for v in sd(r["s_pi"], axis=1)] == [0.039, 0.039]
    # Environmental random effects (:155-159): loc, dam, Residual
    assert [round(v, 2) for v in r["Vr"]] == [8.10, 54.29]
    assert [round(v, 3) for v in sd(r["s_Vr"], axis=1)] == [4.785, 10.096]
    assert round(r["Ve"], 2) == 30.78 or round(r["Ve"] - 5e-4, 2) == 30.77     # printed 30.77 (R prints 4 significant digits of 30.7753)
    assert round(sd(r["s_Ve"]), 3) == 6.323
    # Fixed effects (:147-153): (Intercept), seasonSpring, seasonSummer, seasonWinter, bwt
    assert round(r["mu"], 3) == 32.992 and round(sd(r["s_mu"]), 3) == 6.609
    assert [round(v, 3) for v in r["beta"]] == [-21.919, -11.484, -11.576, 2.399]
    assert [round(v, 3) for v in sd(r["s_beta"], axis=1)] == [1.437, 1.410, 1.549, 0.792]
    # Residuals (:143-145) and Marker effects (:169-171): R's quantile() type 7 == numpy's default, 5 significant digits
    sig = lambda v, d: [float("%.*g" % (d, x)) for x in v]                   # R prints summary() quantiles to d significant digits
    assert sig(np.quantile(r["e"], [0, .25, .5, .75, 1]), 5) == [-8.6113, -2.2907, 0.17169, 2.3326, 9.7695]
    assert sig(np.quantile(r["alpha"], [0, .25, .5, .75, 1]), 6) == [-1.98438, -0.0242465, 0.0, 0.0253073, 1.9202]
    assert len(r["r"]) == 50 + 150 and r["alpha"].size == 1000              # "group: loc, 50; dam, 150", "Number of markers: 1000"


def _arma_mean(y):
    # arma::mean's two interleaved accumulators (the start value of mu, src/Bayes.cpp:469)
    return (y[0::2].cumsum()[-1] + y[1::2].cumsum()[-1]) / y.size


WARM_MODELS = [("BayesCpi", [0.95, 0.05], None), ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2]), ("BayesL", [0.95, 0.05], None),
               ("BayesRR", [0.95, 0.05], None), ("BayesA", [0.95, 0.05], None), ("BayesB", [0.9, 0.1], None)]


@pytest.mark.parametrize("model,Pi,fold", WARM_MODELS)
def test_warm_state_equal_to_the_prior_defaults_is_the_cold_chain(demo, model, Pi, fold):
    """hbo_args.warm (the oracle's side of hb_warm_state) only replaces the START values of the scalars the loop carries
    (src/Bayes.cpp:319-374, :469); handing it exactly the cold run's start values must reproduce the cold chain bit for bit —
    which pins that nothing else (the prior constants s2varg_, rate0 ...) moved."""
    y, X = demo["y"], demo["M"][:, :300]
    kw = dict(fold=fold, niter=6, nburn=0, thin=1, seed=3, store_alpha=True)
    a = O.bayes(y, X, model, Pi, **kw)
    w = dict(mu=_arma_mean(y), vare=a["vare0"], varg=a["varg0"], pi=Pi, lambda2=a["lambda2_0"])
    b = O.bayes(y, X, model, Pi, warm=w, **kw)
    assert np.array_equal(a["s_alpha"], b["s_alpha"])
    for k in ("Vg", "Ve", "h2", "mu"):
        assert a[k] == b[k]


def test_a_continued_chain_starts_where_the_first_one_stopped(demo):
    """`last` (effects + scalars after the final iteration) handed back as g_init + warm: the continued run's first sweep sees the
    same markers in the model and the same pi / varg / vare — not the prior's pi = 0.95 that re-admits 5 % of the markers."""
    y, X = demo["y"], demo["M"]
    a = O.bayes(y, X, "BayesCpi", [0.95, 0.05], niter=400, nburn=399, thin=1, seed=11, store_alpha=True)
    lw = a["last"]["warm"]
    assert 0 < lw["pi"][0] < 1 and lw["vare"] > 0 and lw["varg"] > 0
    assert lw["vare"] == a["Ve"] and lw["mu"] == a["mu"]        # one stored record, the last iteration: its scalars ARE the last state
    assert lw["pi"][0] == a["pi"][0]
    assert np.array_equal(a["last"]["g"], a["s_alpha"][:, -1])
    kw = dict(niter=1, nburn=0, thin=1, seed=12, store_alpha=True, g_init=a["last"]["g"])
    cont = O.bayes(y, X, "BayesCpi", [0.95, 0.05], warm=lw, **kw)
    cold = O.bayes(y, X, "BayesCpi", [0.95, 0.05], **kw)
    # same effects, same draws, different hyper-parameters in the first sweep: the two differ, and the continued one is the one
    # whose first intercept draw started from the reported mu
    assert not np.array_equal(cont["s_alpha"], cold["s_alpha"])
    assert cont["last"]["warm"]["mu"] != lw["mu"]


def test_threaded_team_is_the_serial_sampler():
    """threads > 1: the BLAS-1 calls of n >= 16384 run on ONE persistent team of row-chunk workers (hb_oracle.c team_*), partial sums
    combined in a fixed two-level order — the same chain as one thread up to the summation order of a dot product."""
    rng = np.random.default_rng(4)
    n, m = 16400, 60
    p = rng.uniform(0.05, 0.5, m)
    X = np.asfortranarray(((rng.random((n, m)) < p).astype(np.float64) + (rng.random((n, m)) < p)))
    beta = np.zeros(m)
    beta[:6] = rng.normal(0, 1, 6)
    y = X @ beta + rng.normal(0, 1, n)
    kw = dict(niter=5, nburn=0, thin=1, seed=5, store_alpha=True)
    one = O.bayes(y, X, "BayesCpi", [0.9, 0.1], threads=1, **kw)
    for thr in (3, 11):          # 11 threads: two groups of the tree, more threads than this container has cores
        t = O.bayes(y, X, "BayesCpi", [0.9, 0.1], threads=thr, **kw)
        assert np.array_equal(t["s_alpha"] != 0, one["s_alpha"] != 0)
        np.testing.assert_allclose(t["s_alpha"], one["s_alpha"], rtol=1e-9, atol=1e-12)
        again = O.bayes(y, X, "BayesCpi", [0.9, 0.1], threads=thr, **kw)
        assert np.array_equal(again["s_alpha"], t["s_alpha"])   # deterministic for a given thread count
