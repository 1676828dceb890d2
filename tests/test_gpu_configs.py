"""BASELINE.json's configs at (or at the per-GPU shard shape of) their stated sizes, on one MI355X.

config 2  BayesCpi n=10k  m=100k            : the whole sampler, pipeline vs serial kernels + invariants
config 3  n=50k m=500k (BayesCpi and BayesR): pipeline (default geometry) vs serial kernels, precise 1 and 0,
          and draw-for-draw against the LIVE oracle run on the same 25 GB of genotypes (2 sweeps from cold)
config 4  BayesCpi n=50k  m=2M / 8 GPUs     : one shard (m=250k, global marker offset) pipeline vs serial + invariants
config 5  BayesB   n=200k m=1M / 8 GPUs     : one shard (m=125k) draw-for-draw against the live oracle
Sizes the oracle cannot finish in seconds are covered by the size-independent properties of the sweep:
yadj + u conserved, u = X g, class counts, monomorphic markers untouched. Reference: src/Bayes.cpp:586-823."""
import os

import numpy as np
import pytest

import hibayes_amd as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def host_free_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def synth_y(c, n, m, seed, m_offset=0, m_global=None):
    """y = X beta + e, h2 = 0.5, 0.1 % causal (SURVEY.md §8 d) from the device-resident genotypes."""
    rng = np.random.default_rng(seed)
    nc = max(1, m // 1000)
    beta = np.zeros(m)
    beta[rng.choice(m, nc, replace=False)] = rng.normal(0, 1, nc)
    xb = np.zeros(n)
    H._lib.check(c.L.hb_ctx_matvec(c.h, beta.ctypes.data, xb.ctypes.data))
    xb -= xb.mean()
    xb *= np.sqrt(0.5 / xb.var())
    return xb + rng.normal(0, np.sqrt(0.5), n)


def invariants(c, y, r):
    """What must hold after any number of sweeps, whatever moved."""
    n, m = c.n, c.m
    ra, u = c.get_residual()
    g, trk, _ = c.get_effects()
    xpx, vx, sumvx, nvar0 = c.marker_stats()
    xg = np.zeros(n)
    H._lib.check(c.L.hb_ctx_matvec(c.h, g.ctypes.data, xg.ctypes.data))
    np.testing.assert_allclose(u, xg, rtol=0, atol=1e-8 * max(1.0, np.abs(xg).max()))        # u = X g
    mu_last = r["MCMCsamples"]["mu"][0, -1]
    np.testing.assert_allclose(ra + u, y - mu_last, rtol=0, atol=1e-8 * np.abs(y).max())     # yadj = y - mu - X g
    assert not g[vx == 0].any() and not trk[vx == 0].any()
    assert np.array_equal(trk != 0, g != 0)
    np.testing.assert_allclose(u, r["g"], rtol=0, atol=0)                                     # results["g"] = final u (:1023)


def run_both_geometries(c, y, model, Pi, fold, geo, niter, precise, nburn=0, thin=1, seed=31337):
    out = []
    for g in (geo, (0, 0, 1)):
        c.set_pipeline(*g)
        assert c.pipeline()[:3] == g
        r = H.Bayes(y, None, model, Pi, fold=fold, niter=niter, nburn=nburn, thin=thin, seed=seed, verbose=False,
                    precise=precise, ctx=c, store_alpha=(c.m <= 300000))
        out.append(r)
        if g == geo:
            invariants(c, y, r)
    return out


def same_chain(a, b, tol, what):
    np.testing.assert_allclose(a["alpha"], b["alpha"], rtol=tol, atol=1e-13, err_msg=what)
    assert np.array_equal(a["pip"], b["pip"]), what
    for k in ("Vg", "Ve", "h2", "mu"):
        assert a[k] == pytest.approx(b[k], rel=tol), (what, k)
    np.testing.assert_allclose(a["pi"], b["pi"], rtol=tol, atol=1e-14)
    np.testing.assert_allclose(a["e"], b["e"], rtol=0, atol=1e-7)


def test_config2_bayescpi_n10k_m100k_pipeline_vs_serial():
    n, m = 10000, 100000
    with H.Context(n, m, seed=2) as c:
        c.generate(20240901, mono_every=1000)
        y = synth_y(c, n, m, 11)
        a, b = run_both_geometries(c, y, "BayesCpi", [0.95, 0.05], None, (1, 2, 7), niter=50, precise=2, nburn=10, thin=2)
        same_chain(a, b, 1e-9, "config 2: pipeline (1,2,7) vs serial kernels")
        assert a["timing"]["mean_events"] > 100


def test_config2_bayescpi_n10k_m100k_against_live_oracle():
    """BASELINE.json configs[1] at its own size against the LIVE oracle (rounds 1-4 compared the pipeline with the per-panel kernels
    there, i.e. HIP with HIP): 12 sweeps from cold on the library's default geometry, 2-bit genotypes, geometry by regime."""
    _full_size_vs_oracle(10000, 100000, "BayesCpi", [0.95, 0.05], None, (1, 3, 7), 0, niter=12, bits=2, adaptive=True)


def test_config4_shard_bayescpi_n50k_m250k_offset_750k_against_live_oracle():
    """BASELINE.json configs[3], rank 3 of 8: global markers [750 000, 1 000 000) of m_global = 2M — the per-marker Philox streams are
    addressed by the GLOBAL marker index (hb_ctx_params.m_offset / hbo_args.marker_offset), which only this case exercises at size
    against the oracle. 2 sweeps from cold, int8 columns, the library's default geometry for them."""
    _full_size_vs_oracle(50000, 250000, "BayesCpi", [0.95, 0.05], None, (1, 2, 7), 750000, niter=2)


def test_config4_shard_shape_bayescpi_n50k_m250k_pipeline_vs_serial():
    # rank 3 of 8 of config 4: global markers [750000, 1000000) of m_global = 2M (RNG addressed by global index)
    n, m = 50000, 250000
    with H.Context(n, m, seed=4, m_offset=750000) as c:
        c.generate(20240901, mono_every=1000)
        y = synth_y(c, n, m, 13)
        a, b = run_both_geometries(c, y, "BayesCpi", [0.95, 0.05], None, (1, 2, 7), niter=6, precise=2)
        same_chain(a, b, 1e-9, "config 4 shard: pipeline vs serial")


@pytest.fixture(scope="module")
def c3():
    """Config 3's genotypes (n = 50k, m = 500k, generated on the device) in ONE context shared by the full-size oracle cases, with
    the int8 matrix downloaded once for the live oracle (25 GB): generating, downloading and re-checking them per case was most of
    the suite's run time."""
    n, m = 50000, 500000
    need = n * m / 1e9 + 8
    if host_free_gb() < need:
        pytest.skip("host has %.0f GB available, the live oracle needs %.0f GB for the int8 genotypes" % (host_free_gb(), need))
    c = H.Context(n, m, seed=20240901, m_offset=0, precise=2)
    c.generate(20240901, mono_every=1000)
    y = synth_y(c, n, m, 17)
    X = c.download()
    yield c, y, X
    c.close()


ORACLE_THREADS = int(os.environ.get("HB_TEST_ORACLE_THREADS", str(max(1, min(8, len(os.sched_getaffinity(0)))))))


def _full_size_vs_oracle(n, m, model, Pi, fold, geo, m_offset, niter=2, precise=2, dense=0.0, shared=None, bits=8, adaptive=False, stationary=0, installed_pi=None):
    """dense > 0: the chain starts from an installed state with that fraction of the markers in the model (g_init on both
    sides): crowded rounds, row-cache misses and band folds of hundreds of moves per mat-vec group from the first panel on.
    bits = 2: the sweep runs on the 2-bit resident layout with the int8 copy dropped after the Gram build, as bench.py's headline does.
    stationary = S > 0: the GPU first runs S sweeps on its own (no oracle: hundreds of sweeps at this size are minutes of CPU), then
    BOTH sides start from the state it reports — effects, mu, vare, varg, pi (hb_bayes_out.last -> g_init + hb_warm_state / hbo_warm) —
    and run `niter` sweeps under a new seed: draw-for-draw parity IN the regime a long run lives in and bench.py times (few markers in
    the model, the wide geometry, hot list and row cache shaped by history), not from a synthetic start."""
    kw = dict(fold=fold, niter=niter, nburn=0, thin=1, seed=20240901)
    if shared is None:
        need = n * m / 1e9 + 8
        if host_free_gb() < need:
            pytest.skip("host has %.0f GB available, the live oracle needs %.0f GB for the int8 genotypes" % (host_free_gb(), need))
        c = H.Context(n, m, seed=20240901, m_offset=m_offset, precise=precise)
        c.generate(20240901, mono_every=1000)
        y = synth_y(c, n, m, 17)
        X = None
    else:
        c, y, X = shared
    try:
        c.set_pipeline(*geo)
        if bits == 2:
            c.build_gram()
            c.set_layout(2, keep_int8=False)
            assert c.layout() == (2, False)
        if adaptive:
            c.set_adaptive(True)
        if dense > 0:
            rs = np.random.default_rng(23)
            g0 = np.zeros(m)
            on = rs.choice(m, int(dense * m), replace=False)
            g0[on] = rs.normal(0, 0.01, on.size)
            kw["g_init"] = g0
            if installed_pi is not None:   # ... and the hyper-parameters of a chain that has found the signal (hb_warm_state on both sides)
                kw["warm"] = dict(mu=float(np.mean(y)), vare=float(0.6 * np.var(y)), varg=5e-5, pi=list(installed_pi))
        if stationary > 0:
            pre = H.Bayes(y, None, model, Pi, verbose=False, precise=precise, ctx=c, store_alpha=False,
                          **dict(kw, niter=stationary, nburn=stationary - 1, seed=777))
            kw["g_init"], kw["warm"] = pre["last"]["g"], pre["last"]["warm"]
            nnz_pre = int((pre["last"]["g"] != 0).sum())
            print("%s n=%d m=%d: state after %d GPU sweeps: %d markers in the model, pi %s, varg %.3g, vare %.4f, %.0f moves per sweep"
                  % (model, n, m, stationary, nnz_pre, np.round(pre["last"]["warm"]["pi"], 5), pre["last"]["warm"]["varg"],
                     pre["last"]["warm"]["vare"], pre["timing"]["mean_events"]))
        r = H.Bayes(y, None, model, Pi, verbose=False, precise=precise, ctx=c, store_alpha=False, **kw)
        assert (adaptive or c.pipeline()[:3] == geo) and c.layout()[0] == bits
        if adaptive:
            c.set_adaptive(False)
        invariants(c, y, r)
        g_gpu, trk, _ = c.get_effects()
        if X is None:
            X = c.download()
    finally:
        if shared is None:
            c.close()
        elif bits == 2:
            c.set_layout(8)          # (the next case finds the int8 columns again)
    # (round 6: the live oracle on its persistent team of row-chunk workers — int8 columns too since this round —: the same sampler, its dot
    # products summed in eight chunks; 25 s of every at-size case were two one-thread sweeps over 25 GB)
    ref = O.bayes(y, X, model, Pi, rng=O.RNG_PHILOX, store_alpha=True, marker_offset=m_offset, threads=ORACLE_THREADS, **kw)
    g_ref = ref["s_alpha"][:, -1]
    assert np.array_equal(g_gpu != 0, g_ref != 0), "%d of %d inclusion decisions differ" % (((g_gpu != 0) != (g_ref != 0)).sum(), m)
    np.testing.assert_allclose(g_gpu, g_ref, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(r["alpha"], ref["alpha"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"], r["mu"]], [ref["Vg"], ref["Ve"], ref["h2"], ref["mu"]], rtol=1e-8)
    np.testing.assert_allclose(r["pi"], ref["pi"], rtol=1e-8)
    np.testing.assert_allclose(r["g"], ref["g"], rtol=1e-7, atol=1e-8)
    assert (g_ref != 0).sum() > 0


# (1, 2, 7) on 2-bit genotypes with the int8 copy dropped is THE shape bench.py's `value` is measured on since the end of round 6 (ld2 = 98 x 128 bytes, a
# ragged last stage; rounds 4-5 and most of 6: (1, 3, 7), still selectable with HB_WIDE_LV=3 and still tested here); (1, 2, 7) on int8 columns: the
# layout north_star names ((1, 3, 7) on int8 columns is covered at n = 32 768 in test_gpu_depth.py)
@pytest.mark.parametrize("geo,bits", [((1, 2, 7), 2), ((1, 3, 7), 2), ((1, 2, 7), 8), ((1, 2, 8), 2)])
def test_config3_bayescpi_n50k_m500k_draw_for_draw_against_live_oracle(c3, geo, bits):
    _full_size_vs_oracle(50000, 500000, "BayesCpi", [0.95, 0.05], None, geo, 0, shared=c3, bits=bits)


@pytest.mark.parametrize("geo", [(1, 2, 1), (1, 2, 2)])
def test_config3_bayesr_n50k_m500k_draw_for_draw_against_live_oracle(c3, geo):
    """BASELINE.json configs[2] is BayesR at n=50k, m=500k: its own model, at its own size, against the live oracle (reference
    src/Bayes.cpp:743-815) — 2 sweeps from cold, ~47 moves per panel: in the geometry a run holds in that regime ((2, 1), k_chain_persist)
    and, round 6, on the group chain it switches to once few markers move ((2, 2): here its crowded path — rounds of 64 candidates,
    the full fold and the exact check)."""
    _full_size_vs_oracle(50000, 500000, "BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], geo, 0, shared=c3)


@pytest.mark.parametrize("model,Pi,fold,geo,bits,sweeps", [("BayesCpi", [0.95, 0.05], None, (1, 2, 7), 2, 300),
                                                           ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], (1, 2, 1), 8, 150),
                                                           ])
def test_config3_stationary_state_against_live_oracle(c3, model, Pi, fold, geo, bits, sweeps):
    """n = 50k, m = 500k IN the regime `value` is measured in: 300 sweeps of burn-in on the GPU (bench.py's --burnin until the last run of round 5, 400 since; BayesR 150),
    then 2 sweeps on both sides from the reported state."""
    _full_size_vs_oracle(50000, 500000, model, Pi, fold, geo, 0, niter=2, shared=c3, bits=bits, adaptive=(bits == 2), stationary=sweeps)


def test_config3_bayesr_sparse_state_on_the_group_chain_against_live_oracle(c3):
    """Round 6: BayesR at n = 50k, m = 500k in the regime its converged leg of bench.py runs in — 0.8 % of the markers in the model and
    pi0 = 0.992 installed on both sides (these phenotypes keep 30 000 markers in the model for a thousand sweeps; bench.py's find the signal after
    700) — geometry (2, 2): the certified group chain k_chain_group<3, 2, 4, 10>, ~16 moves per two-panel group, 2 sweeps against the live oracle."""
    _full_size_vs_oracle(50000, 500000, "BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], (1, 2, 2), 0, niter=2, shared=c3, dense=0.008,
                         installed_pi=[0.992, 0.004, 0.003, 0.001])


@pytest.mark.parametrize("model,Pi,fold,geo", [("BayesCpi", [0.95, 0.05], None, (1, 3, 7)),
                                                ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], (1, 2, 1))])
def test_config3_dense_installed_state_against_live_oracle(c3, model, Pi, fold, geo):
    """The same size from an installed state with 5 % of the markers in the model (25 000 certain movers in the first sweep)."""
    _full_size_vs_oracle(50000, 500000, model, Pi, fold, geo, 0, niter=2, dense=0.05, shared=c3)


def test_config5_shard_shape_bayesb_n200k_m125k_draw_for_draw_against_live_oracle():
    # rank 5 of 8 of config 5 (BayesB, n = 200k, m_global = 1M)
    _full_size_vs_oracle(200000, 125000, "BayesB", [0.95, 0.05], None, (1, 3, 7), 625000)


@pytest.mark.parametrize("model,Pi,fold,geo", [("BayesCpi", [0.95, 0.05], None, (1, 2, 7)),
                                                ("BayesR", [0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2], (1, 2, 1))])
def test_config3_pipeline_vs_serial_kernels_precise_and_fast(model, Pi, fold, geo):
    """n=50k, m=500k, 3 sweeps from cold. fp64 mat-vec: the pipeline and the serial per-panel kernels must give the same
    chain (identical decisions and move lists; effects and residual to 1e-9 — the two differ only in the order the band
    corrections are added to a right-hand side). fp32 mat-vec image (precise=0): decisions may flip where q sits within
    ~1e-6 of its threshold; the flip rate is measured and bounded."""
    n, m = 50000, 500000
    res = {}
    y = None
    for precise in (2, 1, 0):
        with H.Context(n, m, seed=99, precise=precise) as c:
            c.generate(20240901, mono_every=1000)
            if y is None:
                y = synth_y(c, n, m, 19)
            xpx, vx, sumvx, nvar0 = c.marker_stats()
            vare, varg = 0.5, 0.5 / (0.05 * sumvx)
            logpi = np.log(Pi)
            fo = fold if fold is not None else [0, 0]
            for g in (geo, (0, 0, 1)):
                c.set_pipeline(*g)
                c.set_effects(np.zeros(m), np.zeros(m, dtype=np.uint8))
                c.set_residual(y - y.mean(), np.zeros(n))
                evs = []
                for it in range(3):
                    s = c.sweep(model, it, vare, varg, logpi=logpi, fold=fo)
                    assert s["class_count"].sum() == m - nvar0
                    cnt, lists = c.events()
                    evs.append((cnt, np.concatenate([l[0] for l in lists]), np.concatenate([l[1] for l in lists])))
                gg, trk, _ = c.get_effects()
                r, u = c.get_residual()
                res[(precise, g == geo)] = (gg, trk, r, u, evs)
    # exact fixed point (2) and fp64 FMA (1): pipeline == serial kernels, and the two arithmetics give the same chain
    for pr in (2, 1):
        (g1, t1, r1, u1, e1), (g0, t0, r0, u0, e0) = res[(pr, True)], res[(pr, False)]
        assert np.array_equal(t1, t0)
        for (c1, i1, d1), (c0, i0, d0) in zip(e1, e0):
            assert np.array_equal(c1, c0) and np.array_equal(i1, i0)       # same markers moved, in the same order
            np.testing.assert_allclose(d1, d0, rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(g1, g0, rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(r1, r0, rtol=0, atol=1e-10)
        np.testing.assert_allclose(u1, u0, rtol=0, atol=1e-10)
    assert np.array_equal(res[(2, True)][1], res[(1, True)][1])
    np.testing.assert_allclose(res[(2, True)][0], res[(1, True)][0], rtol=1e-9, atol=1e-14)
    g1, t1 = res[(2, True)][0], res[(2, True)][1]
    # fp32 image: pipeline vs serial, and fast vs exact
    (f1, ft1, fr1, fu1, _), (f0, ft0, _, _, _) = res[(0, True)], res[(0, False)]
    decisions = 3.0 * (m - nvar0)
    flips_geo = int((ft1 != ft0).sum())
    flips_prec = int((ft1 != t1).sum())
    print("%s n=50k m=500k: class differences after 3 sweeps — fp32 pipeline vs fp32 serial %d, fp32 vs fp64 %d (of %.0f decisions); "
          "max |g32 - g64| = %.3g" % (model, flips_geo, flips_prec, decisions, np.abs(f1 - g1).max()))
    assert flips_geo <= 1e-4 * decisions and flips_prec <= 1e-4 * decisions
    same = ft1 == t1
    np.testing.assert_allclose(f1[same], g1[same], rtol=0, atol=2e-3 * max(1e-12, np.abs(g1).max()))
    np.testing.assert_allclose(fr1 + fu1, y - y.mean(), rtol=0, atol=1e-9)  # the f64 master residual stays exact in both modes


@pytest.mark.parametrize("model,geo", [("BayesRR", (1, 2, 2)), ("BayesA", (1, 2, 1)), ("BayesL", (1, 2, 2))])
def test_all_move_models_at_n50k_against_live_oracle(model, geo):
    """BayesRR / A / L with the BASELINE's 50 000 individuals (src/Bayes.cpp:587-625, :719-741): 784 update blocks per launch —
    one per 64 rows, more than the chip has compute units, the shape under which a late chain workgroup never found a free one
    before k_gate —, the slab of every block through LDS-DMA, k_chain_dense and k_fold_dense over 100 panels. Three sweeps from
    cold against the oracle on the downloaded genotypes: every effect to 1e-8 (BayesL 1e-6: see test_gpu_depth)."""
    n, m = 50000, 51200
    kw = dict(niter=3, nburn=0, thin=1, seed=20240901)
    with H.Context(n, m, seed=20240901, panel=512) as c:
        c.generate(20240901, mono_every=1000)
        y = synth_y(c, n, m, 17)
        c.set_pipeline(*geo)
        r = H.Bayes(y, None, model, [0.95, 0.05], verbose=False, ctx=c, store_alpha=False, **kw)
        invariants(c, y, r)
        g_gpu, trk, _ = c.get_effects()
        X = c.download()
    ref = O.bayes(y, X, model, [0.95, 0.05], rng=O.RNG_PHILOX, store_alpha=True, **kw)
    g_ref = ref["s_alpha"][:, -1]
    tol = 1e-6 if model == "BayesL" else 1e-8
    assert np.array_equal(g_gpu != 0, g_ref != 0)
    np.testing.assert_allclose(g_gpu, g_ref, rtol=tol, atol=1e-12)
    np.testing.assert_allclose([r["Vg"], r["Ve"], r["h2"], r["mu"]], [ref["Vg"], ref["Ve"], ref["h2"], ref["mu"]], rtol=tol)
    np.testing.assert_allclose(r["g"], ref["g"], rtol=1e-7, atol=1e-8)
    assert r["timing"]["mean_events"] > 0.99 * m
