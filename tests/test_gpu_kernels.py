"""Kernel-level parity on the MI355X, through the C ABI (hb_ctx_*): integer-exact pieces must match
bit for bit, floating-point pieces within the stated tolerance."""
import os

import numpy as np
import pytest

import hibayes_amd as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def rand_geno(rng, n, m, signed=False):
    p = rng.uniform(0.05, 0.5, m)
    X = (rng.random((n, m)) < p).astype(np.int8) + (rng.random((n, m)) < p).astype(np.int8)
    if signed:
        X = X - 1
    return np.asfortranarray(X.astype(np.int8))


@pytest.mark.parametrize("n,m,panel", [(300, 1000, 0), (1000, 777, 64), (4097, 130, 128), (257, 2049, 512)])
def test_marker_stats_exact(n, m, panel):
    rng = np.random.default_rng(n + m)
    X = rand_geno(rng, n, m)
    X[:, 5] = 1
    X[:, m - 1] = 0
    with H.Context(n, m, panel=panel) as c:
        c.upload(X)
        assert np.array_equal(c.download(), X)
        xpx, vx, sumvx, nvar0 = c.marker_stats()
    Xd = X.astype(np.float64)
    assert np.array_equal(xpx, (Xd ** 2).sum(0))              # integer-exact
    np.testing.assert_allclose(vx, Xd.var(0, ddof=1), rtol=1e-13, atol=0)
    assert vx[5] == 0 and vx[m - 1] == 0 and nvar0 == int((Xd.var(0) == 0).sum())
    assert sumvx == pytest.approx(vx.sum(), rel=1e-13)


def test_marker_stats_on_demo_match_reference_facts(demo):
    with H.Context(300, 1000) as c:
        c.upload(demo["M"])
        xpx, vx, sumvx, nvar0 = c.marker_stats()
    assert xpx[:5].tolist() == [483, 101, 209, 464, 65] and nvar0 == 50
    assert sumvx == pytest.approx(294.93311036789305, rel=1e-13)


@pytest.mark.parametrize("panel", [64, 128, 256, 512])
def test_gram_blocks_exact(panel):
    rng = np.random.default_rng(panel)
    n, m = 1111, 3 * panel + 17
    X = rand_geno(rng, n, m, signed=(panel == 128))
    with H.Context(n, m, panel=panel) as c:
        c.upload(X)
        c.build_gram()
        for p in range((m + panel - 1) // panel):
            cols = X[:, p * panel:(p + 1) * panel].astype(np.int64)
            G = c.gram(p)
            k = cols.shape[1]
            assert np.array_equal(G[:k, :k], cols.T @ cols)      # int32, exact; asymmetric off-diagonal tiles
            assert not G[k:, :].any() and not G[:, k:].any()      # zero padding


@pytest.mark.parametrize("signed", [False, True])
def test_panel_matvec_against_fp64(signed):
    rng = np.random.default_rng(3)
    n, m = 5000, 1300
    X = rand_geno(rng, n, m, signed=signed)
    r = rng.normal(0, 2.0, n)
    ref = X.astype(np.float64).T @ r
    scale = np.sqrt((X.astype(np.float64) ** 2).sum(0) * r.var()) + 1e-9   # natural scale of x_j . r
    with H.Context(n, m, precise=True) as c:
        c.upload(X)
        c.set_residual(r, np.zeros(n))
        d = c.dot()
    assert np.max(np.abs(d - ref) / scale) < 1e-13                      # fp64 accumulation
    with H.Context(n, m, precise=2) as c:                               # exact fixed-point digits: error = quantisation of r only
        c.upload(X)
        c.set_residual(r, np.zeros(n))
        dq = c.dot()
    assert np.max(np.abs(dq - ref) / scale) < 1e-13
    # ... and it is exactly the dot product with the quantised residual: q = rint(r * 2^E), 2^E * max|r| < 2^54
    E = 53 - int(np.floor(np.log2(np.abs(r).max())))
    q = np.rint(np.ldexp(r, E))
    exact = np.array([int(v) for v in (X.astype(object).T @ q.astype(np.int64).astype(object))], dtype=object)
    want = np.array([float(v) * 2.0 ** -E for v in exact])
    assert np.array_equal(dq, want)                                     # bit for bit, whatever the launch geometry
    with H.Context(n, m, precise=False) as c:
        c.upload(X)
        c.set_residual(r, np.zeros(n))
        d = c.dot()
    assert np.max(np.abs(d - ref) / scale) < 2e-6                       # fp32 accumulation: stated tolerance


def test_matvec_and_residual_helpers():
    rng = np.random.default_rng(4)
    n, m = 2000, 900
    X = rand_geno(rng, n, m)
    a = np.zeros(m)
    a[rng.choice(m, 40, replace=False)] = rng.normal(size=40)
    out = np.zeros(n)
    with H.Context(n, m) as c:
        c.upload(X)
        H._lib.check(c.L.hb_ctx_matvec(c.h, a.ctypes.data, out.ctypes.data))
        np.testing.assert_allclose(out, X.astype(float) @ a, rtol=1e-12, atol=1e-12)
        r = rng.normal(size=n)
        c.set_residual(r, np.zeros(n))
        s1, s2 = c.residual_sums()
        assert s1 == pytest.approx(r.sum(), abs=1e-9) and s2 == pytest.approx((r * r).sum(), rel=1e-12)
        c.residual_shift(0.25)
        Cm = rng.normal(size=(n, 2))
        c.set_covariates(Cm)
        assert c.cov_dot(1) == pytest.approx(Cm[:, 1] @ (r + 0.25), rel=1e-12)
        c.cov_axpy(0, -0.5)
        zid = rng.integers(0, 7, size=(n, 1))
        c.set_levels(zid, [7])
        cur = r + 0.25 - 0.5 * Cm[:, 0]
        np.testing.assert_allclose(c.level_sums(0), np.bincount(zid[:, 0], weights=cur, minlength=7), rtol=1e-11)
        delta = rng.normal(size=7)
        c.level_axpy(0, delta)
        got, _ = c.get_residual()
        np.testing.assert_allclose(got, cur + delta[zid[:, 0]], rtol=1e-13, atol=1e-13)


def test_covariate_and_random_effect_blocks_on_the_device():
    """hb_ctx_blocks_step (reference src/Bayes.cpp:484-516 with the deviates passed in) against the same arithmetic in
    numpy: two iterations, two covariates, two random terms; tolerance 1e-11 relative (sums in a different order)."""
    rng = np.random.default_rng(11)
    n, m = 3000, 128
    X = rand_geno(rng, n, m)
    r = rng.normal(size=n)
    Cm = rng.normal(size=(n, 2))
    zid = np.stack([rng.integers(0, 9, size=n), rng.integers(0, 150, size=n)], axis=1)
    nlev = [9, 150]
    first = [0, 9]
    zz = np.concatenate([np.bincount(zid[:, t], minlength=nlev[t]) for t in range(2)]).astype(float)
    cpc = (Cm * Cm).sum(0)
    vrtmp = np.array([0.3, 0.7])
    dfr, s2r, vare = -1.0, 0.0, 1.3
    beta = np.zeros(2)
    estR = np.zeros(159)
    vr = np.zeros(2)
    with H.Context(n, m) as c:
        c.upload(X)
        c.set_residual(r, np.zeros(n))
        c.set_covariates(Cm)
        c.set_levels(zid, nlev)
        c.blocks_setup(cpc, zz, vrtmp)
        cur = r.copy()
        for it in range(2):
            zb, zl = rng.normal(size=2), rng.normal(size=159)
            ch = rng.chisquare([nlev[0] + dfr, nlev[1] + dfr])
            c.blocks_step(vare, zb, zl, ch, dfr, s2r)
            for i in range(2):
                rhs = Cm[:, i] @ cur + cpc[i] * beta[i]
                gi = rhs / cpc[i] + np.sqrt(vare / cpc[i]) * zb[i]
                cur += (beta[i] - gi) * Cm[:, i]
                beta[i] = gi
            for t in range(2):
                sl = slice(first[t], first[t] + nlev[t])
                rhs = np.bincount(zid[:, t], weights=cur, minlength=nlev[t]) + zz[sl] * estR[sl]
                l = zz[sl] + vare / vrtmp[t]
                en = rhs / l + np.sqrt(vare / l) * zl[sl]
                cur += (estR[sl] - en)[zid[:, t]]
                estR[sl] = en
                vrtmp[t] = (en @ en + s2r * dfr) / ch[t]
                vr[t] = en.var(ddof=1)
            vare = 0.9
        b, e, vt, v = c.blocks_state()
        got, _ = c.get_residual()
    np.testing.assert_allclose(b, beta, rtol=1e-11)
    np.testing.assert_allclose(e, estR, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(vt, vrtmp, rtol=1e-11)
    np.testing.assert_allclose(v, vr, rtol=1e-11)
    np.testing.assert_allclose(got, cur, rtol=1e-10, atol=1e-11)


def test_bed_decode_on_device_matches_golden(demo):
    raw = open(demo["prefix"] + ".bed", "rb").read()
    with H.Context(300, 1000) as c:
        c.upload_bed(raw, 600, rows=demo["rows"])
        assert np.array_equal(c.download(), demo["M"])
    with H.Context(600, 1000) as c:
        c.upload_bed(raw, 600)
        g = c.download()
    assert g[:4, :5].tolist() == [[2, 1, 1, 1, 0], [1, 0, 1, 1, 0], [0, 2, 0, 0, 0], [1, 1, 1, 1, 0]]  # README.md:81-86
    # ragged individuals + missing calls + imputation from the counts over ALL individuals
    rng = np.random.default_rng(8)
    nind, nsnp = 1003, 60
    body = rng.integers(0, 256, size=nsnp * ((nind + 3) // 4), dtype=np.uint8)
    img = bytes([0x6C, 0x1B, 0x01]) + body.tobytes()
    ref = O.decode_bed(img, nind, nsnp, impute=True)
    rows = rng.permutation(nind)[:400]
    with H.Context(400, nsnp) as c:
        c.upload_bed(img, nind, rows=rows)
        assert np.array_equal(c.download(), ref[rows, :])
    with pytest.raises(H.HibayesError):
        with H.Context(400, nsnp) as c:
            c.upload_bed(b"\x00\x01\x02" + body.tobytes(), nind, rows=rows)


def test_f64_upload_checks_integrality():
    n, m = 64, 70
    X = np.random.default_rng(1).integers(0, 3, size=(n, m)).astype(np.float64)
    with H.Context(n, m) as c:
        c.upload(X)
        assert np.array_equal(c.download(), X.astype(np.int8))
        X[5, 9] = 0.5   # an imputed fractional genotype (ssbrm) must be refused, never rounded
        with pytest.raises(H.HibayesError) as ei:
            c.upload(X)
        assert ei.value.status == 4


def test_synthetic_generator_is_the_documented_philox_stream():
    n, m, seed = 101, 9, 20240901
    with H.Context(n, m, m_offset=1000) as c:
        c.generate(seed, mono_every=4)
        X = c.download()
    for j in range(m):
        gj = 1000 + j
        sub = (3 << 56) | gj
        pj = 0.05 + 0.45 * O.lib().hbo_philox_uniform(seed, sub, 0xFFFFFFFFFF)
        thr = int(pj * 65536.0)
        for i in range(n):
            w = int(O.philox_block(seed, sub, i // 4)[i % 4])
            x = int((w & 0xFFFF) < thr) + int((w >> 16) < thr)
            if gj % 4 == 3:
                x = 0
            assert X[i, j] == x


def test_gebv_sample_matrix_on_device():
    # MCMCsamples$g = M %*% MCMCsamples$alpha (reference R/bayes.r:303-305): int8 genotypes x fp64 effect samples
    rng = np.random.default_rng(12)
    n, m, R = 1537, 700, 19
    X = rand_geno(rng, n, m, signed=True)
    A = np.zeros((m, R))
    for r in range(R):                                   # point-mass samples: a few markers in the model per record
        j = rng.choice(m, 12, replace=False)
        A[j, r] = rng.normal(0, 0.3, 12)
    A[:, 7] = rng.normal(0, 0.01, m)                     # one dense record (BayesRR-like)
    A[:, 8:16] = 0.0                                     # a whole block of eight empty records
    with H.Context(n, m) as c:
        c.upload(X)
        G = c.matmul(A)
        g1 = c.matmul(A[:, 3])
    ref = X.astype(np.float64) @ A
    np.testing.assert_allclose(G, ref, rtol=1e-13, atol=1e-13)
    assert not G[:, 8:16].any()
    np.testing.assert_allclose(g1[:, 0], ref[:, 3], rtol=1e-13, atol=1e-13)


def test_bigmemory_file_to_device_without_a_host_copy(tmp_path, demo):
    # the .bin a previous read_plink(out=) session left behind, memory-mapped and uploaded as it is (int8, column-major)
    out = str(tmp_path / "bm")
    H.write_bigmatrix(out, demo["plink"]["geno"])
    g = H.attach_bigmatrix(out + ".desc")
    assert isinstance(g, np.memmap) and g.dtype == np.int8
    with H.Context(600, 1000) as c:
        c.upload(g)
        assert np.array_equal(c.download(), demo["plink"]["geno"])
        xpx, vx, sumvx, nvar0 = c.marker_stats()
    assert xpx[0] == float((demo["plink"]["geno"][:, 0].astype(np.int64) ** 2).sum())
    sub = g[np.asarray(demo["rows"]), :]                 # ibrm()'s row subset (R/bayes.r:286-291)
    r = H.Bayes(demo["y"], sub, "BayesCpi", [0.95, 0.05], niter=20, nburn=10, thin=2, seed=3, verbose=False)
    r2 = H.Bayes(demo["y"], demo["M"], "BayesCpi", [0.95, 0.05], niter=20, nburn=10, thin=2, seed=3, verbose=False)
    assert np.array_equal(r["alpha"], r2["alpha"])


@pytest.mark.parametrize("shape", [{"HB_Q2M_G": "1"}, {"HB_Q2M_G": "2"}, {"HB_Q2M_G": "3"}, {"HB_Q2M_G": "1", "HB_Q2M_CT": "8"},
                                   {"HB_Q2M_G": "1", "HB_Q2M_CT": "16", "HB_Q2M_SC": "0"}, {"HB_Q2M_G": "0", "HB_Q2M_SC": "0"}])
def test_every_shape_of_the_matrix_core_matvec_gives_the_same_integers(shape, monkeypatch):
    """k_dotq2m's template shapes (hb_dotq2.hpp; round 5): 256-individual stages singly or in pairs, 512-individual stages of whole-line DMA
    pieces in plain and in bank-conflict-free lane order (the default is G = 0), 64 / 128 / 256 columns per wave, one accumulator set per
    genotype scale or one in all. Integer sums in another order: every shape must give the int8 layout's dot products bit for bit, on
    a padded length that is a multiple of 512 and on one that is not (the 512-individual shapes then fall back to 256)."""
    for k, v in shape.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(31)
    for n, m, panel in ((2000, 1024, 512), (1300, 640, 128), (5100, 2048, 512)):    # ld = 2048, 1536 (odd multiple of 256), 5120
        X = rand_geno(rng, n, m)
        r = rng.normal(0, 3.0, n)
        r[rng.integers(0, n, 5)] *= 1e6
        with H.Context(n, m, panel=panel, precise=2) as c:
            c.upload(X)
            c.set_residual(r, np.zeros(n))
            d8 = c.dot()
            c.set_layout(2, keep_int8=False)
            c.set_matvec_kernel(2)
            assert np.array_equal(c.dot(), d8), (shape, n, m)


def test_two_bit_layout_pack_unpack_dot_and_products():
    """SURVEY §8 f1 (second half): genotypes resident at PLINK's density, 2 bits each (reference src/read_bed.cpp:116-167 is
    the format; hb_dotq2.hpp the packed word). Packing on the device, dropping the int8 copy, unpacking (genotype download),
    the fixed-point mat-vec and X * alpha must all give what the int8 layout gives — the dot products bit for bit: they are
    exact integers either way (checked against a Python big-integer dot product)."""
    rng = np.random.default_rng(22)
    for n, m, panel in ((777, 300, 64), (1300, 1100, 128), (5000, 1024, 512)):   # ld = 1024 (two stages of 512), 1536 (an odd multiple of 256), 5120
        X = rand_geno(rng, n, m)
        r = rng.normal(0, 3.0, n)
        r[rng.integers(0, n, 5)] *= 1e6
        with H.Context(n, m, panel=panel, precise=2) as c:
            c.upload(X)
            c.set_residual(r, np.zeros(n))
            d8 = c.dot()
            xpx8, vx8, sumvx8, nvar08 = c.marker_stats()
            c.set_layout(2, keep_int8=True)
            assert c.layout() == (2, True)
            assert np.array_equal(c.dot(), d8)                # identical integers -> identical doubles
            c.set_layout(2, keep_int8=False)
            assert c.layout() == (2, False)
            d2 = c.dot()
            assert np.array_equal(d2, d8)
            for kind in (1, 2, 0):                            # k_dotq2r (individuals across the lanes), k_dotq2m (the digit planes
                c.set_matvec_kernel(kind)                     # as an int8 GEMM on the matrix cores: an A/B), back to k_dotq2:
                assert np.array_equal(c.dot(), d8), kind      # integer sums in another order — the same integers
            assert np.array_equal(c.download(), X)            # unpacked on the device
            E = 53 - int(np.floor(np.log2(np.abs(r).max())))
            q = np.rint(np.ldexp(r, E))
            exact = np.array([int(v) for v in (X.astype(object).T @ q.astype(np.int64).astype(object))], dtype=object)
            assert np.array_equal(d2, np.array([float(v) * 2.0 ** -E for v in exact]))
            alpha = np.zeros(m)
            alpha[rng.integers(0, m, 40)] = rng.normal(0, 1, 40)
            xa = np.zeros(n)
            H._lib.check(c.L.hb_ctx_matvec(c.h, alpha.ctypes.data, xa.ctypes.data))
            np.testing.assert_allclose(xa, X.astype(np.float64) @ alpha, rtol=0, atol=1e-9)
            A = np.zeros((m, 3))
            A[rng.integers(0, m, 30), rng.integers(0, 3, 30)] = rng.normal(0, 1, 30)
            np.testing.assert_allclose(c.matmul(A), X.astype(np.float64) @ A, rtol=0, atol=1e-9)
            xpx2, vx2, sumvx2, nvar02 = c.marker_stats()      # computed before the int8 copy was dropped: still valid
            assert np.array_equal(xpx2, xpx8) and np.array_equal(vx2, vx8) and nvar02 == nvar08
            with pytest.raises(H.HibayesError, match="2-bit layout only"):
                c.upload(X)
            c.set_layout(8)                                   # back: the int8 copy is unpacked again
            assert c.layout() == (8, True) and np.array_equal(c.download(), X) and np.array_equal(c.dot(), d8)


def test_two_bit_layout_refuses_what_it_cannot_hold():
    rng = np.random.default_rng(23)
    X = rand_geno(rng, 300, 200, signed=True)               # -1 / 0 / 1 coding (README.md:55-59): int8 layout only
    with H.Context(300, 200, panel=64, precise=2) as c:
        c.upload(X)
        with pytest.raises(H.HibayesError, match="codes 0..3"):
            c.set_layout(2)
        assert c.layout() == (8, True)
    with H.Context(300, 200, panel=64, precise=1) as c:
        c.upload(np.abs(X))
        with pytest.raises(H.HibayesError, match="precise = 2"):
            c.set_layout(2)


def test_gram_rebuild_from_the_two_bit_layout_alone():
    """A context that dropped its int8 copy (hb_ctx_set_layout(c, 2, 0)) and is then asked for a wider band rebuilds the Gram blocks
    from the packed genotypes: a window of panels at a time is unpacked into a scratch buffer (never the whole matrix), and every
    block is the exact int32 product."""
    rng = np.random.default_rng(31)
    n, panel, m = 900, 64, 64 * 9 + 5
    X = rand_geno(rng, n, m)
    with H.Context(n, m, panel=panel, precise=2) as c:
        c.upload(X)
        c.set_pipeline(1, 1, 1)
        c.build_gram()
        c.set_layout(2, keep_int8=False)
        assert c.layout() == (2, False)
        c.set_pipeline(1, 2, 2)                                  # band 5 > the stored band 1: the next build must come from X2
        c.build_gram()
        assert c.layout() == (2, False)                          # (no int8 copy was materialised behind the caller's back)
        band = c.pipeline()[3]
        Xi = np.zeros((n, (m + panel - 1) // panel * panel), dtype=np.int64)
        Xi[:, :m] = X
        for p in range((m + panel - 1) // panel):
            for l in range(0, min(band, p) + 1):
                a, b = Xi[:, (p - l) * panel:(p - l + 1) * panel], Xi[:, p * panel:(p + 1) * panel]
                assert np.array_equal(c.gram_band(p, l), a.T @ b), (p, l)
