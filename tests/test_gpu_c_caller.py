"""A caller on the C side of the boundary: examples/ibrm_demo.c (plain C99 against include/hibayes_gpu.h) compiled with gcc,
run on the reference's demo data, compared with the committed golden vectors of the oracle (tests/golden/make_golden.py: the
ibrm(T1 ~ 1, BayesCpi, niter = 2000, nburn = 1200, thin = 5) example of reference R/bayes.r:93-94)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def build(tmp_path):
    exe = str(tmp_path / "ibrm_demo")
    libdir = os.path.join(ROOT, "hibayes_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "ibrm_demo.c"), "-L", libdir, "-lhibayes_gpu", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def parse(out):
    res = {}
    for line in out.decode().splitlines():
        f = line.split()
        if f and f[0] in ("n", "m", "n_records", "nzct", "Vg", "Ve", "h2", "mu", "pi", "alpha", "pip"):
            res[f[0]] = np.array([float(x) for x in f[1:]])
    return res


@pytest.mark.parametrize("bits", [8, 2])
def test_c_caller_reproduces_the_golden_demo_fit(tmp_path, bits):
    exe = build(tmp_path)
    out = subprocess.check_output([exe, os.path.join(ROOT, "tests", "golden", "demo", "demo"), "2000", "1200", "5", str(bits)])
    r = parse(out)
    g = np.load(os.path.join(ROOT, "tests", "golden", "demo_bayescpi_philox.npz"))
    assert r["n"][0] == 300 and r["m"][0] == 1000 and r["n_records"][0] == 160 and r["nzct"][0] == 800
    np.testing.assert_allclose([r["Vg"][0], r["Ve"][0], r["h2"][0], r["mu"][0]], g["scal"], rtol=1e-7)
    np.testing.assert_allclose(r["pi"], g["pi"], rtol=1e-7)
    np.testing.assert_allclose(r["alpha"], g["alpha"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(r["pip"], g["pip"], rtol=0, atol=1e-12)


def test_c_caller_sees_the_reference_error_texts(tmp_path):
    exe = build(tmp_path)
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "demo", "demo"), "5", "10", "1"], capture_output=True)
    assert p.returncode == 1 and b"Number of total iteration ('niter') shold be larger than burn-in ('nburn')." in p.stderr
