"""ctypes front-end of the CPU oracle (oracle/hb_oracle.c).

ORACLE — TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package hibayes_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RNG_R, RNG_PHILOX = 0, 1
MAX_FOLD = 16


class _Args(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32),
        ("y", C.c_void_p), ("X", C.c_void_p), ("X8", C.c_void_p),
        ("model", C.c_char_p),
        ("Pi", C.c_void_p), ("n_pi", C.c_int32),
        ("fold", C.c_void_p), ("n_fold", C.c_int32),
        ("C", C.c_void_p), ("nc", C.c_int32),
        ("R", C.c_void_p), ("nr", C.c_int32),
        ("niter", C.c_int32), ("nburn", C.c_int32), ("thin", C.c_int32),
        ("dfvr", C.c_double), ("s2vr", C.c_double), ("vg", C.c_double), ("dfvg", C.c_double),
        ("s2vg", C.c_double), ("ve", C.c_double), ("dfve", C.c_double), ("s2ve", C.c_double),
        ("windindx", C.c_void_p),
        ("threads", C.c_int32),
        ("rng_kind", C.c_int32), ("seed", C.c_uint64), ("marker_offset", C.c_int64),
        ("trace_iter", C.c_int32),
        ("trace_rhs", C.c_void_p), ("trace_cls", C.c_void_p), ("trace_g", C.c_void_p),
        ("g_init", C.c_void_p),
        ("warm", C.c_void_p),
    ]


class Warm(C.Structure):
    """hbo_warm == hb_warm_state (include/hibayes_gpu.h), field for field."""
    _fields_ = [("mu", C.c_double), ("vare", C.c_double), ("varg", C.c_double), ("lambda2", C.c_double),
                ("pi", C.c_double * 8), ("vargL", C.c_void_p)]


def make_warm(mu, vare, varg=0.0, pi=(), lambda2=0.0, vargL=None):
    w = Warm()
    w.mu, w.vare, w.varg, w.lambda2 = float(mu), float(vare), float(varg), float(lambda2)
    for j, p in enumerate(pi):
        w.pi[j] = float(p)
    w._keep = None
    if vargL is not None:
        w._keep = np.ascontiguousarray(vargL, dtype=np.float64)
        w.vargL = w._keep.ctypes.data
    return w


class _Out(C.Structure):
    _fields_ = [
        ("Vg", C.c_double), ("Ve", C.c_double), ("h2", C.c_double), ("mu", C.c_double),
        ("n_records", C.c_int32), ("nzct", C.c_int32), ("nw", C.c_int32), ("n_levels", C.c_int32),
        ("beta", C.c_void_p), ("alpha", C.c_void_p), ("pi", C.c_void_p), ("Vr", C.c_void_p),
        ("r_est", C.c_void_p), ("g", C.c_void_p), ("e", C.c_void_p), ("pip", C.c_void_p),
        ("gwas", C.c_void_p),
        ("s_Vg", C.c_void_p), ("s_Ve", C.c_void_p), ("s_h2", C.c_void_p), ("s_mu", C.c_void_p),
        ("s_beta", C.c_void_p), ("s_alpha", C.c_void_p), ("s_pi", C.c_void_p), ("s_Vr", C.c_void_p),
        ("vary", C.c_double), ("sumvx", C.c_double), ("varg0", C.c_double), ("s2varg", C.c_double),
        ("vara0", C.c_double), ("s2vara", C.c_double), ("vare0", C.c_double),
        ("lambda2_0", C.c_double), ("rate0", C.c_double),
        ("nvar0", C.c_int32),
        ("xpx", C.c_void_p), ("vx", C.c_void_p),
        ("loop_seconds", C.c_double), ("iters_done", C.c_int32),
        ("error", C.c_char * 256),
        ("last", Warm), ("g_last", C.c_void_p), ("vargL_last", C.c_void_p),
    ]


class _SbArgs(C.Structure):
    _fields_ = [
        ("m", C.c_int32), ("sumstat", C.c_void_p), ("ldm", C.c_void_p), ("model", C.c_char_p),
        ("Pi", C.c_void_p), ("n_pi", C.c_int32), ("fold", C.c_void_p), ("n_fold", C.c_int32),
        ("niter", C.c_int32), ("nburn", C.c_int32), ("thin", C.c_int32),
        ("vg", C.c_double), ("dfvg", C.c_double), ("s2vg", C.c_double), ("ve", C.c_double), ("dfve", C.c_double), ("s2ve", C.c_double),
        ("windindx", C.c_void_p), ("rng_kind", C.c_int32), ("seed", C.c_uint64),
    ]


class _SbOut(C.Structure):
    _fields_ = [
        ("Vg", C.c_double), ("Ve", C.c_double), ("h2", C.c_double),
        ("n_records", C.c_int32), ("nzct", C.c_int32), ("nw", C.c_int32), ("n", C.c_int32), ("count_y", C.c_int32),
        ("vary", C.c_double),
        ("alpha", C.c_void_p), ("pi", C.c_void_p), ("pip", C.c_void_p), ("gwas", C.c_void_p),
        ("s_Vg", C.c_void_p), ("s_Ve", C.c_void_p), ("s_h2", C.c_void_p), ("s_alpha", C.c_void_p), ("s_pi", C.c_void_p),
        ("r_hat", C.c_void_p), ("g_last", C.c_void_p),
        ("loop_seconds", C.c_double), ("iters_done", C.c_int32), ("error", C.c_char * 256),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libhb_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        try:
            _LIB = C.CDLL(so)
        except OSError:
            _LIB = C.CDLL(build(force=True))
        L = _LIB
        L.hbo_bayes.argtypes = [C.POINTER(_Args), C.POINTER(_Out)]
        L.hbo_bayes.restype = C.c_int
        L.hbo_sbayes.argtypes = [C.POINTER(_SbArgs), C.POINTER(_SbOut)]
        L.hbo_sbayes.restype = C.c_int
        L.hbo_decode_bed.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int, C.c_void_p]
        L.hbo_decode_bed.restype = C.c_int
        L.hbo_mt_set_seed.argtypes = [C.c_void_p, C.c_uint32]
        L.hbo_mt_unif_rand.argtypes = [C.c_void_p]
        L.hbo_mt_unif_rand.restype = C.c_double
        L.hbo_mt_norm_rand.argtypes = [C.c_void_p]
        L.hbo_mt_norm_rand.restype = C.c_double
        L.hbo_mt_exp_rand.argtypes = [C.c_void_p]
        L.hbo_mt_exp_rand.restype = C.c_double
        L.hbo_mt_rgamma.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.hbo_mt_rgamma.restype = C.c_double
        L.hbo_qnorm.argtypes = [C.c_double]
        L.hbo_qnorm.restype = C.c_double
        L.hbo_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hbo_philox_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.hbo_philox_uniform.restype = C.c_double
        L.hbo_philox_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.hbo_philox_normal.restype = C.c_double
        L.hbo_philox_block.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.hbo_stream_init_r.argtypes = [C.c_void_p, C.c_uint32]
        L.hbo_stream_init_philox.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        for f in ("hbo_unif", "hbo_norm"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_double
        L.hbo_gamma.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.hbo_gamma.restype = C.c_double
        L.hbo_chisq.argtypes = [C.c_void_p, C.c_double]
        L.hbo_chisq.restype = C.c_double
        L.hbo_invgauss.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.hbo_invgauss.restype = C.c_double
    return _LIB


class MT:
    """R's Mersenne-Twister after set.seed(seed)."""

    def __init__(self, seed):
        self._buf = C.create_string_buffer(624 * 4 + 16)
        lib().hbo_mt_set_seed(self._buf, seed)

    def unif(self):
        return lib().hbo_mt_unif_rand(self._buf)

    def norm(self):
        return lib().hbo_mt_norm_rand(self._buf)

    def exp(self):
        return lib().hbo_mt_exp_rand(self._buf)

    def gamma(self, shape, scale=1.0):
        return lib().hbo_mt_rgamma(self._buf, shape, scale)


class Stream:
    """Sequential draw stream (R kind or Philox kind) as the sampler's host draws use it."""

    def __init__(self, kind, seed, sub=0, blk0=0):
        self._buf = C.create_string_buffer(624 * 4 + 64)
        if kind == RNG_R:
            lib().hbo_stream_init_r(self._buf, seed)
        else:
            lib().hbo_stream_init_philox(self._buf, seed, sub, blk0)

    def unif(self):
        return lib().hbo_unif(self._buf)

    def norm(self):
        return lib().hbo_norm(self._buf)

    def gamma(self, shape, scale=1.0):
        return lib().hbo_gamma(self._buf, shape, scale)

    def chisq(self, df):
        return lib().hbo_chisq(self._buf, df)

    def invgauss(self, mu, lam):
        return lib().hbo_invgauss(self._buf, mu, lam)


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().hbo_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return o


def philox_block(seed, sub, blk):
    o = np.zeros(4, dtype=np.uint32)
    lib().hbo_philox_block(seed, sub, blk, o.ctypes.data)
    return o


def decode_bed(raw, nind, nsnp, impute=True):
    raw = np.frombuffer(raw, dtype=np.uint8)
    out = np.zeros((nind, nsnp), dtype=np.int8, order="F")
    rc = lib().hbo_decode_bed(raw.ctypes.data, raw.size, nind, nsnp, int(impute), out.ctypes.data)
    if rc:
        raise ValueError("hbo_decode_bed failed: %d" % rc)
    return out


def _nan(v):
    return float("nan") if v is None else float(v)


def bayes(y, X, model, Pi, fold=None, Cmat=None, R=None, niter=50000, nburn=20000, thin=5,
          dfvr=None, s2vr=None, vg=None, dfvg=None, s2vg=None, ve=None, dfve=None, s2ve=None,
          windindx=None, threads=1, rng=RNG_PHILOX, seed=666666, marker_offset=0,
          store_alpha=False, trace_iter=None, g_init=None, warm=None):
    """Mirror of reference Bayes() (src/Bayes.cpp:60-88). X: n x m, float64 or int8."""
    L = lib()
    y = np.ascontiguousarray(y, dtype=np.float64)
    n = y.size
    X = np.asarray(X)
    if X.dtype == np.int8:
        Xa = np.asfortranarray(X)
    else:
        Xa = np.asfortranarray(X, dtype=np.float64)
    assert Xa.shape[0] == n, "Number of individuals not equals."
    m = Xa.shape[1]
    Pi = np.ascontiguousarray(Pi, dtype=np.float64)
    a = _Args()
    a.n, a.m = n, m
    a.y = y.ctypes.data
    if Xa.dtype == np.int8:
        a.X, a.X8 = None, Xa.ctypes.data
    else:
        a.X, a.X8 = Xa.ctypes.data, None
    a.model = model.encode()
    a.Pi, a.n_pi = Pi.ctypes.data, Pi.size
    keep = [y, Xa, Pi]
    if fold is not None:
        fold = np.ascontiguousarray(fold, dtype=np.float64)
        a.fold, a.n_fold = fold.ctypes.data, fold.size
        keep.append(fold)
    nc = 0
    if Cmat is not None:
        Cm = np.asfortranarray(Cmat, dtype=np.float64).reshape(n, -1, order="F")
        nc = Cm.shape[1]
        a.C, a.nc = Cm.ctypes.data, nc
        keep.append(Cm)
    nr = 0
    if R is not None:
        Rm = np.asarray(R, dtype=object).reshape(n, -1)
        nr = Rm.shape[1]
        strs = [str(Rm[i, j]).encode() for j in range(nr) for i in range(n)]
        arr = (C.c_char_p * len(strs))(*strs)
        a.R, a.nr = C.cast(arr, C.c_void_p), nr
        keep += [strs, arr]
    a.niter, a.nburn, a.thin = niter, nburn, thin
    a.dfvr, a.s2vr, a.vg, a.dfvg = _nan(dfvr), _nan(s2vr), _nan(vg), _nan(dfvg)
    a.s2vg, a.ve, a.dfve, a.s2ve = _nan(s2vg), _nan(ve), _nan(dfve), _nan(s2ve)
    nw = 0
    if windindx is not None:
        w = np.ascontiguousarray(windindx, dtype=np.uint32)
        a.windindx = w.ctypes.data
        nw = int(w.max())
        keep.append(w)
    a.threads = threads
    if g_init is not None:
        gi = np.ascontiguousarray(g_init, dtype=np.float64)
        assert gi.size == m
        a.g_init = gi.ctypes.data
        keep.append(gi)
    if warm is not None:   # a Warm (make_warm) or a dict of its arguments
        if isinstance(warm, dict):
            warm = make_warm(**warm)
        a.warm = C.addressof(warm)
        keep.append(warm)
    a.rng_kind, a.seed, a.marker_offset = rng, seed, marker_offset
    nrec = max((niter - nburn) // thin, 0)
    o = _Out()
    res = {}

    def buf(name, shape, dtype=np.float64):
        arr = np.zeros(shape, dtype=dtype, order="F")
        res[name] = arr
        return arr.ctypes.data

    o.beta = buf("beta", nc) if nc else None
    o.alpha = buf("alpha", m)
    o.pi = buf("pi", Pi.size)
    o.Vr = buf("Vr", nr) if nr else None
    o.r_est = buf("r", n * max(nr, 1)) if nr else None
    o.g = buf("g", n)
    o.e = buf("e", n)
    o.pip = buf("pip", m)
    o.gwas = buf("gwas", nw) if nw else None
    o.s_Vg, o.s_Ve = buf("s_Vg", nrec), buf("s_Ve", nrec)
    o.s_h2, o.s_mu = buf("s_h2", nrec), buf("s_mu", nrec)
    o.s_beta = buf("s_beta", (nc, nrec)) if nc else None
    o.s_alpha = buf("s_alpha", (m, nrec)) if store_alpha else None
    o.s_pi = buf("s_pi", (Pi.size, nrec))
    o.s_Vr = buf("s_Vr", (nr, nrec)) if nr else None
    o.xpx, o.vx = buf("xpx", m), buf("vx", m)
    o.g_last = buf("g_last", m)
    if model == "BayesL":
        o.vargL_last = buf("vargL_last", m)
    if trace_iter is not None:
        a.trace_iter = trace_iter
        a.trace_rhs = buf("trace_rhs", m)
        a.trace_cls = buf("trace_cls", m, np.int32)
        a.trace_g = buf("trace_g", m)
    rc = L.hbo_bayes(C.byref(a), C.byref(o))
    if rc:
        raise RuntimeError(o.error.decode())
    for k in ("Vg", "Ve", "h2", "mu", "n_records", "nzct", "nw", "n_levels", "vary", "sumvx", "varg0",
              "s2varg", "vara0", "s2vara", "vare0", "lambda2_0", "rate0", "nvar0", "loop_seconds",
              "iters_done"):
        res[k] = getattr(o, k)
    if nr:
        res["r"] = res["r"][: o.n_levels]
    lw = {"mu": o.last.mu, "vare": o.last.vare, "varg": o.last.varg, "lambda2": o.last.lambda2, "pi": [o.last.pi[j] for j in range(Pi.size)]}
    if model == "BayesL":
        lw["vargL"] = res.pop("vargL_last")
    res["last"] = {"g": res.pop("g_last"), "warm": lw}   # bayes(..., g_init=last["g"], warm=last["warm"]) continues the chain
    del keep
    return res


def sbayes(sumstat, ldm, model, Pi, fold=None, niter=50000, nburn=20000, thin=5, vg=None, dfvg=None, s2vg=None, ve=None,
           dfve=None, s2ve=None, windindx=None, rng=RNG_PHILOX, seed=666666, store_alpha=False):
    """Mirror of reference SBayesD() (src/SBayesD.cpp:5-26). sumstat: m x 4 (MAF, BETA, SE, NMISS; NaN = NA), ldm: m x m dense."""
    L = lib()
    ss = np.asfortranarray(sumstat, dtype=np.float64)
    ld = np.asfortranarray(ldm, dtype=np.float64)
    m = ss.shape[0]
    if ld.shape[0] != m:
        raise RuntimeError("Number of SNPs not equals.")
    Pi = np.ascontiguousarray(Pi, dtype=np.float64)
    a = _SbArgs()
    a.m, a.sumstat, a.ldm, a.model = m, ss.ctypes.data, ld.ctypes.data, model.encode()
    a.Pi, a.n_pi = Pi.ctypes.data, Pi.size
    keep = [ss, ld, Pi]
    if fold is not None:
        fo = np.ascontiguousarray(fold, dtype=np.float64)
        a.fold, a.n_fold = fo.ctypes.data, fo.size
        keep.append(fo)
    a.niter, a.nburn, a.thin = niter, nburn, thin
    a.vg, a.dfvg, a.s2vg, a.ve, a.dfve, a.s2ve = _nan(vg), _nan(dfvg), _nan(s2vg), _nan(ve), _nan(dfve), _nan(s2ve)
    nw = 0
    if windindx is not None:
        w = np.ascontiguousarray(windindx, dtype=np.uint32)
        a.windindx, nw = w.ctypes.data, int(w.max())
        keep.append(w)
    a.rng_kind, a.seed = rng, seed
    nrec = max((niter - nburn) // thin, 0)
    o, res = _SbOut(), {}

    def buf(name, shape):
        arr = np.zeros(shape, order="F")
        res[name] = arr
        return arr.ctypes.data

    o.alpha, o.pi, o.pip = buf("alpha", m), buf("pi", Pi.size), buf("pip", m)
    o.gwas = buf("gwas", nw) if nw else None
    o.s_Vg, o.s_Ve, o.s_h2 = buf("s_Vg", nrec), buf("s_Ve", nrec), buf("s_h2", nrec)
    o.s_alpha = buf("s_alpha", (m, nrec)) if store_alpha else None
    o.s_pi = buf("s_pi", (Pi.size, nrec))
    o.r_hat, o.g_last = buf("r_hat", m), buf("g_last", m)
    if L.hbo_sbayes(C.byref(a), C.byref(o)):
        raise RuntimeError(o.error.decode())
    for k in ("Vg", "Ve", "h2", "n_records", "nzct", "nw", "n", "count_y", "vary", "loop_seconds", "iters_done"):
        res[k] = getattr(o, k)
    del keep
    return res
