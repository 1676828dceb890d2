"""Host mirror of the reference's summary-level interface on a dense LD matrix: SBayesD() (src/SBayesD.cpp:5-26, the generated
wrapper R/RcppExports.R:8-10) and the slice of sbrm() (R/sbayes.r:101-239) that leads to it. Everything computes on the device
through hb_sbayes_run (include/hibayes_gpu.h); there is no CPU fallback."""
import ctypes as C

import numpy as np

from ._lib import LOG_FN, SBayesArgs, SBayesOut, check, lib


def SBayesD(sumstat, ldm, model, Pi, niter=50000, nburn=20000, thin=5, fold=None, windindx=None, vg=None, dfvg=None, s2vg=None,
            ve=None, dfve=None, s2ve=None, outfreq=100, threads=0, verbose=True, *, seed=666666, device=0, store_alpha=True, log=None):
    """sumstat: m x 4 (MAF, BETA, SE, NMISS — what sbrm() keeps of the COJO file, R/sbayes.r:207; NaN = NA); ldm: m x m dense."""
    L = lib()
    ss = np.asfortranarray(sumstat, dtype=np.float64)
    ld = np.asfortranarray(ldm, dtype=np.float64)
    if ss.ndim != 2 or ss.shape[1] != 4:
        raise ValueError("sumstat must have the four columns MAF, BETA, SE, NMISS")
    m = ss.shape[0]
    a = SBayesArgs()
    a.m = m if ld.ndim == 2 and ld.shape[0] == m and ld.shape[1] == m else -1   # -> "Number of SNPs not equals."
    a.sumstat, a.ld_sumstat, a.ldm, a.ld_ldm = ss.ctypes.data, m, ld.ctypes.data, (ld.shape[0] if ld.ndim == 2 else 0)
    a.model = model.encode()
    pv = np.ascontiguousarray(Pi, dtype=np.float64)
    a.Pi, a.n_pi = pv.ctypes.data, pv.size
    keep = [ss, ld, pv]
    if fold is not None:
        fv = np.ascontiguousarray(fold, dtype=np.float64)
        a.fold, a.n_fold = fv.ctypes.data, fv.size
        keep.append(fv)
    a.niter, a.nburn, a.thin = int(niter), int(nburn), int(thin)
    nw = 0
    if windindx is not None:
        w = np.ascontiguousarray(windindx, dtype=np.uint32)
        a.windindx, nw = w.ctypes.data, int(w.max())
        keep.append(w)
    for name, val in (("vg", vg), ("dfvg", dfvg), ("s2vg", s2vg), ("ve", ve), ("dfve", dfve), ("s2ve", s2ve)):
        if val is not None:
            setattr(a, "has_" + name, 1)
            setattr(a, name, float(val))
    a.outfreq, a.threads, a.verbose = int(outfreq), int(threads), int(bool(verbose))
    a.seed, a.device, a.store_alpha = int(seed), int(device), int(bool(store_alpha))
    if log is not None:
        cb = LOG_FN(lambda line, _u: log(line.decode("utf-8", "replace")))
        a.log = cb
        keep.append(cb)
    nrec = max((a.niter - a.nburn) // max(a.thin, 1), 0)
    o, res, mc = SBayesOut(), {}, {}

    def buf(d, name, shape):
        arr = np.zeros(shape, order="F")
        d[name] = arr
        return arr.ctypes.data

    mm = max(m, 1)
    o.alpha, o.pi, o.pip = buf(res, "alpha", mm), buf(res, "pi", pv.size), buf(res, "pip", mm)
    o.gwas = buf(res, "gwas", nw) if nw else None
    o.s_Vg, o.s_Ve, o.s_h2 = buf(mc, "Vg", (1, nrec)), buf(mc, "Ve", (1, nrec)), buf(mc, "h2", (1, nrec))
    o.s_alpha = buf(mc, "alpha", (mm, nrec)) if store_alpha else None
    o.s_pi = buf(mc, "pi", (pv.size, nrec))
    o.r_hat, o.g_last = buf(res, "r_hat", mm), buf(res, "g_last", mm)
    check(L.hb_sbayes_run(C.byref(a), C.byref(o)))
    for k in ("Vg", "Ve", "h2", "n_records", "nzct", "nw", "n", "count_y"):
        res[k] = getattr(o, k)
    res["MCMCsamples"] = mc
    res["timing"] = {"setup_seconds": o.setup_seconds, "loop_seconds": o.loop_seconds, "iters_done": o.iters_done, "mean_events": o.mean_events}
    del keep
    return res


def sbrm(sumstat, ldm, method="BayesB", map=None, Pi=None, fold=None, niter=None, nburn=None, thin=5, windsize=None, windnum=None,
         windindx=None, vg=None, dfvg=None, s2vg=None, ve=None, dfve=None, s2ve=None, printfreq=100, seed=666666, threads=4, verbose=True, **kw):
    """The dense-LD slice of sbrm() (R/sbayes.r:101-239): the ldm type check (:126-132), GWAS windows cut from `map` by windsize /
    windnum (:135-187; res["gwas"] then carries WIND / CHR / NUM / START / END / WPPA like the reference's data frame, :231-234),
    defaults (:189-203), the column selection sumstat[, c(4, 5, 6, 8)] of the 8-column COJO table (:207), then SBayesD().
    Sparse LD matrices (dgCMatrix -> SBayesS) and method = "CG" are outside the GPU path and refused."""
    try:
        import scipy.sparse as sp
        if sp.issparse(ldm):
            raise NotImplementedError("a sparse ldm (dgCMatrix: SBayesS, src/SBayesS.cpp) is outside the GPU path; pass the dense LD matrix")
    except ImportError:
        pass
    if not (isinstance(ldm, np.ndarray) or hasattr(ldm, "__array__")):
        raise ValueError("Unrecognized type of ldm.")
    if method == "CG":
        raise NotImplementedError("method = 'CG' (conjgt_den / conjgt_spa) is outside the GPU path")
    windinfo = None
    if windsize is not None or windnum is not None:
        if method in ("BayesA", "BayesRR", "BayesL"):
            raise ValueError("can not implement GWAS analysis for the method: " + method)
        if map is None:
            raise ValueError("map information must be provided.")
        from .bayes import _map_columns
        from .windows import cutwind_by_bp, cutwind_by_num
        chrom, bp = _map_columns(map)          # R/sbayes.r:140-172, the same checks as ibrm()
        if windnum is not None:
            if len(chrom) < windnum:
                raise ValueError("Number of markers specified in a window is larger than the total number of markers.")
            windindx = cutwind_by_num(chrom, bp, windnum)
        else:
            if bp.max() < windsize:
                raise ValueError("Maximum of physical position is smaller than wind size.")
            windindx = cutwind_by_bp(chrom, bp, windsize)
        nwin = int(windindx.max())
        windinfo = {"WIND": ["wind%d" % (w + 1) for w in range(nwin)],
                    "CHR": [chrom[np.flatnonzero(windindx == w + 1)[0]] for w in range(nwin)],
                    "NUM": [int((windindx == w + 1).sum()) for w in range(nwin)],
                    "START": [float(bp[windindx == w + 1].min()) for w in range(nwin)],
                    "END": [float(bp[windindx == w + 1].max()) for w in range(nwin)]}
    if niter is None:
        niter = 50000 if method == "BayesR" else 20000
    if nburn is None:
        nburn = 30000 if method == "BayesR" else 12000
    if thin >= (niter - nburn):
        raise ValueError("bad setting for collecting frequency 'thin'.")
    if printfreq <= 0:
        verbose = False
    if Pi is None:
        if method == "BayesR":
            Pi = [0.95, 0.02, 0.02, 0.01]
            if fold is None:
                fold = [0, 0.0001, 0.001, 0.01]
        else:
            Pi = [0.95, 0.05]
    ss = np.asarray(sumstat, dtype=np.float64)
    if ss.ndim == 2 and ss.shape[1] >= 8:      # the COJO table (:207); a 4-column matrix is taken as MAF, BETA, SE, NMISS already
        ss = ss[:, [3, 4, 5, 7]]
    res = SBayesD(ss, ldm, method, Pi, niter=niter, nburn=nburn, thin=thin, fold=fold, windindx=windindx, vg=vg, dfvg=dfvg, s2vg=s2vg,
                  ve=ve, dfve=dfve, s2ve=s2ve, outfreq=printfreq, threads=threads, verbose=verbose, seed=seed, **kw)
    if windinfo is not None:
        res["gwas"] = dict(windinfo, WPPA=res["gwas"])
    res["call"] = "b ~ nD^{-1}V alpha + e"
    res["model"] = "Summary level Bayesian model fit by [%s]" % method
    return res
