"""Host-side mirror of the reference's operator interface for the individual-level sampler.

`Bayes()`  == the Rcpp export `Bayes(...)` (reference src/Bayes.cpp:60-88, R/RcppExports.R:4-6):
             same argument names, order, defaults and returned list fields (src/Bayes.cpp:919-1040).
`ibrm()`   == R/bayes.r:121-320 restricted to what reaches Bayes(): formula -> fixed matrix X and
             random-effect columns R, ID alignment (:161-165), NA mask (:199-207), defaults
             (:264-279), the call (:296) and the GEBV post-step (:303-308).

Both run the sampler through libhibayes_gpu.so (hand-written gfx950 kernels); nothing here
computes on the CPU except argument marshalling and the tiny GEBV/`e` bookkeeping ibrm() does in R.
"""
import ctypes as ct
import re

import numpy as np

from . import _lib
from ._lib import BayesArgs, BayesOut, HibayesError, check, lib

METHODS = ("BayesCpi", "BayesA", "BayesL", "BSLMM", "BayesR", "BayesB", "BayesC", "BayesBpi", "BayesRR")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def Bayes(y, X, model, Pi, Kival=None, Ki=None, C_=None, R=None, fold=None, niter=50000, nburn=20000,
          thin=5, epsl_y_J=None, epsl_Gi=None, epsl_index=None, dfvr=None, s2vr=None, vg=None,
          dfvg=None, s2vg=None, ve=None, dfve=None, s2ve=None, windindx=None, outfreq=100, threads=0,
          verbose=True, *, seed=666666, device=0, panel=0, precise=2, store_alpha=True,
          comm=None, m_global=None, m_offset=0, log=None, C=None, g_init=None, ctx=None, sync_every_blocks=1, genotype_bits=0,
          shard_rows=False, n_global=None, row_offset=0, warm=None):
    """Individual-level Gibbs sampler on one MI355X (or one marker shard of it when `comm` is given).

    X is n x m: int8 (fast path, no double blow-up) or any integer-valued float array in the
    reference's layout. Returns a dict with the fields of the reference's Rcpp::List.
    `precise` selects the arithmetic of the panel mat-vec x_j . yadj (everything else is fp64 in every mode):
    2 (default) exact fixed point — the fp64 residual as 7 int8 digit planes, int8 x int8 -> int32 dot products, error below
    an fp64 ddot's own rounding and independent of the launch geometry; 1 fp64 FMA; 0 the fp32 image of the residual.
    `ctx` (an engine.Context with genotypes already resident) replaces X: one upload serves several fits;
    the context's own pipeline geometry, seed addressing (m_offset) and panel are then used as they are.
    `genotype_bits`: resident layout of the sweep — 0 (default) auto: 2 bits per genotype where that is exact and the faster sweep
    (every code in 0..3, precise = 2, BayesB / BayesBpi / BayesC / BayesCpi / BayesR with up to four classes, at panel 512), int8 columns otherwise; 8 / 2 force one.
    The chain is the same bit for bit; the result's "resident_bits" reports what ran.
    `warm` (a dict mu / vare / varg / pi / lambda2 / vargL, or a _lib.WarmState) with `g_init` continues a chain from a reported
    state instead of the prior defaults (hb_warm_state, include/hibayes_gpu.h).
    """
    if C is not None and C_ is None:
        C_ = C
    L = lib()
    y = _f64(y).ravel()
    n = y.size
    a = BayesArgs()
    keep = []
    if ctx is not None:
        if X is not None:
            raise HibayesError(1, "give X or ctx, not both")
        if ctx.n != n:
            raise HibayesError(1, "Number of individuals not equals.")
        m = ctx.m
        a.ctx = ctx.h
    else:
        X = np.asarray(X)
        if X.ndim != 2 or X.shape[0] != n:
            raise HibayesError(1, "Number of individuals not equals.")
        m = X.shape[1]
    a.n, a.m = n, m
    a.y = y.ctypes.data
    if ctx is not None:
        pass
    elif X.dtype == np.int8:
        Xa = np.asfortranarray(X)
        a.X_i8, a.ld_i8 = Xa.ctypes.data, Xa.strides[1]
    else:
        Xa = np.asfortranarray(X, dtype=np.float64)
        a.X_f64, a.ld_f64 = Xa.ctypes.data, Xa.strides[1] // 8
    if ctx is None:
        keep.append(Xa)
    a.model = str(model).encode()
    Pi = _f64(Pi).ravel()
    a.Pi, a.n_pi = Pi.ctypes.data, Pi.size
    if Kival is not None:
        Kival = _f64(Kival); a.Kival = Kival.ctypes.data
    if Ki is not None:
        Ki = _f64(Ki); a.Ki = Ki.ctypes.data
    nc = 0
    if C_ is not None:
        Cm = np.asfortranarray(np.asarray(C_, dtype=np.float64).reshape(n, -1, order="F"))
        nc = Cm.shape[1]
        a.C, a.nc = Cm.ctypes.data, nc
        keep.append(Cm)
    nr = 0
    if R is not None:
        Rm = np.asarray(R, dtype=object).reshape(n, -1)
        nr = Rm.shape[1]
        strs = [None if Rm[i, j] is None else str(Rm[i, j]).encode() for j in range(nr) for i in range(n)]
        arr = (ct.c_char_p * len(strs))(*strs)
        a.R, a.nr = ct.cast(arr, ct.c_void_p), nr
        keep += [strs, arr]
    if fold is not None:
        fold = _f64(fold).ravel()
        a.fold, a.n_fold = fold.ctypes.data, fold.size
    a.niter, a.nburn, a.thin = int(niter), int(nburn), int(thin)
    if epsl_y_J is not None or epsl_Gi is not None or epsl_index is not None:
        dummy = np.zeros(1)
        keep.append(dummy)
        a.epsl_index = dummy.ctypes.data  # refused by the library with HB_ERR_UNSUPPORTED
    for name, val in (("dfvr", dfvr), ("s2vr", s2vr), ("vg", vg), ("dfvg", dfvg), ("s2vg", s2vg),
                      ("ve", ve), ("dfve", dfve), ("s2ve", s2ve)):
        if val is not None:
            setattr(a, "has_" + name, 1)
            setattr(a, name, float(val))
    nw = 0
    if windindx is not None:
        w = np.ascontiguousarray(windindx, dtype=np.uint32)
        if w.size != m:
            raise HibayesError(1, "windindx must have one entry per marker")
        a.windindx = w.ctypes.data
        nw = int(w.max())
        keep.append(w)
    a.outfreq, a.threads, a.verbose = int(outfreq), int(threads), int(bool(verbose))
    a.seed, a.device, a.panel, a.precise = int(seed), int(device), int(panel), int(precise)
    nrec = max((int(niter) - int(nburn)) // max(int(thin), 1), 0)
    a.store_alpha = int(bool(store_alpha))
    if comm is not None and (comm.world > 1 or hasattr(comm, "handle")):
        if comm.world > 1 and m_global is None and not shard_rows:
            raise HibayesError(1, "a sharded run needs m_global (markers over all ranks) and m_offset (first global marker of this shard)")
        a.rank, a.world = comm.rank, comm.world
        a.m_global = int(m_global if m_global is not None else m)
        a.m_offset = int(m_offset)
        if hasattr(comm, "handle"):      # RcclComm: the collective runs inside the library
            a.comm = comm.handle
        else:                            # TorchComm: the library calls back for every exchange
            cb, xbuf_ptr = comm.make_callback(L.hb_exchange_count(4096 if shard_rows else n))   # (row shards differ in n: a fixed message)
            a.allreduce = cb
            a.exchange_buf = xbuf_ptr
            keep.append(cb)
        if nw:
            nw = comm.max_int(nw)
    if g_init is not None:
        gi = _f64(g_init).ravel()
        if gi.size != m:
            raise HibayesError(1, "g_init must have one entry per marker")
        a.g_init = gi.ctypes.data
        keep.append(gi)
    if warm is not None:
        ws = _lib.WarmState.make(**warm) if isinstance(warm, dict) else warm
        a.warm = ct.addressof(ws)
        keep.append(ws)
    if shard_rows:  # exact cross-check mode: this process holds rows [row_offset, row_offset + n) of every marker (include/hibayes_gpu.h)
        a.shard_rows, a.n_global, a.row_offset = 1, int(n if n_global is None else n_global), int(row_offset)
    a.genotype_bits = int(genotype_bits)  # 0 (default): auto — 2 bits where exact and faster (codes 0..3, BayesB / C at panel 512), int8 otherwise; 8 / 2 force a layout; same chain
    a.sync_blocks = int(sync_every_blocks)  # exchanges per sweep of a sharded run; the chain itself does not depend on it
    if log is not None:
        logcb = _lib.LOG_FN(lambda line, _u: log(line.decode("utf-8", "replace")))
        a.log = logcb
        keep.append(logcb)

    o = BayesOut()
    bufs = {}

    def buf(name, shape, dtype=np.float64):
        arr = np.zeros(shape, dtype=dtype, order="F")
        bufs[name] = arr
        return arr.ctypes.data

    n_lev_cap = n * nr
    if nc:
        o.beta, o.s_beta = buf("beta", nc), buf("s_beta", (nc, nrec))
    o.alpha, o.pi = buf("alpha", m), buf("pi", Pi.size)
    if nr:
        o.Vr, o.s_Vr = buf("Vr", nr), buf("s_Vr", (nr, nrec))
        o.r_est = buf("r_est", n_lev_cap)
        o.r_term_nlevels = buf("r_nlev", nr, np.int32)
        o.s_r = buf("s_r", n_lev_cap * nrec)
    o.g, o.e, o.pip = buf("g", n), buf("e", n), buf("pip", m)
    if nw:
        o.gwas = buf("gwas", nw)
    o.s_Vg, o.s_Ve, o.s_h2, o.s_mu = buf("s_Vg", nrec), buf("s_Ve", nrec), buf("s_h2", nrec), buf("s_mu", nrec)
    o.s_pi = buf("s_pi", (Pi.size, nrec))
    if store_alpha:
        o.s_alpha = buf("s_alpha", (m, nrec))
    o.alpha_sd = buf("alpha_sd", m)
    o.g_last = buf("g_last", m)
    if model == "BayesL":
        o.vargL_last = buf("vargL_last", m)

    check(L.hb_bayes_run(ct.byref(a), ct.byref(o)))

    res = {}
    mc = {}
    if nr:
        res["Vr"] = bufs["Vr"]
        mc["Vr"] = bufs["s_Vr"]
    res["Vg"], res["Ve"], res["h2"] = o.Vg, o.Ve, o.h2
    mc["Vg"], mc["Ve"], mc["h2"] = (bufs["s_Vg"].reshape(1, -1), bufs["s_Ve"].reshape(1, -1),
                                    bufs["s_h2"].reshape(1, -1))
    res["mu"] = o.mu
    mc["mu"] = bufs["s_mu"].reshape(1, -1)
    if nc:
        res["beta"] = bufs["beta"]
        mc["beta"] = bufs["s_beta"]
    res["alpha"] = bufs["alpha"]
    if store_alpha:
        mc["alpha"] = bufs["s_alpha"]
    res["pi"] = bufs["pi"]
    mc["pi"] = bufs["s_pi"]
    if nr:
        nl = o.n_levels
        levels = []
        Rm = np.asarray(R, dtype=object).reshape(n, -1)
        for j in range(nr):
            levels += sorted(set(str(v) for v in Rm[:, j]))
        res["r"] = {"Levels": levels, "Estimation": bufs["r_est"][:nl].copy()}
        mc["r"] = bufs["s_r"][: nl * nrec].reshape((nl, nrec), order="F")
    res["g"], res["e"], res["pip"] = bufs["g"], bufs["e"], bufs["pip"]
    if nw:
        res["gwas"] = bufs["gwas"]
    res["MCMCsamples"] = mc
    # the chain after its last iteration: Bayes(..., g_init=last["g"], warm=last["warm"]) continues it (hb_warm_state)
    lw = o.last.as_dict(Pi.size)
    if model == "BayesL":
        lw["vargL"] = bufs["vargL_last"]
    res["last"] = {"g": bufs["g_last"], "warm": lw}
    res["alpha_sd"] = bufs["alpha_sd"]
    res["timing"] = {"setup_seconds": o.setup_seconds, "loop_seconds": o.loop_seconds,
                     "iters_done": o.iters_done, "mean_events": o.mean_events,
                     "sweeps_replayed": o.sweeps_replayed, "resident_bits": o.resident_bits}
    res["nzct"], res["n_records"] = o.nzct, o.n_records
    del keep
    return res


# --------------------------------------------------------------------------------------------
# ibrm(): formula handling, R/bayes.r:151-320
# --------------------------------------------------------------------------------------------
_RAND = re.compile(r"\(\s*1\s*\|\s*([:\w\d.]+)\s*\)")


def _isna(v):
    if v is None:
        return True
    if isinstance(v, float) and np.isnan(v):
        return True
    return isinstance(v, str) and v in ("NA", "")


def _column(data, name):
    if hasattr(data, "columns"):  # pandas
        return list(data[name])
    return list(data[name])


def _first_column_name(data):
    if hasattr(data, "columns"):
        return list(data.columns)[0]
    return next(iter(data))


def _model_matrix(cols, names, rows):
    """R's model.matrix() minus the intercept for numeric and factor main effects:
    numeric columns as they are, character columns as treatment contrasts over the sorted
    levels present (first level dropped). R/bayes.r:205-207."""
    out, labels = [], []
    for nm in names:
        vals = [cols[nm][i] for i in rows]
        try:
            num = np.array([float(v) for v in vals], dtype=np.float64)
            out.append(num)
            labels.append(nm)
        except (TypeError, ValueError):
            lev = sorted(set(str(v) for v in vals))
            for l in lev[1:]:
                out.append(np.array([1.0 if str(v) == l else 0.0 for v in vals]))
                labels.append(nm + l)
    if not out:
        return None, []
    Xm = np.column_stack(out)
    keepc = [j for j in range(Xm.shape[1]) if not np.all(Xm[:, j] == 1)]  # :206
    if not keepc:
        return None, []
    return np.asfortranarray(Xm[:, keepc]), [labels[j] for j in keepc]


def _map_columns(map):
    """Chromosome and position of the `map` argument of ibrm(), validated as R/bayes.r:217-246 does: columns 2 and 3 of a
    table (SNP, chr, pos), or the dict read_plink() returns (keys Chr/Pos or chr/pos). Non-numeric chromosome labels
    (X, Y, MT ...) are numbered after the largest numeric one, in order of appearance."""
    if map is None:
        raise ValueError("map information must be provided.")
    if isinstance(map, dict):
        ck = next((k for k in ("Chr", "chr", "CHROM", "chrom") if k in map), None)
        pk = next((k for k in ("Pos_text", "Pos", "pos", "POS") if k in map), None)
        if ck is None or pk is None:
            raise ValueError("At least 3 columns in map.")
        chr_raw, pos_raw = list(map[ck]), list(map[pk])
    else:
        arr = np.asarray(map, dtype=object)
        if arr.ndim != 2 or arr.shape[1] < 3:
            raise ValueError("At least 3 columns in map.")
        chr_raw, pos_raw = list(arr[:, 1]), list(arr[:, 2])

    def isna(v):
        return v is None or (isinstance(v, float) and np.isnan(v)) or (isinstance(v, str) and v in ("NA", ""))

    def tonum(v):
        try:
            return float(v)
        except (TypeError, ValueError):
            return None

    if any(isna(v) for v in chr_raw):
        raise ValueError("NAs are not allowed in chromosome.")
    if any(tonum(v) == 0 for v in chr_raw):
        raise ValueError("0 is not allowed in chromosome.")
    if any(isna(v) for v in pos_raw):
        raise ValueError("NAs are not allowed in physical position.")
    if any(tonum(v) == 0 for v in pos_raw):
        raise ValueError("0 is not allowed in physical position.")
    pos = [tonum(v) for v in pos_raw]
    if any(v is None for v in pos):
        raise ValueError("Characters are not allowed in physical position.")
    cnum = [tonum(v) for v in chr_raw]
    known = [v for v in cnum if v is not None]
    max_chr = max(known) if known else 0
    extra = {}
    for v, c in zip(chr_raw, cnum):   # :237-243: labels outside 0..max.chr get max.chr + 1, + 2, ... by first appearance
        if c is None or c != int(c) or c < 0:
            if str(v) not in extra:
                extra[str(v)] = max_chr + len(extra) + 1
    chrom = np.array([extra[str(v)] if str(v) in extra else c for v, c in zip(chr_raw, cnum)], dtype=np.float64)
    return chrom, np.array(pos, dtype=np.float64)


def ibrm(formula, data=None, M=None, M_id=None, method="BayesCpi", map=None, Pi=None, fold=None,
         niter=None, nburn=None, thin=5, windsize=None, windnum=None, dfvr=None, s2vr=None, vg=None,
         dfvg=None, s2vg=None, ve=None, dfve=None, s2ve=None, printfreq=100, seed=666666,
         threads=4, verbose=True, *, windindx=None, device=0, panel=0, precise=2,
         store_alpha=True, comm=None, m_global=None, m_offset=0, gebv_samples=None):
    """Mirror of ibrm() (reference R/bayes.r:121-320). `formula` is a string such as
    "T1 ~ 1" or "T1 ~ season + bwt + (1 | loc) + (1 | dam)"; `data` a dict of columns or a
    pandas DataFrame whose first column holds the individual ids; `M` the n_all x m genotype
    matrix (int8 preferred) with row ids `M_id`.
    Sharded runs (`comm` with world > 1): M holds this rank's contiguous marker range [m_offset, m_offset + M.shape[1]) of
    m_global markers (hibayes_amd.dist.shard_range); `map` / `windindx` likewise cover the local markers only, and the GEBV
    partial products of the shards are summed over the ranks.
    `gebv_samples` (default: store_alpha): also return MCMCsamples["g"] = M %*% MCMCsamples$alpha (n_all x n_records,
    R/bayes.r:303-305), computed on the device."""
    if data is None:
        raise ValueError("no data assigned.")
    if M is None:
        raise ValueError("no genotype data.")
    if M_id is None:
        raise ValueError("please assign the individuals id to 'M.id'.")
    M = np.asarray(M)
    M_id = [str(v) for v in M_id]
    if len(M_id) != M.shape[0]:
        raise ValueError("number of individuals mismatched in 'M' and 'M.id'.")
    if method not in METHODS:
        raise ValueError("'arg' should be one of " + ", ".join(METHODS))
    idcol = _first_column_name(data)
    ids = [str(v) for v in _column(data, idcol)]
    pos = {}
    for i, v in enumerate(ids):
        pos.setdefault(v, i)
    if not any(v in pos for v in M_id):
        raise ValueError("no shared individuals between 'M.id' and the first column in 'data'.")
    match = [pos.get(v, -1) for v in M_id]  # data[match(M.id, data[,1]), ]   (:165)

    lhs, rhs = [s.strip() for s in formula.split("~", 1)]
    rand_terms = _RAND.findall(rhs)
    fixed = _RAND.sub("", rhs)
    fixed_terms = [t.strip() for t in re.split(r"\+", fixed) if t.strip() and t.strip() != "1"]
    for t in fixed_terms:
        if "|" in t or "(" in t:
            raise ValueError("Invalid random effects expression '%s',\n  it should be in the format "
                             "'(1 | x)' or '+ (1 | x1:x2:...:xn)'." % t)
        if ":" in t or "*" in t:
            raise NotImplementedError("interaction terms in the fixed part are not supported")
    names = set([lhs] + fixed_terms + [c for r in rand_terms for c in r.split(":")])
    cols = {}
    for nm in names:
        src = _column(data, nm)
        cols[nm] = [src[j] if j >= 0 else None for j in match]
    nall = len(M_id)
    yNA = np.zeros(nall, dtype=bool)  # :199-202
    for nm in names:
        yNA |= np.array([_isna(v) for v in cols[nm]])
    if yNA.all():
        raise ValueError("no effective data left.")
    rows = np.flatnonzero(~yNA)
    Xfix, fixed_names = _model_matrix(cols, fixed_terms, rows)
    R = None
    if rand_terms:
        Rc = []
        for r in rand_terms:
            parts = r.split(":")
            Rc.append([":".join(str(cols[p][i]) for p in parts) for i in rows])
        R = np.array(Rc, dtype=object).T
    if niter is None:
        niter = 50000 if method == "BayesR" else 20000  # :264-266
    if nburn is None:
        nburn = 30000 if method == "BayesR" else 12000
    if thin >= (niter - nburn):
        raise ValueError("bad setting for collecting frequency 'thin'.")
    if printfreq <= 0:
        verbose = False
    if Pi is None:  # :272-279
        if method == "BayesR":
            Pi = [0.95, 0.02, 0.02, 0.01]
            if fold is None:
                fold = [0, 0.0001, 0.001, 0.01]
        else:
            Pi = [0.95, 0.05]
    if (windsize is not None or windnum is not None) and windindx is None:
        if method in ("BayesA", "BayesRR", "BayesL"):
            raise ValueError("can not implement GWAS analysis for the method: " + method)
        if comm is not None and comm.world > 1:
            # window ids are GLOBAL in a sharded run (the shards' window counts are summed by id, reference src/Bayes.cpp:836-843):
            # windows cut from this rank's part of the map would number 1..nw_local on every rank and be added up under the
            # same ids. Cut them on the global map (cutwind_by_bp / cutwind_by_num) and pass this rank's slice as windindx.
            raise ValueError("sharded run: pass windindx (window ids cut on the GLOBAL map, this rank's slice) instead of windsize / windnum")
        chrom, bp = _map_columns(map)          # R/bayes.r:216-246
        from .windows import cutwind_by_bp, cutwind_by_num
        if windnum is not None:
            if len(chrom) < windnum:
                raise ValueError("Number of markers specified in a window is larger than the total number of markers.")
            windindx = cutwind_by_num(chrom, bp, windnum)
        else:
            if bp.max() < windsize:
                raise ValueError("Maximum of physical position is smaller than wind size.")
            windindx = cutwind_by_bp(chrom, bp, windsize)
    y = np.array([float(cols[lhs][i]) for i in rows])
    Mfit = M[rows, :]
    res = Bayes(y=y, X=Mfit, model=method, Pi=Pi, fold=fold, C_=Xfix, R=R, niter=niter, nburn=nburn,
                thin=thin, windindx=windindx, dfvr=dfvr, s2vr=s2vr, vg=vg, dfvg=dfvg, s2vg=s2vg, ve=ve,
                dfve=dfve, s2ve=s2ve, outfreq=printfreq, threads=threads, verbose=verbose, seed=seed,
                device=device, panel=panel, precise=precise, store_alpha=store_alpha, comm=comm,
                m_global=m_global, m_offset=m_offset)
    if "beta" in res:
        res["beta_names"] = fixed_names
    if "Vr" in res:
        res["Vr_names"] = list(rand_terms)
    # GEBV, R/bayes.r:303-308: MCMCsamples$g = M %*% MCMCsamples$alpha over ALL genotyped individuals, g$gebv = its row means.
    # On the device (hb_ctx_matmul: int8 genotypes x fp64 effects, eight records per pass over the non-zero columns).
    if gebv_samples is None:
        gebv_samples = bool(store_alpha)
    from .engine import Context
    with Context(nall, M.shape[1], device=device, panel=panel) as cg:
        cg.upload(M)
        if gebv_samples and "alpha" in res["MCMCsamples"]:
            gs = cg.matmul(res["MCMCsamples"]["alpha"])
            if comm is not None and comm.world > 1:
                gs = comm.sum_array(gs)
            res["MCMCsamples"]["g"] = gs
            gebv = gs.mean(axis=1)
        else:  # rowMeans(M %*% samples) == M %*% rowMeans(samples) up to rounding
            gebv = cg.matmul(res["alpha"])[:, 0]
            if comm is not None and comm.world > 1:
                gebv = comm.sum_array(gebv)
    res["u_last"] = res["g"]
    res["g"] = {"id": list(M_id), "gebv": gebv}
    res["e"] = {"id": [M_id[i] for i in rows], "e": res["e"]}
    res["call"] = "%s ~ %s + M" % (lhs, rhs)
    res["model"] = "Individual level Bayesian model fit by [%s]" % method
    return res
