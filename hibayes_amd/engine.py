"""Context: thin object wrapper over the fine-grained C ABI (hb_ctx_*), used by bench.py and by
the kernel-level parity tests. Everything computes on the device; see include/hibayes_gpu.h."""
import ctypes as C

import numpy as np

from ._lib import CtxParams, LaunchStats, SweepIn, SweepOut, SweepTiming, HB_MAX_FOLD, check, lib

MODEL_INDEX = {"BayesRR": 1, "BayesA": 2, "BayesB": 3, "BayesBpi": 3, "BayesC": 4, "BayesCpi": 4,
               "BayesL": 5, "BayesR": 6}


class Context:
    def __init__(self, n, m, device=0, panel=0, precise=2, m_offset=0, seed=666666):
        self.L = lib()
        p = CtxParams(device=device, n=n, m=m, panel=panel, precise=int(precise), m_offset=m_offset, seed=seed)
        h = C.c_void_p()
        check(self.L.hb_ctx_create(C.byref(p), C.byref(h)))
        self.h, self.n, self.m = h, n, m

    def close(self):
        if self.h:
            self.L.hb_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def panel(self):
        return self.L.hb_ctx_panel(self.h)

    @property
    def ld(self):
        return self.L.hb_ctx_ld(self.h)

    # ---- genotypes ----
    def upload(self, X, col0=0):
        X = np.asarray(X)
        if X.dtype == np.int8:
            Xa = np.asfortranarray(X)
            check(self.L.hb_ctx_upload_genotype_i8(self.h, Xa.ctypes.data, Xa.strides[1], col0, Xa.shape[1]))
        else:
            Xa = np.asfortranarray(X, dtype=np.float64)
            check(self.L.hb_ctx_upload_genotype_f64(self.h, Xa.ctypes.data, Xa.strides[1] // 8, col0, Xa.shape[1]))

    def upload_bed(self, raw, nind, rows=None, col0=0, ncols=None):
        raw = np.frombuffer(raw, dtype=np.uint8)
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
        check(self.L.hb_ctx_upload_bed(self.h, raw.ctypes.data, raw.size, nind, None if r is None else r.ctypes.data,
                                       col0, self.m if ncols is None else ncols))

    def generate(self, seed, mono_every=0):
        check(self.L.hb_ctx_generate_genotype(self.h, seed, mono_every))

    def download(self, col0=0, ncols=None):
        ncols = self.m - col0 if ncols is None else ncols
        out = np.zeros((self.n, ncols), dtype=np.int8, order="F")
        check(self.L.hb_ctx_download_genotype(self.h, out.ctypes.data, self.n, col0, ncols))
        return out

    def marker_stats(self):
        xpx, vx = np.zeros(self.m), np.zeros(self.m)
        s, z = C.c_double(), C.c_int32()
        check(self.L.hb_ctx_marker_stats(self.h, xpx.ctypes.data, vx.ctypes.data, C.byref(s), C.byref(z)))
        return xpx, vx, s.value, z.value

    def set_layout(self, bits=2, keep_int8=True):
        """Resident genotype layout the sweep reads: 8 (int8 columns) or 2 (2 bits per genotype, a quarter of the bytes)."""
        check(self.L.hb_ctx_set_layout(self.h, int(bits), 1 if keep_int8 else 0))

    def layout(self):
        b, k = C.c_int32(), C.c_int32()
        check(self.L.hb_ctx_get_layout(self.h, C.byref(b), C.byref(k)))
        return b.value, bool(k.value)

    def set_adaptive(self, on=True):
        """Let Bayes() choose the geometry of each sweep of a point-mass model from the number of moves of the previous one."""
        check(self.L.hb_ctx_set_adaptive(self.h, 1 if on else 0))

    def set_pipeline(self, pipeline=1, lookahead=2, dotgroup=4):
        check(self.L.hb_ctx_set_pipeline(self.h, pipeline, lookahead, dotgroup))

    def set_matvec_kernel(self, kind):
        """2-bit resident genotypes: 0 = k_dotq2 (v_dot4, default), 1 = k_dotq2r, 2 = k_dotq2m (matrix cores; an A/B). Same integers."""
        check(self.L.hb_ctx_set_matvec_kernel(self.h, kind))

    def debug_inject_abort(self, panel, times=1):
        """Debug hook: abort the next `times` pipeline sweeps once the chain has published `panel` panels (hb_run_step replays them)."""
        check(self.L.hb_ctx_debug_inject_abort(self.h, panel, times))

    def build_gram(self):
        s = C.c_double()
        check(self.L.hb_ctx_build_gram(self.h, C.byref(s)))
        return s.value

    def gram(self, p):
        P = self.panel
        G = np.zeros((P, P), dtype=np.int32)
        check(self.L.hb_ctx_download_gram(self.h, p, G.ctypes.data))
        return G

    def pipeline_note(self):
        v = self.L.hb_ctx_pipeline_note(self.h)
        return None if v is None else v.decode()

    def matmul(self, A):
        """X @ A on the device (A: m x R); the GEBV sample matrix of R/bayes.r:303-305."""
        A = np.asfortranarray(A, dtype=np.float64).reshape(self.m, -1, order="F")
        out = np.zeros((self.n, A.shape[1]), order="F")
        check(self.L.hb_ctx_matmul(self.h, A.ctypes.data, self.m, A.shape[1], out.ctypes.data, self.n))
        return out

    def gram_band(self, p, l):
        P = self.panel
        G = np.zeros((P, P), dtype=np.int32)
        check(self.L.hb_ctx_download_gram_band(self.h, p, l, G.ctypes.data))
        return G

    def pipeline(self):
        v = [C.c_int32() for _ in range(4)]
        check(self.L.hb_ctx_get_pipeline(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)   # (pipeline, lookahead groups, panels per mat-vec, band blocks)

    def events(self):
        """Move lists of the last sweep as [(panel-local marker indices, deltas)] per panel."""
        P = self.panel
        npan = (self.m + P - 1) // P
        cnt = np.zeros(npan, dtype=np.int32)
        idx = np.zeros(npan * P, dtype=np.int32)
        dl = np.zeros(npan * P)
        check(self.L.hb_ctx_get_events(self.h, cnt.ctypes.data, idx.ctypes.data, dl.ctypes.data))
        return cnt, [(idx[p * P:p * P + cnt[p]].copy(), dl[p * P:p * P + cnt[p]].copy()) for p in range(npan)]

    # ---- state ----
    def set_residual(self, yadj=None, u=None):
        a = None if yadj is None else np.ascontiguousarray(yadj, dtype=np.float64)
        b = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
        check(self.L.hb_ctx_set_residual(self.h, None if a is None else a.ctypes.data, None if b is None else b.ctypes.data))

    def get_residual(self):
        r, u = np.zeros(self.n), np.zeros(self.n)
        check(self.L.hb_ctx_get_residual(self.h, r.ctypes.data, u.ctypes.data))
        return r, u

    def set_effects(self, g=None, tracker=None, vargL=None):
        ga = None if g is None else np.ascontiguousarray(g, dtype=np.float64)
        ta = None if tracker is None else np.ascontiguousarray(tracker, dtype=np.uint8)
        va = None if vargL is None else np.ascontiguousarray(vargL, dtype=np.float64)
        check(self.L.hb_ctx_set_effects(self.h, *(None if x is None else x.ctypes.data for x in (ga, ta, va))))

    def get_effects(self):
        g, t, v = np.zeros(self.m), np.zeros(self.m, dtype=np.uint8), np.zeros(self.m)
        check(self.L.hb_ctx_get_effects(self.h, g.ctypes.data, t.ctypes.data, v.ctypes.data))
        return g, t, v

    def dot(self, col0=0, ncols=None):
        ncols = self.m - col0 if ncols is None else ncols
        d = np.zeros(ncols)
        check(self.L.hb_ctx_dot(self.h, col0, ncols, d.ctypes.data))
        return d

    def residual_sums(self):
        a, b = C.c_double(), C.c_double()
        check(self.L.hb_ctx_residual_sums(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def residual_shift(self, a):
        check(self.L.hb_ctx_residual_shift(self.h, float(a)))

    def set_covariates(self, Cm):
        Cm = np.asfortranarray(Cm, dtype=np.float64).reshape(self.n, -1, order="F")
        check(self.L.hb_ctx_set_covariates(self.h, Cm.ctypes.data, Cm.shape[1]))

    def cov_dot(self, i):
        v = C.c_double()
        check(self.L.hb_ctx_cov_dot(self.h, i, C.byref(v)))
        return v.value

    def cov_axpy(self, i, a):
        check(self.L.hb_ctx_cov_axpy(self.h, i, float(a)))

    def set_levels(self, zid, nlev):
        z = np.asfortranarray(zid, dtype=np.int32).reshape(self.n, -1, order="F")
        nl = np.ascontiguousarray(nlev, dtype=np.int32)
        self._nlev = list(nl)
        check(self.L.hb_ctx_set_levels(self.h, z.ctypes.data, z.shape[1], nl.ctypes.data))

    def level_sums(self, t):
        s = np.zeros(self._nlev[t])
        check(self.L.hb_ctx_level_sums(self.h, t, s.ctypes.data))
        return s

    def level_axpy(self, t, delta):
        d = np.ascontiguousarray(delta, dtype=np.float64)
        check(self.L.hb_ctx_level_axpy(self.h, t, d.ctypes.data))

    def blocks_setup(self, cpc, zz, vrtmp0):
        """Device-resident covariate / random-effect state (reference src/Bayes.cpp:484-516); after set_covariates / set_levels."""
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (cpc, zz, vrtmp0)]
        self._blk = (len(a[0]), len(a[1]), len(a[2]))
        check(self.L.hb_ctx_blocks_setup(self.h, *[x.ctypes.data for x in a]))

    def blocks_step(self, vare, z_beta, z_levels, chisq, dfr, s2r):
        """Enqueue the iteration's covariate and random-effect kernels with the host's pre-drawn deviates (no host sync)."""
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (z_beta, z_levels, chisq)]
        check(self.L.hb_ctx_blocks_step(self.h, float(vare), *[x.ctypes.data for x in a], float(dfr), float(s2r)))

    def blocks_state(self):
        nc, nl, nr = self._blk
        out = [np.zeros(nc), np.zeros(nl), np.zeros(nr), np.zeros(nr)]
        check(self.L.hb_ctx_blocks_state(self.h, *[x.ctypes.data for x in out]))
        return out

    def set_windows(self, windindx):
        w = np.ascontiguousarray(windindx, dtype=np.uint32)
        self._nw = int(w.max())
        check(self.L.hb_ctx_set_windows(self.h, w.ctypes.data, self._nw))

    def get_windows(self):
        w = np.zeros(self._nw)
        check(self.L.hb_ctx_get_windows(self.h, w.ctypes.data))
        return w

    def counters(self):
        a, b, c = np.zeros(self.m), np.zeros(self.m), np.zeros(self.m)
        check(self.L.hb_ctx_get_counters(self.h, a.ctypes.data, b.ctypes.data, c.ctypes.data))
        return a, b, c

    # ---- one marker sweep ----
    def sweep(self, model, it, vare, varg=0.0, logpi=(0.0, 0.0), fold=(0.0, 0.0), vara_fold=None, s2varg_df=0.0,
              dfvara=4.0, lam=0.0, lam2=0.0, count_pip=False, store=False):
        si = SweepIn()
        si.model_index = MODEL_INDEX[model] if isinstance(model, str) else int(model)
        si.n_fold = len(logpi)
        si.iter = it
        si.vare, si.varg, si.s2varg_df, si.dfvara = vare, varg, s2varg_df, dfvara
        for k in range(len(logpi)):
            si.logpi[k] = logpi[k]
            si.fold[k] = fold[k] if k < len(fold) else 0.0
            si.vara_fold[k] = (vara_fold[k] if vara_fold is not None else varg * si.fold[k])
        si.lambda_, si.lambda2 = lam, lam2
        si.count_pip, si.store = int(count_pip), int(store)
        so = SweepOut()
        check(self.L.hb_ctx_sweep(self.h, C.byref(si), C.byref(so)))
        return {"sum_g2": so.sum_g2, "class_count": np.array(so.class_count[:]), "sum_vargL": so.sum_vargL,
                "sum_r": so.sum_r, "sum_r2": so.sum_r2, "var_u": so.var_u, "n_events": so.n_events}

    def time_matvec(self, reps=3):
        ms, nl, nc = C.c_double(), C.c_int32(), C.c_int32()
        check(self.L.hb_ctx_time_matvec(self.h, reps, C.byref(ms), C.byref(nl), C.byref(nc)))
        return ms.value, nl.value, nc.value

    def time_stream_read(self, reps=3):
        """(ms per pass, bytes per pass) of a plain streaming read of the resident genotype buffer: the measured ceiling the
        mat-vec's achieved bandwidth is quoted against beside the nominal HBM peak."""
        ms, nb = C.c_double(), C.c_int64()
        check(self.L.hb_ctx_time_stream_read(self.h, reps, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def matvec_stamps(self):
        """In-situ statistics of the mat-vec launches of the last sweep (set_profiling(8) first); see hb_launch_stats."""
        st = LaunchStats()
        check(self.L.hb_ctx_matvec_stamps(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in LaunchStats._fields_}

    def set_profiling(self, on):
        check(self.L.hb_ctx_set_profiling(self.h, int(on)))

    def last_timing(self):
        t = SweepTiming()
        check(self.L.hb_ctx_last_timing(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in SweepTiming._fields_}
