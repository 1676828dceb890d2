"""Development aid: cycle stamps inside k_chain, per panel (not part of the product or tests)."""
import sys, os, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import check
n, m = int(sys.argv[1]), int(sys.argv[2]); model = sys.argv[3] if len(sys.argv) > 3 else "BayesR"
panel = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nsw = int(sys.argv[5]) if len(sys.argv) > 5 else 30
c = H.Context(n, m, panel=panel); c.generate(20240901, 1000)
rng = np.random.default_rng(3)
beta = np.zeros(m); idx = rng.choice(m, max(1, m // 1000), replace=False); beta[idx] = rng.normal(0, 0.05, idx.size)
xb = np.zeros(n); check(c.L.hb_ctx_matvec(c.h, beta.ctypes.data, xb.ctypes.data))
y = xb - xb.mean(); y = y * np.sqrt(0.5 / y.var()) + rng.normal(0, np.sqrt(0.5), n)
c.set_profiling(3 if os.environ.get('HB_PIPELINE','1')=='0' else 2)
Pi, fold = ([0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2]) if model == "BayesR" else ([0.95, 0.05], None)
from hibayes_amd._lib import BayesArgs, RunInfo
a = BayesArgs(); a.n, a.m = n, m; yv = np.ascontiguousarray(y); a.y = yv.ctypes.data; a.model = model.encode()
pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
if fold: fv = np.array(fold, dtype=float); a.fold, a.n_fold = fv.ctypes.data, fv.size
a.niter, a.nburn, a.thin = nsw + 5, 0, 5; a.seed = 1; a.ctx = c.h
run = ct.c_void_p(); check(c.L.hb_run_create(ct.byref(a), ct.byref(run)))
fin = ct.c_int32(); check(c.L.hb_run_step(run, nsw, ct.byref(fin)))
tm = c.last_timing(); info = RunInfo(); check(c.L.hb_run_state(run, ct.byref(info)))
P = c.panel; npan = (m + P - 1) // P
st = np.zeros((npan, 32), dtype=np.int64)
c.L.hb_ctx_debug_stamps.argtypes = [ct.c_void_p, ct.c_void_p]; check(c.L.hb_ctx_debug_stamps(c.h, st.ctypes.data))
S = P // 64
d = st - st[:, :1]
print("panel", P, "timing(ms/sweep)", {k: round(v, 3) for k, v in tm.items()}, "events/sweep", info.mean_events, "nnz", info.nnz)
if os.environ.get('HB_PIPELINE','1')!='0':
    dd = np.diff(st[5:, :7], axis=1); per = np.diff(st[5:, 0])
    print("persist: median cycles per phase [take,prefetch-issue,turns,tail,publish,results,fwd]:", [int(np.median(dd[:, i])) for i in range(6)], "per panel total", int(np.median(per)), "-> us", np.median(per)/2100.)
print("cycles (100MHz? shader clk) medians: staged+coef", np.median(d[:, 1]), "turns", [int(np.median(d[:, 2 + s] - d[:, 1 + s])) for s in range(min(S, 24))],
      "loop_end", np.median(d[:, 26]), "end", np.median(d[:, 27]))
print("per panel avg us: dot %.2f chain %.2f update %.2f" % (tm["dot_ms"] / npan * 1e3, tm["chain_ms"] / npan * 1e3, tm["update_ms"] / npan * 1e3))
