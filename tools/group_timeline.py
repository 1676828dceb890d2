"""Where k_chain_group's time goes: cycle stamps per mat-vec group for one sweep in the stationary regime.
   tools/build_variant.sh stamps "-DHB_STAMPS=1"; python tools/group_timeline.py [model] [burn] [n m]"""
import os, sys, ctypes as ct
_v = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "variants", "stamps.so")
if "HIBAYES_GPU_LIB" not in os.environ and os.path.exists(_v):
    os.environ["HIBAYES_GPU_LIB"] = _v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import BayesArgs, check, RunInfo
import bench as B

model = sys.argv[1] if len(sys.argv) > 1 else "BayesCpi"
burn = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
m = int(sys.argv[4]) if len(sys.argv) > 4 else 500000
L = H.lib()
ctx = H.Context(n, m, seed=20240901)
ctx.generate(20240901, 1000)
y = B.synth_phenotype(ctx, n, m, 0, m, 20240901, None, model)
geo = B.PIPELINE[model]
ctx.set_pipeline(*geo)
ctx.build_gram()
ctx.set_adaptive(True)
if os.environ.get("GT_BITS") == "2":
    ctx.set_layout(2, keep_int8=False)
Pi, fold = B.prior(model)
a = BayesArgs()
a.n, a.m = n, m
yv = np.ascontiguousarray(y); a.y = yv.ctypes.data
a.model = model.encode()
pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
if fold is not None:
    fv = np.array(fold); a.fold, a.n_fold = fv.ctypes.data, fv.size
a.niter, a.nburn, a.thin = burn + 20, 0, 5
a.seed, a.precise, a.ctx = 20240901, 2, ctx.h
_keep = []
if os.environ.get("GT_STATE"):   # start from a stored chain state (bench.py --save-state): the converged regime without its 2 500-sweep burn-in
    from hibayes_amd._lib import WarmState
    z = np.load(os.environ["GT_STATE"])
    g0 = np.zeros(m); g0[z["idx"]] = z["val"]
    w0 = WarmState.make(float(z["mu"]), float(z["vare"]), float(z["varg"]), [float(x) for x in z["pi"]])
    a.g_init, a.warm = g0.ctypes.data, ct.addressof(w0)
    _keep += [g0, w0]
run = ct.c_void_p(); check(L.hb_run_create(ct.byref(a), ct.byref(run)))
fin = ct.c_int32()
check(L.hb_run_step(run, burn, ct.byref(fin)))
ctx.set_profiling(2)
check(L.hb_run_step(run, 3, ct.byref(fin)))
P = ctx.panel; npan = (m + P - 1) // P
D = ctx.pipeline()[2]
ng = (npan + D - 1) // D
st = np.zeros(32 * npan, dtype=np.int64)
L.hb_ctx_debug_stamps.argtypes = [ct.c_void_p, ct.c_void_p]
check(L.hb_ctx_debug_stamps(ctx.h, st.ctypes.data))
st = st.reshape(npan, 32)[:ng]
info = RunInfo(); check(L.hb_run_state(run, ct.byref(info)))
nmv, cand, rounds, redo, rep = st[:, 10], st[:, 12], st[:, 13], st[:, 14], st[:, 15]
per = np.append(st[1:, 16] - st[:-1, 16], st[-1, 17] - st[-1, 16])
cyc = float(st[-1, 17] - st[0, 16])
names = ["dots (waiting)", "ranking", "exact data", "gram gather", "serial pass", "fold+verify", "commit+publish", "forward", "end of group", "certificate"]
print("%s geometry %s: %d groups, %d moves, %d candidates (first rounds), %d committed rounds, %d rolled back after a full fold, %d repeated by the certificate before the fold; chain span %d cycles" % (
    model, ctx.pipeline(), ng, nmv.sum(), cand.sum(), rounds.sum(), redo.sum(), rep.sum(), cyc))
print("  whole sweep, cycles by phase (accumulated over all rounds): " + " | ".join("%s %d" % (names[k], st[:, k].sum()) for k in range(10)))
for name, sel in (("quiet", cand == 0), ("candidates, no move", (cand > 0) & (nmv == 0)), ("1-4 moves", (nmv >= 1) & (nmv <= 4)),
                  ("5-12 moves", (nmv >= 5) & (nmv <= 12)), (">12 moves", nmv > 12)):
    k = sel.sum()
    if not k:
        continue
    s = st[sel]
    print("  %-20s groups %4d period %7.0f cyc (%.1f %% of the sweep) waited for dots %3.0f %% committed rounds %.2f rolled back %.2f repeated %.2f cand %.1f" % (
        name, k, per[sel].mean(), 100 * per[sel].sum() / cyc, 100 * (s[:, 11] > 0).mean(), rounds[sel].mean(), redo[sel].mean(), rep[sel].mean(), cand[sel].mean()))
    print("      " + " | ".join("%s %.0f" % (names[q], s[:, q].mean()) for q in range(10)))
    if s[:, 21:24].any():
        print("      inside the serial phase, cycles from its start (accumulated over the passes): data in registers %.0f | loop done %.0f | wave 0 at the barrier %.0f | barrier passed %.0f; speculated blocks run again %.2f" % (s[:, 21].mean(), s[:, 22].mean(), s[:, 23].mean(), s[:, 4].mean(), s[:, 24].mean()))
    if s[:, 18:21].any():
        print("      inside the opening, cycles from its start: prefetch barrier passed %.0f | values in registers %.0f | polled (if at all) %.0f" % tuple(s[:, q].mean() for q in (18, 19, 20)))
print("moves/sweep %.0f" % info.mean_events)
busy = st[:, 1:10].sum(axis=1)   # the group's work once its dots are in hand
opening = st[:, 0]
q = lambda a: "mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (a.mean(), *np.percentile(a, [50, 90, 99]), a.max())
print("busy cycles per group:   ", q(busy))
print("opening (dots, waiting): ", q(opening))
print("period:                  ", q(per))
print("busy / span %.3f" % (busy.sum() / cyc))
