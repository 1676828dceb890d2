"""Where the chain workgroup's time goes: cycle stamps of k_chain_persist (hb_ctx_set_profiling bit 1) for one sweep in the
stationary regime. python tools/chain_timeline.py [model] [burn] [n m]"""
import os, sys, ctypes as ct
# the stamps are compiled in only with -DHB_STAMPS=1:  tools/build_variant.sh stamps "-DHB_STAMPS=1"
_v = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hibayes_amd", "variants", "stamps.so")
if "HIBAYES_GPU_LIB" not in os.environ and os.path.exists(_v):
    os.environ["HIBAYES_GPU_LIB"] = _v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import BayesArgs, check, RunInfo
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B

model = sys.argv[1] if len(sys.argv) > 1 else "BayesCpi"
burn = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
m = int(sys.argv[4]) if len(sys.argv) > 4 else 500000
L = H.lib()
ctx = H.Context(n, m, seed=20240901)
ctx.generate(20240901, 1000)
class A: pass
y = B.synth_phenotype(ctx, n, m, 0, m, 20240901, None, model)
geo = B.PIPELINE[model]
ctx.set_pipeline(*geo)
ctx.build_gram()
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
run = ct.c_void_p(); check(L.hb_run_create(ct.byref(a), ct.byref(run)))
fin = ct.c_int32()
check(L.hb_run_step(run, burn, ct.byref(fin)))
alone = bool(os.environ.get("CT_ALONE"))  # HB_CHAIN_ALONE timing diagnostic for the stamped sweep only
if alone:
    os.environ["HB_CHAIN_ALONE"] = "1"
ctx.set_profiling(2)
check(L.hb_run_step(run, 1 if alone else 3, ct.byref(fin)))
P = ctx.panel; npan = (m + P - 1) // P
st = np.zeros(32 * npan, dtype=np.int64)
L.hb_ctx_debug_stamps.argtypes = [ct.c_void_p, ct.c_void_p]
check(L.hb_ctx_debug_stamps(ctx.h, st.ctypes.data))
st = st.reshape(npan, 32)
info = RunInfo(); check(L.hb_run_state(run, ct.byref(info)))
nev = st[:, 10]
tot = st[1:, 0] - st[:-1, 0]                     # panel-to-panel period
tot = np.append(tot, st[-1, 6] - st[-1, 0])
print("%s: %d panels, %d moves in the stamped sweep; sweep chain span %.3f ms at 100 MHz wall? (cycles %d)" % (
    model, npan, nev.sum(), 0, st[-1, 6] - st[0, 0]))
cyc = float(st[-1, 6] - st[0, 0])
for name, sel in (("quiet (no candidate)", (st[:, 12] == 0)), ("candidates, no move", (st[:, 12] != 0) & (nev == 0)),
                  ("1-2 moves", (nev >= 1) & (nev <= 2)), ("3-9 moves", (nev >= 3) & (nev <= 9)), ("10-39", (nev >= 10) & (nev < 40)), (">=40", nev >= 40)):
    k = sel.sum()
    if not k:
        continue
    print("  %-22s panels %5d  period avg %8.0f cyc  share of sweep %5.1f %%  waited for dots in %4.1f %%" % (
        name, k, tot[sel].mean(), 100 * tot[sel].sum() / cyc, 100 * st[sel, 11].mean()))
    s = st[sel]
    seg = lambda a, b: np.where((s[:, a] > 0) & (s[:, b] > 0), s[:, b] - s[:, a], 0).mean()
    print("       open->barrier %6.0f | prefetch issue %6.0f | compaction %6.0f | serial pass %6.0f | apply %6.0f | rounds total %6.0f | publish+results %6.0f | forward %6.0f | group publish %6.0f | dma issue %6.0f" % (
        seg(0, 1), seg(1, 7), seg(7, 12), seg(12, 13), seg(13, 14), seg(7, 2), seg(3, 5), seg(5, 8), seg(8, 9), seg(9, 6)))
print("moves/sweep %.0f, misses %.0f" % (info.mean_events, info.mean_misses))
