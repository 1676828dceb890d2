"""Development aid: burn a run in to steady state once, then time sweeps under several pipeline geometries.
usage: geo_sweep.py n m model burn panel "Lv,D Lv,D ..." [sweeps]"""
import sys, os, ctypes as ct, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import check, BayesArgs, RunInfo
import bench
n, m = int(sys.argv[1]), int(sys.argv[2]); model = sys.argv[3]; burn = int(sys.argv[4]); panel = int(sys.argv[5])
geos = [tuple(int(v) for v in g.split(",")) for g in sys.argv[6].split()]
K = int(sys.argv[7]) if len(sys.argv) > 7 else 60
c = H.Context(n, m, panel=panel); c.set_pipeline(1, *geos[0]); c.generate(20240901, 1000)
y = bench.synth_phenotype(c, n, m, 0, m, 20240901, None, model)
Pi, fold = ([0.95, 0.02, 0.02, 0.01], [0, 1e-4, 1e-3, 1e-2]) if model == "BayesR" else ([0.95, 0.05], None)
a = BayesArgs(); a.n, a.m = n, m; yv = np.ascontiguousarray(y); a.y = yv.ctypes.data; a.model = model.encode()
pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
if fold: fv = np.array(fold, dtype=float); a.fold, a.n_fold = fv.ctypes.data, fv.size
a.precise = int(os.environ.get("PRECISE", "2")); a.niter, a.nburn, a.thin = burn + (K + 20) * len(geos) * 12 + 5, 0, 5; a.seed = 1; a.ctx = c.h
run = ct.c_void_p(); check(c.L.hb_run_create(ct.byref(a), ct.byref(run)))
fin = ct.c_int32(); info = RunInfo()
t0 = time.time(); check(c.L.hb_run_step(run, burn, ct.byref(fin))); print("burn %d sweeps %.1f s" % (burn, time.time() - t0), flush=True)
tunes = [tuple(float(v) for v in g.split(",")) for g in os.environ.get("TUNES", "6,0.64").split()]
c.L.hb_ctx_debug_tune.argtypes = [ct.c_void_p, ct.c_double, ct.c_double]
for tune in tunes:
    for geo in geos:
        c.set_pipeline(1, *geo); c.build_gram(); check(c.L.hb_ctx_debug_tune(c.h, *tune)); print("kappa,candf", tune, end=" ")
        check(c.L.hb_run_step(run, 20, ct.byref(fin)))
        check(c.L.hb_run_state(run, ct.byref(info))); e0, m0, i0, r0 = info.mean_events * info.iter, info.mean_misses * info.iter, info.iter, info.mean_redo * info.iter
        t0 = time.time(); check(c.L.hb_run_step(run, K, ct.byref(fin))); dt = time.time() - t0
        check(c.L.hb_run_state(run, ct.byref(info)))
        print("P %d geo %s: %.3f ms/sweep (%.1f sweeps/s) moves %.0f misses %.0f redo %.1f nnz %d" % (
            panel, geo, dt / K * 1e3, K / dt, (info.mean_events * info.iter - e0) / K, (info.mean_misses * info.iter - m0) / K, (info.mean_redo * info.iter - r0) / K, info.nnz), flush=True)
if os.environ.get("STAMPS"):
    P = c.panel; npan = (m + P - 1) // P
    c.set_profiling(2); check(c.L.hb_run_step(run, 3, ct.byref(fin)))
    st = np.zeros((npan, 32), dtype=np.int64)
    c.L.hb_ctx_debug_stamps.argtypes = [ct.c_void_p, ct.c_void_p]; check(c.L.hb_ctx_debug_stamps(c.h, st.ctypes.data))
    dd = np.diff(st[5:-2, :7], axis=1); per = np.diff(st[5:-1, 0])
    names = ["take", "prefetch-issue", "turns", "tail", "publish", "results", "fwd+landing"]
    print("phase mean cycles:", {names[i]: int(dd[:, i].mean()) for i in range(6)}, "loop-top gap", int((per - dd.sum(1)).mean()), "per panel", int(per.mean()), "-> us %.2f" % (per.mean() / 2100.))
    a = st[5:-2]
    print("fine: 0->1 take %d, 1->7 issue+barrier %d, 7->2 rounds %d, 2->3 ticket %d, 3->6 results+fwd+landing %d; moves/panel %.2f; waited-for-matvec frac %.3f" % (
        (a[:, 1] - a[:, 0]).mean(), (a[:, 7] - a[:, 1]).mean(), (a[:, 2] - a[:, 7]).mean(), (a[:, 3] - a[:, 2]).mean(), (a[:, 6] - a[:, 3]).mean(), a[:, 10].mean(), a[:, 11].mean()))
    print("  candidates at the opening %.1f per panel, rounds %.2f, repeats of a speculated block %.2f per panel" % (a[:, 15].mean(), a[:, 16].mean(), a[:, 17].mean()))
    print("  take: ring waves' counted wait %d | barrier that hands over the early pieces %d | reads, sentinel check, filter %d" % ((a[:,20]-a[:,0]).mean(), (a[:,21]-a[:,20]).mean(), (a[:,1]-a[:,21]).mean()))
    print("  arrival of waves 0..7 at that barrier, cycles after wave 0's loop top:", [int((a[:, 22 + w] - a[:, 0]).mean()) for w in range(8)])
    cr = a[:, 15] >= 8
    if cr.sum():
        o = a[cr]
        print("  first round of crowded panels: staged->gathered %d | serial pass %d | barrier+apply %d | violation barrier %d | rest of the rounds %d" % (
            (o[:,18]-o[:,12]).mean(), (o[:,13]-o[:,18]).mean(), (o[:,14]-o[:,13]).mean(), (o[:,19]-o[:,14]).mean(), (o[:,2]-o[:,19]).mean()))
    nm = a[:, 10]
    for label, one in (("1-move", nm == 1), ("16+-move", nm >= 16)):
      if one.sum():
        o = a[one]
        print("  " + label + " panels: take %d | issue+barrierA %d | to staged(B2) %d | chain %d | B3+apply %d | B4..end rounds %d | 2->3 %d | publish(3->4) %d | results(4->5) %d | fwd+landing(5->6) %d" % (
            (o[:,1]-o[:,0]).mean(), (o[:,7]-o[:,1]).mean(), (o[:,12]-o[:,7]).mean(), (o[:,13]-o[:,12]).mean(), (o[:,14]-o[:,13]).mean(), (o[:,2]-o[:,14]).mean(), (o[:,3]-o[:,2]).mean(), (o[:,4]-o[:,3]).mean(), (o[:,5]-o[:,4]).mean(), (o[:,6]-o[:,5]).mean()))
    for lo, hi in ((0, 0), (1, 1), (2, 3), (4, 7), (8, 15), (16, 1000)):
        sel = (nm >= lo) & (nm <= hi)
        if sel.sum(): print("  moves %d-%d: %d panels, rounds phase %d cyc, whole panel %d cyc, take %d" % (lo, hi, sel.sum(), (a[sel, 2] - a[sel, 7]).mean(), per[:len(sel)][sel[:len(per)]].mean(), (a[sel,1]-a[sel,0]).mean()))
    0 and print("phase median cycles:", {names[i]: int(np.median(dd[:, i])) for i in range(6)}, "per panel", int(np.median(per)))
