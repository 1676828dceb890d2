"""Development aid: long runs for stability (no device-side time-outs, sane posterior) — config 2 (BayesCpi n=10k m=100k, 5000 iterations)
and a few thousand sweeps at config-3 size; `soak.py dense`: BayesRR / A / L at config-3 size."""
import sys, os, time, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import check, BayesArgs, RunInfo, BayesOut
import bench

def run(n, m, model, niter, nburn, seed=20240901, bits=8):
    c = H.Context(n, m); c.set_pipeline(*bench.PIPELINE.get(model, (1, 1, 1))); c.generate(seed, 1000)
    if bits == 2: c.build_gram(); c.set_layout(2, keep_int8=False)  # (the headline's layout: k_dotq2m beside the certified group chain)
    y = bench.synth_phenotype(c, n, m, 0, m, seed, None, model)
    Pi, fold = bench.prior(model)
    a = BayesArgs(); a.n, a.m = n, m; yv = np.ascontiguousarray(y); a.y = yv.ctypes.data; a.model = model.encode()
    pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
    if fold is not None: fv = np.array(fold, dtype=float); a.fold, a.n_fold = fv.ctypes.data, fv.size
    a.niter, a.nburn, a.thin = niter, nburn, 5; a.seed = 1; a.ctx = c.h
    a.precise = int(os.environ.get("HB_SOAK_PRECISE", "2"))  # (the library default: the exact fixed-point mat-vec, update rows riding in the launches;
    #  until round 4 this script left the field at 0 and so soaked the fp32 path, whose dense update is a kernel of its own)
    run = ct.c_void_p(); check(c.L.hb_run_create(ct.byref(a), ct.byref(run)))
    fin = ct.c_int32(); info = RunInfo()
    t0 = time.time(); done = 0
    while done < niter:
        k = min(500, niter - done); check(c.L.hb_run_step(run, k, ct.byref(fin))); done += k
        check(c.L.hb_run_state(run, ct.byref(info)))
        print("  %s n=%d m=%d iter %5d: %.2f ms/sweep so far, nnz %d, vara %.3f vare %.3f redo/sweep %.1f, sweeps replayed %d" % (
            model, n, m, info.iter, (time.time() - t0) / done * 1e3, info.nnz, info.vara, info.vare, info.mean_redo, info.sweeps_replayed), flush=True)
    r, u = c.get_residual()
    print("  done: %d sweeps, %d replayed after a device time-out; h2-ish %.3f, max|r+u-(y-mu)| %.2e" % (
        done, info.sweeps_replayed, info.vara / (info.vara + info.vare), np.max(np.abs(r + u - (y - info.mu)))), flush=True)
    c.L.hb_run_destroy(run); c.close()
    return info.sweeps_replayed

if len(sys.argv) > 1 and sys.argv[1] == "dense":  # the models in which every marker moves (k_chain_dense, k_fold_dense)
    # soak.py dense [all|rr|a|l] [sweeps]: "all" = BayesRR / A / L at config-3 size + a ragged BayesRR; a run that loses a sweep to a
    # device time-out replays it (hb_run_step) — the counts of replayed sweeps and of FAILED runs are the result
    nfail = nrep = ntot = 0
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    cases = {"all": (("BayesRR", 50000, 500000, k), ("BayesA", 50000, 500000, k // 2), ("BayesL", 50000, 500000, k // 2),
                     ("BayesRR", 20000, 100000 + 300, k)),  # (the last: ragged last panel)
             "rr": (("BayesRR", 50000, 500000, k),), "a": (("BayesA", 50000, 500000, k),), "l": (("BayesL", 50000, 500000, k),)}[which]
    for model, n, m, it in cases:
        try:
            nrep += run(n, m, model, it, min(500, it // 2))
            ntot += it
        except Exception as e:  # a run that could not be completed
            nfail += 1
            print("  FAILED %s n=%d m=%d: %s" % (model, n, m, e), flush=True)
    print("dense soak: %d of %d runs failed; %d sweeps completed, %d of them replayed after a device time-out" % (nfail, len(cases), ntot, nrep), flush=True)
    sys.exit(1 if nfail else 0)
elif len(sys.argv) > 1 and sys.argv[1] == "cpi":  # soak.py cpi [sweeps]: the headline — BayesCpi at config-3 size on the 2-bit layout (k_chain_group certified + k_fwd + k_dotq2m)
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    rep = run(50000, 500000, "BayesCpi", k, min(500, k // 2), bits=2)
    print("BayesCpi soak (2-bit): %d sweeps completed, %d of them replayed after a device time-out" % (k, rep), flush=True)
elif len(sys.argv) > 1 and sys.argv[1] == "bayesr":  # soak.py bayesr [sweeps]: config 3's model on its own (k_chain_persist + k_fwd + k_warm)
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    rep = run(50000, 500000, "BayesR", k, min(500, k // 2))
    print("BayesR soak: %d sweeps completed, %d of them replayed after a device time-out" % (k, rep), flush=True)
else:
    run(10000, 100000, "BayesCpi", 5000, 2500)
    run(50000, 500000, "BayesCpi", 3000, 1500)
    run(50000, 500000, "BayesR", 600, 300)
