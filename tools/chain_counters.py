"""Hardware counters of the persistent chain workgroup (k_chain_persist), which the real pipeline cannot give: under
rocprofv3 --pmc kernels serialise and the pipeline's two branches never meet. Here the chain is brought to its stationary
regime with the event-ordered per-panel kernels (they run anywhere), then two sweeps run in the chain-alone diagnostic mode
(hb_ctx_set_profiling bit 2: mat-vec launches first, the chain afterwards with the device to itself; those two sweeps are not
MCMC). Usage, one counter set per pass (SQ has 8 counters):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS \
        -d out -o chain -- python tools/chain_counters.py [model burn]
    python tools/rocprof_summary.py out/.../chain_results.db"""
import os, sys, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
from hibayes_amd._lib import BayesArgs, check
import bench as B

model = sys.argv[1] if len(sys.argv) > 1 else "BayesCpi"
burn = int(sys.argv[2]) if len(sys.argv) > 2 else 120
n, m = 50000, 500000
L = H.lib()
ctx = H.Context(n, m, seed=20240901)
ctx.generate(20240901, 1000)
y = B.synth_phenotype(ctx, n, m, 0, m, 20240901, None, model)
geo = B.PIPELINE[model]
ctx.set_pipeline(*geo)      # (under a serialising profiler this stays on the per-panel kernels)
ctx.build_gram()
Pi, fold = B.prior(model)
a = BayesArgs()
a.n, a.m = n, m
yv = np.ascontiguousarray(y); a.y = yv.ctypes.data
a.model = model.encode()
pv = np.array(Pi); a.Pi, a.n_pi = pv.ctypes.data, pv.size
if fold is not None:
    fv = np.array(fold); a.fold, a.n_fold = fv.ctypes.data, fv.size
a.niter, a.nburn, a.thin = burn + 2, 0, 1
a.seed, a.precise, a.ctx = 20240901, 2, ctx.h
run = ct.c_void_p(); check(L.hb_run_create(ct.byref(a), ct.byref(run)))
fin = ct.c_int32()
check(L.hb_run_step(run, burn, ct.byref(fin)))
print("burn-in done on", ctx.pipeline(), ctx.pipeline_note())
ctx.set_profiling(4)
ctx.set_pipeline(*geo)
print("chain-alone mode on", ctx.pipeline())
check(L.hb_run_step(run, 2, ct.byref(fin)))
from hibayes_amd._lib import RunInfo
info = RunInfo(); check(L.hb_run_state(run, ct.byref(info)))
print("done: iteration", info.iter, "finished", fin.value, "geometry", ctx.pipeline())
