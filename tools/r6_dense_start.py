"""Where is the ~1 ms at the start of every dense sweep (tools/r6_long_launch.py: launch 2 of 489 lives 1 ms — its update rows wait for the chain)?
k_chain_dense's per-panel stamps (-DHB_STAMPS=1 build) for the first panels of a sweep.  python tools/r6_dense_start.py"""
import os, sys, ctypes as ct
_v = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "variants", "stamps.so")
if "HIBAYES_GPU_LIB" not in os.environ and os.path.exists(_v):
    os.environ["HIBAYES_GPU_LIB"] = _v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
import bench as B
from hibayes_amd._lib import check
n, m = 50000, 500000
with H.Context(n, m, panel=512, seed=20240901) as c:
    c.generate(20240901, 1000)
    y = B.synth_phenotype(c, n, m, 0, m, 20240901, None, "BayesCpi")
    c.set_pipeline(1, 2, 2)
    c.build_gram()
    vare, varg = 0.5, 0.5 / (0.5 * m)
    c.set_effects(np.zeros(m), np.zeros(m, dtype=np.uint8))
    c.set_residual(y - y.mean(), np.zeros(n))
    kw = dict(logpi=(0.0, 0.0), lam=1.0, lam2=1.0, s2varg_df=varg * 4.0 * 0.5)
    for it in range(6):
        c.sweep("BayesRR", it, vare, varg, **kw)
    c.set_profiling(2)
    npan = (m + 511) // 512
    c.L.hb_ctx_debug_stamps.argtypes = [ct.c_void_p, ct.c_void_p]
    for it in range(6, 9):
        c.sweep("BayesRR", it, vare, varg, **kw)
        st = np.zeros((npan, 32), dtype=np.int64)
        check(c.L.hb_ctx_debug_stamps(c.h, st.ctypes.data))
        t0 = st[0, 11]
        print("sweep %d: panel: loop top / opening done / end of steps (cycles after panel 0's loop top)" % it)
        print("   panel 0, clock after each step, cycles after its opening: %s" % [int(st[0, k] - st[0, 0]) for k in range(1, 10)])
        print("   panel 1, the same: %s" % [int(st[1, k] - st[1, 0]) for k in range(1, 10)])
        for p in list(range(0, 3)) + [100]:
            print("   panel %3d: top %9d | opening done %9d (+%d) | after the eight steps %9d | period to the next panel %d" % (
                p, st[p, 11] - t0, st[p, 0] - t0, st[p, 0] - st[p, 11], st[p, 9] - t0, st[p + 1, 11] - st[p, 11]))
