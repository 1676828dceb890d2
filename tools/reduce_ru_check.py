"""k_reduce_ru on sixteen workgroups against the one-workgroup kernel of a library built before the change (HIBAYES_GPU_LIB): the residual's sums
after a few sweeps must be the same BITS. usage: python tools/reduce_ru_check.py  (run once per library; prints the sums as hex)"""
import sys, os, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
n, m = 5003, 4096
with H.Context(n, m, panel=512, seed=11) as c:
    c.generate(20240901, 1000)
    rng = np.random.default_rng(5)
    y = rng.standard_normal(n)
    r = H.Bayes(y, None, "BayesCpi", [0.95, 0.05], niter=30, nburn=10, thin=1, seed=3, verbose=False, ctx=c)
    a, b = ct.c_double(), ct.c_double()
    c.L.hb_ctx_residual_sums.argtypes = [ct.c_void_p, ct.POINTER(ct.c_double), ct.POINTER(ct.c_double)]
    assert c.L.hb_ctx_residual_sums(c.h, ct.byref(a), ct.byref(b)) == 0
    print("sum r %s sum r2 %s Ve %s Vg %s mu %s" % (a.value.hex(), b.value.hex(), float(r["Ve"]).hex(), float(r["Vg"]).hex(), float(r["mu"]).hex()))
