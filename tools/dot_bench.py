"""Development probe: mat-vec launch duration vs panels per launch."""
import sys, os, ctypes as ct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
n, m = int(sys.argv[1]), int(sys.argv[2])
c = H.Context(n, m, panel=512); c.generate(1, 1000); c.marker_stats()
c.set_residual(np.random.default_rng(0).normal(size=n), np.zeros(n))
f = c.L.hbk_dot_bench; f.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_double)]
for D in (1, 2, 4, 8, 16):
    for tk in (0, 1):
        v = ct.c_double(); rc = f(c.h, D, 5, tk, ct.byref(v))
        mb = n * 512 * D / 1e6
        print("D=%2d ticket=%d: %.2f us per launch, %.1f MB -> %.2f TB/s" % (D, tk, v.value, mb, mb / v.value / 1e6 * 1e6 / 1e6))
