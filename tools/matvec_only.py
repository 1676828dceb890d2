"""The sweep's panel mat-vec launches alone (hb_ctx_time_matvec: same columns per launch and finalize rows as the pipeline,
no chain, no update row) — the command the hardware-counter passes profile, because the real pipeline's two graph branches
hand-shake through memory and cannot run under --pmc serialisation.
    python tools/matvec_only.py [n m precise reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H

n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, int(sys.argv[2]) if len(sys.argv) > 2 else 500000
precise = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
with H.Context(n, m, precise=precise, seed=1) as c:
    c.generate(20240901, 1000)
    c.marker_stats()
    c.set_pipeline(1, 2, 7)
    bits = int(os.environ.get("HB_MV_BITS", "8"))
    if bits == 2:
        c.set_layout(2, keep_int8=False)
    c.set_residual(np.random.default_rng(0).normal(size=n), np.zeros(n))
    ms, nl, nc = c.time_matvec(reps=reps)
    print("precise=%d bits=%d: %d launches of %d columns, %.2f us per launch, %.3f TB/s of resident bytes (%.3f at one byte per genotype)" % (
        precise, bits, nl, nc, ms * 1e3, n * nc * bits / 8 / (ms * 1e-3) / 1e12, n * nc / (ms * 1e-3) / 1e12))
