"""Why are the pipeline's mat-vec launches longer in situ than replayed alone? Stamps of the launches of a sweep (a) as the sweep runs
them and (b) in the chain-alone diagnostic mode (hb_ctx_set_profiling bit 2: the same launches with their update and finalize blocks,
chain_done pre-set, the chain workgroup afterwards: no concurrent chain, empty move lists).  python tools/insitu_probe.py [bits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hibayes_amd as H
import bench as B

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, m = 50000, 500000
with H.Context(n, m, seed=20240901) as c:
    c.generate(20240901, 1000)
    y = B.synth_phenotype(c, n, m, 0, m, 20240901, None, "BayesCpi")
    c.set_pipeline(1, 2, 7)
    c.build_gram()
    if bits == 2:
        c.set_layout(2, keep_int8=False)
    xpx, vx, sumvx, nvar0 = c.marker_stats() if bits == 8 else (None, None, 100000.0, 0)
    vare, varg = 0.5, 0.5 / (0.05 * 1.5e5)
    logpi = np.log([0.999, 0.001])
    c.set_effects(np.zeros(m), np.zeros(m, dtype=np.uint8))
    c.set_residual(y - y.mean(), np.zeros(n))
    for it in range(40):
        c.sweep("BayesCpi", it, vare, varg, logpi=logpi)
    for mode, name in ((8, "in situ (chain beside the launches)"), (12, "chain-alone mode (same launches, no concurrent chain, empty move lists)")):
        c.set_profiling(mode)
        acc = []
        for it in range(40, 46):
            s = c.sweep("BayesCpi", it, vare, varg, logpi=logpi)
            st = c.matvec_stamps()
            acc.append((st["avg_ms"] * 1e3, st["span_ms"], s["n_events"]))
        c.set_profiling(0)
        a = np.array(acc[1:])
        print("bits %d %s: %.2f us per launch, stream span %.3f ms, moves per sweep %.0f" % (bits, name, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean()))
    ms, nl, nc = c.time_matvec(reps=3)
    print("bits %d isolated replay (no update blocks): %.2f us per launch period" % (bits, ms * 1e3))
