#!/usr/bin/env python
"""Per-kernel statistics of a rocprofv3 --kernel-trace run (rocpd sqlite output) restricted to the TIMED sweeps of bench.py.

    python tools/rocprof_window.py gpurun_out/prof/bench_results.db --after 11 --sweeps 20

A sweep is delimited by the dispatch of its chain workgroup (k_chain_group / k_chain_persist: one per sweep). `--after` skips
that many sweeps at the end of the run (bench.py's stamped sweeps: 1 capture + --stamped), `--sweeps` is the number of timed
steps; the window runs from the start of the first timed sweep's chain kernel to the end of the last one's. Kernels are grouped
by (name, grid size): the pipeline's k_dotq launches (update rows + finalize + tiles) differ from the isolated replay's by grid.
Prints a table for profiles/."""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--after", type=int, default=11)
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--chain", default="k_chain_")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size from kernels order by start").fetchall()
    chain = [r for r in rows if a.chain in r[0]]
    # (run bench.py with --secondary "" under the tracer: every chain dispatch then belongs to the headline model)
    ch = chain
    first_model = ch[-1][0]
    hi = len(ch) - a.after
    lo = hi - a.sweeps
    t0, t1 = ch[lo][1], ch[hi - 1][2]
    print("window: sweeps %d..%d of %d of %s: %.3f ms for %d sweeps = %.4f ms per sweep" %
          (lo, hi - 1, len(ch), first_model.split("(")[0][:60], (t1 - t0) * 1e-6, a.sweeps, (t1 - t0) * 1e-6 / a.sweeps))
    agg = {}
    for name, s, e, gx, wx, vg, sg, lds in rows:
        if s < t0 or e > t1 + 2000000:
            continue
        k = (name.split("(")[0][:70], gx // max(wx, 1), wx)
        d = agg.setdefault(k, [0, 0, 10 ** 18, 0, vg, sg, lds])
        d[0] += 1
        d[1] += e - s
        d[2] = min(d[2], e - s)
        d[3] = max(d[3], e - s)
    print("%-72s %8s %6s %9s %12s %10s %10s %10s %5s %5s %7s" % ("kernel", "blocks", "wg", "calls", "total_us", "avg_ns", "min_ns", "max_ns", "vgpr", "sgpr", "lds"))
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %8d %6d %9d %12.1f %10.0f %10d %10d %5d %5d %7d" % (k[0], k[1], k[2], d[0], d[1] * 1e-3, d[1] / d[0], d[2], d[3], d[4], d[5], d[6]))


if __name__ == "__main__":
    main()
