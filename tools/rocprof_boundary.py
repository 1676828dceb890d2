#!/usr/bin/env python
"""What runs between the END of one sweep's chain workgroup and the START of the next one's (the sweep's boundary: closing updates, sums,
the host's draws, sweep start), from a rocprofv3 --kernel-trace run of bench.py (rocpd sqlite): every dispatch of one boundary in the
middle of the run with its offset from the chain's end and its duration, then the mean gap over the timed sweeps.

    python tools/rocprof_boundary.py gpurun_out/trace_f/.../bench_results.db [--after 11] [--sweeps 100]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--after", type=int, default=11)
ap.add_argument("--sweeps", type=int, default=100)
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
rows = cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
ch = [r for r in rows if "k_chain_" in r[0]]
hi = len(ch) - a.after
lo = hi - a.sweeps
gaps = [ch[i + 1][1] - ch[i][2] for i in range(lo, hi - 1)]
print("chain end -> next chain start over %d boundaries: mean %.1f us, min %.1f, max %.1f; chain kernel mean %.1f us; sweep mean %.1f us" % (
    len(gaps), sum(gaps) / len(gaps) * 1e-3, min(gaps) * 1e-3, max(gaps) * 1e-3, sum(c[2] - c[1] for c in ch[lo:hi]) / (hi - lo) * 1e-3,
    (ch[hi - 1][1] - ch[lo][1]) / (hi - 1 - lo) * 1e-3))
mid = (lo + hi) // 2
e0, s1 = ch[mid][2], ch[mid + 1][1]
print("one boundary (sweep %d): dispatches that start after the chain's last 60 us and before the next chain's first 60 us" % mid)
for name, s, e, gx, wx in rows:
    if s >= e0 - 60000 and s <= s1 + 60000 and "k_chain_" not in name:
        print("  %+9.1f us  %8.1f us  %5d x %4d  %s" % ((s - e0) * 1e-3, (e - s) * 1e-3, gx // max(wx, 1), wx, name.split("(")[0][:60]))
print("  next chain starts at %+.1f us" % ((s1 - e0) * 1e-3))
