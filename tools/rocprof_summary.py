"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel text summary kept under profiles/."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by 6 desc"))
tot = sum(r[5] for r in rows)
print("%-70s %8s %12s %10s %12s %10s %6s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ms", "%"))
for r in rows:
    print("%-70s %8d %12.1f %10d %12d %10.2f %6.2f" % (r[0][:70], r[1], r[2], r[3], r[4], r[5] / 1e6, 100.0 * r[5] / tot))
try:
    cols = [d[1] for d in cur.execute("pragma table_info(pmc_events)")]
    if cols:
        q = "select k.name, p.counter_name, count(*), avg(p.value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name order by 4 desc"
        print("\ncounters (avg per dispatch):")
        for r in cur.execute(q):
            print("%-60s %-16s n=%6d avg=%.1f" % (r[0][:60], r[1], r[2], r[3]))
except Exception as e:
    print("(no counter table: %s)" % e)
