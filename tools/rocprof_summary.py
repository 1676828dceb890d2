"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel text summary kept under profiles/.

k_dot launches are listed by grid height: the pipeline's launches carry the residual-update row (grid.y = nsplit + 2), whose
duration includes that row's wait for the chain workgroup; the launches bench.py times for the roofline (hb_ctx_time_matvec)
carry only the mat-vec rows and the partial-sum row (grid.y = nsplit + 1) — that line is the one to compare with
roofline.avg_launch_ms."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute(
    "select case when name like '%k_dot%' then name || ' grid=(' || cast((grid_x/workgroup_x) as text) || ',' || cast(grid_y as text) || ')' else name end as nm, "
    "count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by nm order by 6 desc"))
tot = sum(r[5] for r in rows)
print("%-86s %8s %12s %10s %12s %10s %6s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ms", "%"))
for r in rows:
    nm = r[0]
    if "k_dot<" in nm:
        nm = "k_dot" + nm[nm.index("<"):nm.index(">") + 1] + nm[nm.rindex(" grid="):]
    elif "k_dotq" in nm:
        nm = nm[:nm.index("(")] + nm[nm.rindex(" grid="):]
    print("%-86s %8d %12.1f %10d %12d %10.2f %6.2f" % (nm[:86], r[1], r[2], r[3], r[4], r[5] / 1e6, 100.0 * r[5] / tot))
try:
    cols = [d[1] for d in cur.execute("pragma table_info(pmc_events)")]
    if cols:
        q = ("select k.name, k.grid_x/k.workgroup_x, k.grid_y, p.counter_name, count(*), avg(p.counter_value), min(p.counter_value), max(p.counter_value) "
             "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, k.grid_x, k.grid_y, p.counter_name order by 6 desc")
        out = list(cur.execute(q))
        if out:
            print("\ncounters per dispatch (FETCH_SIZE is in KiB; on gfx950 corrected HBM bytes = 2 * 1024 * FETCH_SIZE, MI355X_MICROARCH.md):")
            print("%-60s %-12s %-12s %7s %14s %14s %14s" % ("kernel", "grid", "counter", "n", "avg", "min", "max"))
            for r in out:
                print("%-60s %-12s %-12s %7d %14.1f %14.1f %14.1f" % (r[0][:60], "(%d,%d)" % (r[1], r[2]), r[3], r[4], r[5], r[6], r[7]))
except Exception as e:
    print("(no counter table: %s)" % e)
