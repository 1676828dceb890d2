#!/bin/bash
O=gpurun_out
for t in 3072 2600 2300 2000 1700; do
  echo "== tiles=$t isolated"; HB_MV_BITS=2 HB_DOTQ2_TILES=$t timeout 300 python tools/matvec_only.py 50000 100000 2 5 2>&1 | tail -1
done > $O/r4_tiles.log 2>&1; cat $O/r4_tiles.log
for t in 2300 2000; do echo "== tiles=$t in situ"; HB_DOTQ2_TILES=$t timeout 300 python tools/launch_roles.py 2 3 2>&1 | tail -6; done > $O/r4_tiles_insitu.log 2>&1; cat $O/r4_tiles_insitu.log
timeout 600 python bench.py --no-ab --tertiary "" --no-cpu > $O/r4_bench_6.json 2> $O/r4_bench_6.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_6.json').read().strip().splitlines()[-1])
print("bench (update rows 16 in flight): value", d["value"], "secondary", d["secondary"]["value"], d["secondary"]["roofline"]["avg_launch_ms"])
PY
