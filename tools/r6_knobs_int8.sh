#!/bin/bash
# round 6, closing check: BayesCpi on int8 columns (north_star's layout; k_dotq) under geometry and tile-count knobs
B='python bench.py --steps 60 --warmup 10 --no-cpu --secondary "" --tertiary "" --no-ab --stamped 0 --bits 8'
run() { label="$1"; shift; v=$(env "$@" timeout 120 bash -c "$B" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f sweeps/s, launch %.2f us, frac %.3f (%s)' % (d['value'], d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac'], d['config']['workload'][-28:]))"); echo "$label: $v"; }
run "defaults" HB_X=0
run "geometry (3,7)" HB_BENCH_GEO_BayesCpi=1,3,7 HB_BENCH_KEEP_GEO=1
run "HB_DOTQ_TILES=700" HB_DOTQ_TILES=700
run "HB_DOTQ_TILES=900" HB_DOTQ_TILES=900
run "HB_DOTQ_TILES=1100" HB_DOTQ_TILES=1100
run "HB_WARM_G=0" HB_WARM_G=0
run "defaults again" HB_X=0
