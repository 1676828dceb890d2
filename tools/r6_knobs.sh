#!/bin/bash
# round 6, closing check: the headline (BayesCpi, 2-bit, certified group chain) under the knobs that were tuned in earlier rounds — are the defaults still the best?
B='python bench.py --steps 60 --warmup 10 --no-cpu --secondary "" --tertiary "" --no-ab --stamped 0'
run() { label="$1"; shift; v=$(env "$@" timeout 120 bash -c "$B" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f sweeps/s, launch %.2f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"); echo "$label: $v"; }
run "defaults" HB_X=0
run "defaults again" HB_X=0
run "HB_WARM_G=0" HB_WARM_G=0
run "HB_WARM_G=2" HB_WARM_G=2
run "HB_WARM_G=8" HB_WARM_G=8
run "HB_CANDF=0.5" HB_CANDF=0.5
run "HB_CANDF=0.7" HB_CANDF=0.7
run "HB_CANDF=0.9" HB_CANDF=0.9
run "HB_KAPPA=2" HB_KAPPA=2
run "HB_KAPPA=6" HB_KAPPA=6
run "HB_STREAM_PRIO=0" HB_STREAM_PRIO=0
run "HB_DOTQ2_TILES=784" HB_DOTQ2_TILES=784
run "HB_DOTQ2_TILES=504" HB_DOTQ2_TILES=504
run "geometry (2,7) fixed" HB_NO_ADAPTIVE=1 HB_BENCH_GEO_BayesCpi=1,2,7
run "geometry (3,7) fixed" HB_NO_ADAPTIVE=1 HB_BENCH_GEO_BayesCpi=1,3,7
