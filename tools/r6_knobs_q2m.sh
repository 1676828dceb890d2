#!/bin/bash
# round 6, closing check: the matrix-core mat-vec's variants under the headline at geometry (2, 7)
B='python bench.py --steps 60 --warmup 10 --no-cpu --secondary "" --tertiary "" --no-ab --stamped 0'
run() { label="$1"; shift; v=$(env "$@" timeout 120 bash -c "$B" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f sweeps/s, launch %.2f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"); echo "$label: $v"; }
run "defaults" HB_X=0
run "HB_Q2M_SC=0" HB_Q2M_SC=0
run "HB_Q2M_CT=8" HB_Q2M_CT=8
run "HB_Q2M_CT=16" HB_Q2M_CT=16
run "HB_Q2M_G=1" HB_Q2M_G=1
run "HB_Q2M_G=2" HB_Q2M_G=2
run "HB_Q2M_G=3" HB_Q2M_G=3
run "HB_DOTQ2_TILES=700" HB_DOTQ2_TILES=700
run "HB_DOTQ2_TILES=560" HB_DOTQ2_TILES=560
run "defaults again" HB_X=0
