#!/bin/bash
# round 6, closing check at the new default geometry (2, 7): structural switches of the headline's chain
B='python bench.py --steps 60 --warmup 10 --no-cpu --secondary "" --tertiary "" --no-ab --stamped 0'
run() { label="$1"; shift; v=$(env "$@" timeout 120 bash -c "$B" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f sweeps/s, launch %.2f us' % (d['value'], d['roofline']['avg_launch_ms']*1e3))"); echo "$label: $v"; }
run "defaults" HB_X=0
run "HB_FWD=0 (the chain folds everything itself)" HB_FWD=0
run "HB_CERT=0 (no certificate)" HB_CERT=0
run "HB_GRAM16=1 (compact band)" HB_GRAM16=1
run "HB_WARM_GROUP=1" HB_WARM_GROUP=1
run "HB_GATE=1" HB_GATE=1
run "HB_SIDE_FIRST=1" HB_SIDE_FIRST=1
run "defaults again" HB_X=0
