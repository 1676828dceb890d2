#!/bin/bash
O=gpurun_out
for t in 1800 1600 1400 1200; do echo "== tiles=$t in situ"; HB_DOTQ2_TILES=$t timeout 300 python tools/launch_roles.py 2 3 2>&1 | tail -6 | head -1; done > $O/r4_tiles_insitu2.log 2>&1; cat $O/r4_tiles_insitu2.log
for t in 2000 1600; do HB_DOTQ2_TILES=$t timeout 600 python bench.py --no-ab --tertiary "" --secondary "" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles $t: value', d['value'], d['roofline']['avg_launch_ms'])"; done
