#!/bin/bash
for cfg in "1 1600" "2 1600" "2 1000" "2 800"; do
  set -- $cfg
  echo "== CPL=$1 tiles=$2"
  HB_DOTQ2_CPL=$1 HB_DOTQ2_TILES=$2 HB_MV_BITS=2 timeout 300 python tools/matvec_only.py 50000 100000 2 5 2>&1 | tail -1
  HB_DOTQ2_CPL=$1 HB_DOTQ2_TILES=$2 timeout 600 python bench.py --no-ab --tertiary "" --secondary "" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], d['roofline']['avg_launch_ms'])"
done
