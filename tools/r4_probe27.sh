#!/bin/bash
for lib in "" "$PWD/build/variants/two.so"; do
 for tiles in 1600 2000; do
  echo "== lib=$lib tiles=$tiles"
  HB_DOTQ2_TILES=$tiles HIBAYES_GPU_LIB=$lib HB_MV_BITS=2 timeout 300 python tools/matvec_only.py 50000 100000 2 5 2>&1 | tail -1
  HB_DOTQ2_TILES=$tiles HIBAYES_GPU_LIB=$lib timeout 600 python bench.py --no-ab --tertiary "" --secondary "" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], d['roofline']['avg_launch_ms'])"
 done
done
