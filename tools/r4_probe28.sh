#!/bin/bash
for rep in 1 2; do
for cfg in "|2000" "$PWD/build/variants/two.so|1600" "$PWD/build/variants/two.so|1400"; do
  lib=${cfg%%|*}; tiles=${cfg##*|}
  echo "== lib=$lib tiles=$tiles"
  HB_DOTQ2_TILES=$tiles HIBAYES_GPU_LIB=$lib timeout 600 python bench.py --no-ab --tertiary "" --secondary "" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], d['roofline']['avg_launch_ms'])"
done
done
