#!/bin/bash
O=gpurun_out
for v in stamps stamps_ff; do
echo "== $v"
HIBAYES_GPU_LIB=$PWD/build/variants/$v.so GT_BITS=2 timeout 300 python tools/group_timeline.py BayesCpi 300 2>&1 | grep -E "5-12 moves|dots .*rank|busy cycles|period:" | head -5
done
for lib in "" "$PWD/build/variants/ff.so"; do
HIBAYES_GPU_LIB=$lib timeout 600 python bench.py --tertiary "" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib value', d['value'], d['roofline']['avg_launch_ms'], 'mfma', d['mfma_ab']['value'], d['mfma_ab']['roofline']['avg_launch_ms'], 'R', d['secondary']['value'])"
done
