#!/bin/bash
O=gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py tests/test_gpu_kernels.py -q -x > $O/r4_p30_tests.txt 2>&1; grep "passed\|failed" $O/r4_p30_tests.txt )
timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1 3,1" 60 2>&1 | tail -2
timeout 900 python bench.py --tertiary "" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], d['roofline']['avg_launch_ms'], 'mfma', d['mfma_ab']['value'], 'int8', d['int8']['value'], 'secondary', d['secondary']['value'])"
