#!/bin/bash
O=gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py -q -x -k "BayesL or dense or BayesRR or BayesA" > $O/r4_p29_tests.txt 2>&1; grep "passed\|failed" $O/r4_p29_tests.txt )
( timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -k "BayesL" > $O/r4_p29_tests2.txt 2>&1; grep "passed\|failed" $O/r4_p29_tests2.txt )
timeout 600 python bench.py --no-ab --secondary "" --tertiary "BayesL,BayesRR" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], [(t['model'], round(t['value'],2)) for t in d['all_move']])"
