#!/bin/bash
# BayesR chain, round-4 late: early requests / pre-issued fold rows / block-speculative serial pass — parity, stamps, rate
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_depth.py -q -x -k "BayesR and not BayesRR" 2>&1 | tail -3 ) 
( timeout 600 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_parity.py -q -x -k "BayesR and not BayesRR" 2>&1 | tail -3 )
export HIBAYES_GPU_LIB=$PWD/build/variants/stamps.so
STAMPS=1 TUNES="6,0.64 6,0.8 6,0.9" timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1 3,1" 40 > $O/r4_bayesr_stamps3.log 2>&1; tail -12 $O/r4_bayesr_stamps3.log
unset HIBAYES_GPU_LIB
timeout 600 python bench.py --no-ab --tertiary "" --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'secondary', d.get('secondary'))"
