#!/bin/bash
O=gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py -q -x > $O/r4_p24_tests.txt 2>&1; grep "passed\|failed" $O/r4_p24_tests.txt )
timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1 3,1" 60 2>&1 | tail -2
export HIBAYES_GPU_LIB=$PWD/build/variants/stamps.so
STAMPS=1 timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 40 > $O/r4_bayesr_stamps7.log 2>&1; tail -8 $O/r4_bayesr_stamps7.log
