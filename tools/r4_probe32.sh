#!/bin/bash
O=gpurun_out
for lib in "" "$PWD/build/variants/nodots.so"; do
echo "== $lib"
HIBAYES_GPU_LIB=$lib timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 60 2>&1 | tail -1
done
HIBAYES_GPU_LIB=$PWD/build/variants/nodots.so timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_recovery.py -q -x -k "BayesR and not BayesRR" 2>&1 | tail -1
export HIBAYES_GPU_LIB=$PWD/build/variants/stamps.so
STAMPS=1 timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 40 > $O/r4_bayesr_stamps6.log 2>&1; tail -8 $O/r4_bayesr_stamps6.log
