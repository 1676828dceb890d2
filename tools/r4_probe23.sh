#!/bin/bash
O=gpurun_out
export HIBAYES_GPU_LIB=$PWD/build/variants/stamps.so
STAMPS=1 timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 40 > $O/r4_bayesr_stamps4.log 2>&1; tail -12 $O/r4_bayesr_stamps4.log
