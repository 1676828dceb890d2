#!/bin/bash
O=gpurun_out
export HIBAYES_GPU_LIB=$PWD/build/variants/stamps.so
STAMPS=1 timeout 400 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1 3,1" 40 > $O/r4_bayesr_stamps_fwd.log 2>&1; tail -12 $O/r4_bayesr_stamps_fwd.log
HB_FWD_R=1 STAMPS=1 timeout 400 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 40 > $O/r4_bayesr_stamps_nofwd.log 2>&1; tail -10 $O/r4_bayesr_stamps_nofwd.log
