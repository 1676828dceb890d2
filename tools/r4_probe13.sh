#!/bin/bash
O=gpurun_out
HB_ROLES_MODEL=BayesR timeout 300 python tools/launch_roles.py 8 2 > $O/r4_roles_bayesr2.txt 2>&1; tail -6 $O/r4_roles_bayesr2.txt
HB_ROLES_MODEL=BayesR timeout 300 python tools/launch_roles.py 8 3 > $O/r4_roles_bayesr3.txt 2>&1; tail -6 $O/r4_roles_bayesr3.txt
export HIBAYES_GPU_LIB=$PWD/build/variants/stamps.so
STAMPS=1 timeout 400 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1 3,1" 40 > $O/r4_bayesr_stamps2.log 2>&1; tail -6 $O/r4_bayesr_stamps2.log
