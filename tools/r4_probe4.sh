#!/bin/bash
O=gpurun_out
export HB_DEBUG_ABORT=1
timeout 330 python tools/soak.py dense rr 9000 > $O/r4_soak_flush.log 2>&1; echo "soak flush rc=$?"
grep -c "replaying" $O/r4_soak_flush.log; grep "long waits" $O/r4_soak_flush.log | tail -3; tail -2 $O/r4_soak_flush.log
HIBAYES_GPU_LIB=$PWD/build/variants/noflush.so timeout 330 python tools/soak.py dense rr 9000 > $O/r4_soak_noflush.log 2>&1; echo "soak noflush rc=$?"
grep -c "replaying" $O/r4_soak_noflush.log; tail -2 $O/r4_soak_noflush.log
HIBAYES_GPU_LIB=$PWD/build/variants/pubatomic.so timeout 330 python tools/soak.py dense rr 9000 > $O/r4_soak_pubatomic.log 2>&1; echo "soak pubatomic rc=$?"
grep -c "replaying" $O/r4_soak_pubatomic.log; tail -2 $O/r4_soak_pubatomic.log
