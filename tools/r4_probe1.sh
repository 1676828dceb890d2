#!/bin/bash
# round 4, first GPU session: the replay of an aborted sweep (tests), then dense soaks with the abort diagnostics on.
export HB_DEBUG_ABORT=1
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_recovery.py -x -q > $O/r4_recovery_tests.log 2>&1; echo "recovery tests rc=$?"
tail -3 $O/r4_recovery_tests.log
timeout 420 python tools/soak.py dense rr 12000 > $O/r4_soak_default.log 2>&1; echo "soak default rc=$?"
grep -c "replaying" $O/r4_soak_default.log; tail -2 $O/r4_soak_default.log
HB_STREAM_PRIO=1 timeout 420 python tools/soak.py dense rr 12000 > $O/r4_soak_prio.log 2>&1; echo "soak prio rc=$?"
grep -c "replaying" $O/r4_soak_prio.log; tail -2 $O/r4_soak_prio.log
HIBAYES_GPU_LIB=$PWD/build/variants/backoff.so timeout 420 python tools/soak.py dense rr 12000 > $O/r4_soak_backoff.log 2>&1; echo "soak backoff rc=$?"
grep -c "replaying" $O/r4_soak_backoff.log; tail -2 $O/r4_soak_backoff.log
GPU_MAX_HW_QUEUES=8 timeout 420 python tools/soak.py dense rr 12000 > $O/r4_soak_q8.log 2>&1; echo "soak q8 rc=$?"
grep -c "replaying" $O/r4_soak_q8.log; tail -2 $O/r4_soak_q8.log
