#!/bin/bash
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py tests/test_gpu_c_caller.py -x -q > $O/r4_tests8.log 2>&1; echo "tests rc=$?"; tail -3 $O/r4_tests8.log
timeout 300 python tools/launch_roles.py 2 3 > $O/r4_roles_valu2.txt 2>&1; tail -6 $O/r4_roles_valu2.txt
HB_DOTQ2_KIND=2 timeout 300 python tools/launch_roles.py 2 3 > $O/r4_roles_mfma2.txt 2>&1; tail -6 $O/r4_roles_mfma2.txt
timeout 900 python bench.py --tertiary BayesRR > $O/r4_bench_5.json 2> $O/r4_bench_5.err; echo "bench rc=$?"; tail -2 $O/r4_bench_5.err
