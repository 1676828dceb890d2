#!/bin/bash
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_recovery.py tests/test_gpu_parity.py -q -x > $O/r4_p33_tests.txt 2>&1; grep "passed\|failed" $O/r4_p33_tests.txt )
timeout 900 python tools/soak.py > $O/r04_sparse_soak.txt 2>&1; grep "done:" $O/r04_sparse_soak.txt
