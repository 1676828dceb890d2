#!/bin/bash
O=gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py -q -x > $O/r4_p34_tests.txt 2>&1; grep "passed\|failed" $O/r4_p34_tests.txt )
timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 60 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
