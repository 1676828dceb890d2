#!/bin/bash
# round 4, validation of the final build: full GPU suite (timed), sparse and dense soaks, bench
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/r4_gpu_suite2.log 2>&1; echo "gpu suite rc=$?"; grep -E "passed|failed|real" $O/r4_gpu_suite2.log
timeout 600 python tools/soak.py > $O/r4_sparse_soak.log 2>&1; echo "sparse soak rc=$?"; grep -E "done:|replaying|FAILED|Error" $O/r4_sparse_soak.log | cut -c1-200
timeout 900 python tools/soak.py dense all 8000 > $O/r4_dense_soak2.log 2>&1; echo "dense soak rc=$?"; grep -E "done:|dense soak|FAILED" $O/r4_dense_soak2.log | cut -c1-200
timeout 900 python bench.py > $O/r4_bench_9.json 2> $O/r4_bench_9.err; echo "bench rc=$?"
