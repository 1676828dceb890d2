#!/bin/bash
# round 4: the full GPU suite (timed), the dense soak on the shipped build (100 ms time-out, replay), the bench line
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/r4_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -25 $O/r4_gpu_suite.log
timeout 900 python tools/soak.py dense all 10000 > $O/r4_dense_soak.log 2>&1; echo "dense soak rc=$?"
grep -E "replaying|FAILED|done:|dense soak" $O/r4_dense_soak.log | cut -c1-220
timeout 900 python bench.py > $O/r4_bench_4.json 2> $O/r4_bench_4.err; echo "bench rc=$?"; tail -2 $O/r4_bench_4.err
