#!/bin/bash
# round-4 closing run: the whole GPU suite, then the default bench line, on the committed build
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/r04_gpu_tests_final.txt 2>&1
tail -3 $O/r04_gpu_tests_final.txt
timeout 900 python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err
tail -c 3000 $O/r04_bench_final.json
