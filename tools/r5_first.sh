#!/bin/bash
# round 5, first GPU session: the new parity tests, the bench line under the driver's own arguments, the threaded CPU baseline
cd /root/repo
mkdir -p gpurun_out
( time python -m pytest tests/test_gpu_depth.py -m gpu -x -q -k "long_chain or continued_chain" -s --durations=10 ) > gpurun_out/r5_long.log 2>&1
( time python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config2 or config4 or stationary" -s --durations=10 ) > gpurun_out/r5_configs.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_bench_driver_args.json 2> gpurun_out/r5_bench_driver_args.err
nproc > gpurun_out/r5_host.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r5_host.txt; free -g >> gpurun_out/r5_host.txt
tail -3 gpurun_out/r5_long.log; tail -3 gpurun_out/r5_configs.log; tail -c 600 gpurun_out/r5_bench_driver_args.err
