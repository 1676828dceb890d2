#!/bin/bash
O=gpurun_out
timeout 900 python bench.py --gpus 2 --backend gloo --m 20000 --no-cpu --steps 50 --warmup 20 --burnin 50 --burnin-secondary 20 > $O/r4_bench_gpus2_gloo.json 2> $O/r4_bench_gpus2_gloo.err; echo "rc=$?"; tail -c 1200 $O/r4_bench_gpus2_gloo.json; tail -5 $O/r4_bench_gpus2_gloo.err
timeout 900 python bench.py --gpus 2 --m 20000 --no-cpu --steps 50 --warmup 20 --burnin 50 > $O/r4_bench_gpus2_nccl.json 2> $O/r4_bench_gpus2_nccl.err; echo "rc=$?"; tail -c 600 $O/r4_bench_gpus2_nccl.json; tail -3 $O/r4_bench_gpus2_nccl.err
