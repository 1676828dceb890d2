#!/bin/bash
O=gpurun_out
export HB_DEBUG_ABORT=1
timeout 1200 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_kernels.py tests/test_gpu_depth.py -x -q > $O/r4_kernel_tests.log 2>&1; echo "kernel+depth tests rc=$?"; tail -5 $O/r4_kernel_tests.log
timeout 400 python tools/soak.py dense rr 9000 > $O/r4_soak_p2.log 2>&1; echo "soak precise=2 rc=$?"
grep -c "replaying" $O/r4_soak_p2.log; tail -2 $O/r4_soak_p2.log
unset HB_DEBUG_ABORT
for cfg in "0 3072 main" "0 3072 noscale" "2 800 main" "2 400 main" "2 1600 main"; do
  set -- $cfg
  lib=""; [ "$3" = "noscale" ] && lib="$PWD/build/variants/noscale.so"
  echo "== kind=$1 tiles=$2 lib=$3"
  HIBAYES_GPU_LIB=$lib HB_MV_BITS=2 HB_DOTQ2_KIND=$1 HB_DOTQ2_TILES=$2 timeout 300 python tools/matvec_only.py 50000 100000 2 5 2>&1 | tail -1
done > $O/r4_matvec_variants2.log 2>&1
cat $O/r4_matvec_variants2.log
timeout 1200 python bench.py --steps 50 --warmup 30 > $O/r4_bench_second.json 2> $O/r4_bench_second.err; echo "bench rc=$?"; tail -3 $O/r4_bench_second.err
