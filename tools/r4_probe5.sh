#!/bin/bash
O=gpurun_out
export HB_DEBUG_ABORT=1
timeout 600 python tools/soak.py dense rr 18000 > $O/r4_soak_flush.log 2>&1; echo "soak flush rc=$?"
grep -c "replaying" $O/r4_soak_flush.log; grep "long waits" $O/r4_soak_flush.log | tail -4; tail -2 $O/r4_soak_flush.log
unset HB_DEBUG_ABORT
for cfg in "0 3072" "2 800" "2 1600" "2 2744"; do
  set -- $cfg
  echo "== kind=$1 tiles=$2"
  HB_MV_BITS=2 HB_DOTQ2_KIND=$1 HB_DOTQ2_TILES=$2 timeout 300 python tools/matvec_only.py 50000 100000 2 5 2>&1 | tail -1
done > $O/r4_matvec_variants3.log 2>&1
cat $O/r4_matvec_variants3.log
timeout 900 python bench.py --steps 50 --warmup 30 --tertiary BayesRR > $O/r4_bench_third.json 2> $O/r4_bench_third.err; echo "bench rc=$?"; tail -3 $O/r4_bench_third.err
