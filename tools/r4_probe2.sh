#!/bin/bash
# round 4, second GPU session: memory-side polls against the stale-line stall (dense soak), exactness of the new kernels and of BayesR with
# k_fwd beside its chain, isolated mat-vec timings of the 2-bit variants, a first bench line
O=gpurun_out
export HB_DEBUG_ABORT=1
timeout 900 python -m pytest tests/test_gpu_recovery.py tests/test_gpu_kernels.py tests/test_gpu_depth.py::test_matrix_core_matvec_is_the_same_chain_bit_for_bit "tests/test_gpu_depth.py::test_default_geometry_draw_for_draw_at_pipeline_depth" tests/test_gpu_depth.py::test_dense_chain_draw_for_draw -x -q > $O/r4_kernel_tests.log 2>&1; echo "kernel tests rc=$?"; tail -5 $O/r4_kernel_tests.log
HB_DEBUG_LDIAG=1 timeout 460 python tools/soak.py dense rr 9000 > $O/r4_soak_fresh.log 2>&1; echo "soak fresh rc=$?"
grep -c "replaying" $O/r4_soak_fresh.log; grep "launch_dotq" $O/r4_soak_fresh.log | head -2; tail -2 $O/r4_soak_fresh.log
unset HB_DEBUG_ABORT
for cfg in "0 1 256 3072" "0 2 256 3072" "0 2 256 1536" "0 2 512 1536" "0 1 512 3072" "0 1 256 6000" "1 1 256 3072" "2 1 256 3072" "2 1 256 1536" "2 1 256 6000" "2 1 256 784"; do
  set -- $cfg
  echo "== kind=$1 cpl=$2 rs=$3 tiles=$4"
  HB_MV_BITS=2 HB_DOTQ2_KIND=$1 HB_DOTQ2_CPL=$2 HB_DOTQ2_RS=$3 HB_DOTQ2_TILES=$4 timeout 300 python tools/matvec_only.py 50000 100000 2 5 2>&1 | tail -1
done > $O/r4_matvec_variants.log 2>&1
cat $O/r4_matvec_variants.log
timeout 1200 python bench.py --steps 50 --warmup 30 > $O/r4_bench_first.json 2> $O/r4_bench_first.err; echo "bench rc=$?"; tail -c 1500 $O/r4_bench_first.json; tail -5 $O/r4_bench_first.err
