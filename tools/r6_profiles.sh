#!/bin/bash
# round 6: rocprofv3 --kernel-trace --stats of bench.py restricted to the timed sweeps (tools/rocprof_window.py), one run per leg
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
leg() { # name, -- bench args
  name=$1; shift; shift
  rm -rf $O/trace_$name
  rocprofv3 --kernel-trace --stats -d $O/trace_$name -o bench -- python $R/bench.py --steps 100 --warmup 30 --no-ab --no-cpu --secondary "" --tertiary "" "$@" > $O/r06_bench_under_rocprof_$name.json 2> $O/trace_$name.err
  db=$(find $O/trace_$name -name "*.db" | head -1)
  python $R/tools/rocprof_window.py $db --after 11 --sweeps 100 > $O/r06_kernel_trace_timed_window_$name.txt 2>&1
  head -8 $O/r06_kernel_trace_timed_window_$name.txt | cut -c1-170
  rm -rf $O/trace_$name
}
for l in ${LEGS:-2bit_mfma int8 bayesr_converged bayesrr}; do
  case $l in
    2bit_mfma) leg 2bit_mfma -- ;;
    int8) leg int8 -- --bits 8 ;;
    bayesr_converged) leg bayesr_converged -- --bits 8 --model BayesR --init-state $R/profiles/state/bayesr_config3.npz --burnin 40 ;;
    bayesrr) leg bayesrr -- --bits 8 --model BayesRR --burnin 20 ;;
  esac
done
