#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats of bench.py, restricted to the timed sweeps (tools/rocprof_window.py), one run per leg
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
leg() { # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  rm -rf $O/trace_$name
  env "${envs[@]}" rocprofv3 --kernel-trace --stats -d $O/trace_$name -o bench -- python $R/bench.py --steps 100 --warmup 30 --no-ab --no-cpu "$@" > $O/r04_bench_under_rocprof_$name.json 2> $O/trace_$name.err
  db=$(find $O/trace_$name -name "*.db" | head -1)
  python $R/tools/rocprof_window.py $db --after 11 --sweeps 100 > $O/r04_kernel_trace_timed_window_$name.txt 2>&1
  head -8 $O/r04_kernel_trace_timed_window_$name.txt | cut -c1-170
  rm -rf $O/trace_$name
}
leg 2bit X=1 -- --secondary "" --tertiary ""
leg 2bit_mfma HB_DOTQ2_KIND=2 -- --secondary "" --tertiary ""
leg int8 X=1 -- --bits 8 --secondary "" --tertiary ""
leg bayesr X=1 -- --bits 8 --model BayesR --secondary "" --tertiary "" --burnin 300
leg bayesrr X=1 -- --bits 8 --model BayesRR --secondary "" --tertiary "" --burnin 20
