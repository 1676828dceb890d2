#!/bin/bash
# round 4, final build: the kernel traces and counter passes of the legs whose kernels changed late in the round (k_dotq2's launch shape, BayesR's chain)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
leg() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  rm -rf $O/trace_$name
  env "${envs[@]}" rocprofv3 --kernel-trace --stats -d $O/trace_$name -o bench -- python $R/bench.py --steps 100 --warmup 30 --no-ab --no-cpu "$@" > $O/r04_bench_under_rocprof_$name.json 2> $O/trace_$name.err
  db=$(find $O/trace_$name -name "*.db" | head -1)
  python $R/tools/rocprof_window.py $db --after 11 --sweeps 100 > $O/r04_kernel_trace_timed_window_$name.txt 2>&1
  head -8 $O/r04_kernel_trace_timed_window_$name.txt | cut -c1-170
  rm -rf $O/trace_$name; }
run_pmc() { name=$1; shift; ctrs=$1; shift
  rm -rf $O/pmc_$name
  env "$@" rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmc_$name -o res -- python $R/tools/matvec_only.py 50000 100000 2 1 > $O/pmc_$name.log 2>&1
  db=$(find $O/pmc_$name -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $db > $O/r04_pmc_$name.txt 2>&1
  grep -E "k_dotq" $O/r04_pmc_$name.txt | head -6; rm -rf $O/pmc_$name; }
leg 2bit X=1 -- --secondary "" --tertiary ""
leg bayesr X=1 -- --bits 8 --model BayesR --secondary "" --tertiary "" --burnin 300
run_pmc fetch_2bit_d7 FETCH_SIZE HB_MV_BITS=2 HB_TIME_MATVEC_D=7
run_pmc sq_k_dotq2 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" HB_MV_BITS=2 HB_TIME_MATVEC_D=7
