#!/bin/bash
# round 5: rocprofv3 --kernel-trace --stats of bench.py restricted to the timed sweeps (tools/rocprof_window.py), one run per leg; then the counter passes
# (each in its own rocprofv3 run: --kernel-trace only beside --pmc) for the mat-vec launch shapes of the sweeps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
leg() { # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  rm -rf $O/trace_$name
  env "${envs[@]}" rocprofv3 --kernel-trace --stats -d $O/trace_$name -o bench -- python $R/bench.py --steps 100 --warmup 30 --no-ab --no-cpu --burnin-converged 0 "$@" > $O/r05_bench_under_rocprof_$name.json 2> $O/trace_$name.err
  db=$(find $O/trace_$name -name "*.db" | head -1)
  python $R/tools/rocprof_window.py $db --after 11 --sweeps 100 > $O/r05_kernel_trace_timed_window_$name.txt 2>&1
  head -8 $O/r05_kernel_trace_timed_window_$name.txt | cut -c1-170
  rm -rf $O/trace_$name
}
leg 2bit_mfma X=1 -- --secondary "" --tertiary ""
leg 2bit_vdot4 X=1 -- --matvec-kernel 0 --secondary "" --tertiary ""
leg int8 X=1 -- --bits 8 --secondary "" --tertiary ""
leg bayesr X=1 -- --bits 8 --model BayesR --secondary "" --tertiary "" --burnin 300
leg bayesrr X=1 -- --bits 8 --model BayesRR --secondary "" --tertiary "" --burnin 20
run_pmc() { # name, counters, env..., then matvec_only args
  name=$1; shift; ctrs=$1; shift
  rm -rf $O/pmc_$name
  env "$@" rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmc_$name -o res -- python $R/tools/matvec_only.py 50000 100000 2 1 > $O/pmc_$name.log 2>&1
  db=$(find $O/pmc_$name -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $db > $O/r05_pmc_$name.txt 2>&1
  grep -E "k_dotq" $O/r05_pmc_$name.txt | head -12
  rm -rf $O/pmc_$name
}
run_pmc fetch_2bit_mfma_d7 FETCH_SIZE HB_MV_BITS=2 HB_TIME_MATVEC_D=7
run_pmc fetch_2bit_vdot4_d7 FETCH_SIZE HB_MV_BITS=2 HB_TIME_MATVEC_D=7 HB_DOTQ2_KIND=0
run_pmc fetch_int8_d7 FETCH_SIZE HB_MV_BITS=8 HB_TIME_MATVEC_D=7
run_pmc sq_k_dotq2m "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" HB_MV_BITS=2 HB_TIME_MATVEC_D=7
# two ranks sharing the one GPU of this box over gloo (what can be checked of `bench.py --gpus N` without a node): the line's per-rank and all-reduce fields
cd $R && HB_BENCH_M=100000 timeout 600 python bench.py --gpus 2 --backend gloo --collective torch --steps 10 --warmup 5 --burnin 30 --no-cpu > $O/r05_bench_gpus2_gloo_one_gpu_box.json 2> $O/r05_bench_gpus2_gloo_one_gpu_box.err; tail -c 1500 $O/r05_bench_gpus2_gloo_one_gpu_box.json | head -c 1500; tail -3 $O/r05_bench_gpus2_gloo_one_gpu_box.err
