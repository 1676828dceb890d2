#!/bin/bash
# round 6 counter pass: FETCH_SIZE of the two launch shapes the bench line's roofline is quoted on (2-bit matrix-core kernel and int8 kernel, 3 584 columns),
# each in its own rocprofv3 run (--kernel-trace only beside --pmc), n = 50 000; outputs profiles-ready under gpurun_out/r06_pmc_*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
run_pmc() { # name, counters, env..., then matvec_only args
  name=$1; shift; ctrs=$1; shift
  rm -rf $O/pmc_$name
  env "$@" rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmc_$name -o res -- python $R/tools/matvec_only.py 50000 100000 2 1 > $O/pmc_$name.log 2>&1
  db=$(find $O/pmc_$name -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $db > $O/r06_pmc_$name.txt 2>&1
  grep -E "k_dotq" $O/r06_pmc_$name.txt | head -6
  rm -rf $O/pmc_$name
}
run_pmc fetch_2bit_mfma_d7 FETCH_SIZE HB_MV_BITS=2 HB_TIME_MATVEC_D=7 HB_DOTQ2_KIND=2
run_pmc fetch_int8_d7 FETCH_SIZE HB_MV_BITS=8 HB_TIME_MATVEC_D=7
