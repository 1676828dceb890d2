#!/bin/bash
# round 4 counter passes (each in its own rocprofv3 run, --kernel-trace only beside --pmc; outputs under gpurun_out/pmc_*):
#   FETCH_SIZE of the mat-vec launch shapes the sweeps run (int8: 512 / 1024 / 3584 columns; 2-bit: 3584 columns, v_dot4 and matrix-core kernels),
#   SQ counters of k_dotq2, and the v_dot4 issue-rate microbenchmark.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
run_pmc() { # name, counters, env..., then matvec_only args
  name=$1; shift; ctrs=$1; shift
  rm -rf $O/pmc_$name
  env "$@" rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmc_$name -o res -- python $R/tools/matvec_only.py 50000 100000 2 1 > $O/pmc_$name.log 2>&1
  db=$(find $O/pmc_$name -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $db > $O/r04_pmc_$name.txt 2>&1
  grep -E "k_dotq" $O/r04_pmc_$name.txt | head -12
}
run_pmc fetch_int8_d1 FETCH_SIZE HB_MV_BITS=8 HB_TIME_MATVEC_D=1
run_pmc fetch_int8_d2 FETCH_SIZE HB_MV_BITS=8 HB_TIME_MATVEC_D=2
run_pmc fetch_int8_d7 FETCH_SIZE HB_MV_BITS=8 HB_TIME_MATVEC_D=7
run_pmc fetch_2bit_d7 FETCH_SIZE HB_MV_BITS=2 HB_TIME_MATVEC_D=7
run_pmc fetch_2bit_mfma_d7 FETCH_SIZE HB_MV_BITS=2 HB_TIME_MATVEC_D=7 HB_DOTQ2_KIND=2
run_pmc sq_k_dotq2 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" HB_MV_BITS=2 HB_TIME_MATVEC_D=7
run_pmc sq2_k_dotq2 "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" HB_MV_BITS=2 HB_TIME_MATVEC_D=7
run_pmc sq_k_dotq2m "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" HB_MV_BITS=2 HB_TIME_MATVEC_D=7 HB_DOTQ2_KIND=2
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/dot4_rate.hip -o /tmp/dot4_rate && /tmp/dot4_rate > $O/r04_dot4_rate.txt 2>&1; cat $O/r04_dot4_rate.txt
