#!/bin/bash
# kernel trace of the headline leg's timed window on the final build (certified check on)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for leg in "2bit_mfma --matvec-kernel 2" "2bit_vdot4 --matvec-kernel 0"; do set -- $leg
  rm -rf $O/trace_f
  rocprofv3 --kernel-trace --stats -d $O/trace_f -o bench -- python $R/bench.py --steps 100 --warmup 30 --no-ab --no-cpu --secondary "" --tertiary "" $2 $3 > $O/r05_bench_under_rocprof_$1.json 2> $O/trace_f.err
  db=$(find $O/trace_f -name "*.db" | head -1)
  python $R/tools/rocprof_window.py $db --after 11 --sweeps 100 > $O/r05_kernel_trace_timed_window_$1.txt 2>&1
  head -6 $O/r05_kernel_trace_timed_window_$1.txt | cut -c1-170
  rm -rf $O/trace_f
done
