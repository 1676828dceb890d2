cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/trace_f
rocprofv3 --kernel-trace -d $O/trace_f -o bench -- python $R/bench.py --steps 100 --warmup 30 --no-ab --no-cpu --secondary "" --tertiary "" > $O/bnd.json 2> $O/trace_f.err
db=$(find $O/trace_f -name "*.db" | head -1)
python $R/tools/rocprof_boundary.py $db > $O/r05_sweep_boundary.txt 2>&1
cat $O/r05_sweep_boundary.txt | cut -c1-150
rm -rf $O/trace_f
