#!/bin/bash
# round-5 closing run on the final build (certified violation check on): GPU suite, bench lines, chain phases, kernel trace of the headline leg
cd /root/repo
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/r05_gpu_tests.txt 2>&1
grep -n "passed\|failed" $O/r05_gpu_tests.txt | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_driver_args.json 2> $O/r05_bench_driver_args.err
timeout 900 python bench.py > $O/r05_bench_default.json 2> $O/r05_bench_default.err
python - <<'PY'
import json
for f in ('r05_bench_driver_args', 'r05_bench_default'):
    d = json.loads(open('gpurun_out/%s.json' % f).read().strip().splitlines()[-1])
    r = d['roofline']
    print(f, 'value %.1f [%s, frac %.3f, launch %.2f us]' % (d['value'], r['kernel'], r['frac'], r['avg_launch_ms'] * 1e3), 'vdot4', round(r.get('vdot4_value', 0), 1), 'int8', round(r.get('int8_value', 0), 1), 'frac', round(r.get('int8_frac', 0), 3),
          'BayesR', round(r.get('secondary_value', 0), 1), 'converged', round(r.get('secondary_converged_value', 0), 1), [(k, round(v, 1)) for k, v in r.items() if k.startswith('all_move')],
          'cpu', {k: round(v, 3) for k, v in d['cpu_baseline']['by_threads'].items()}, 'regimes', d['regime'], r.get('int8_regime'), r.get('vdot4_regime'))
PY
GT_BITS=2 python tools/group_timeline.py BayesCpi 300 > $O/r05_group_phases_final.txt 2>&1; tail -14 $O/r05_group_phases_final.txt | head -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $O/trace_f
rocprofv3 --kernel-trace --stats -d $O/trace_f -o bench -- python $R/bench.py --steps 100 --warmup 30 --no-ab --no-cpu --secondary "" --tertiary "" > $O/r05_bench_under_rocprof_2bit_mfma.json 2> $O/trace_f.err
db=$(find $O/trace_f -name "*.db" | head -1)
python $R/tools/rocprof_window.py $db --after 11 --sweeps 100 > $O/r05_kernel_trace_timed_window_2bit_mfma.txt 2>&1
head -6 $O/r05_kernel_trace_timed_window_2bit_mfma.txt | cut -c1-170
rm -rf $O/trace_f
