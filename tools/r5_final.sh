#!/bin/bash
# round-5 closing run: the whole GPU suite, the bench line under the driver's arguments and under the defaults, on the committed build
cd /root/repo
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/r05_gpu_tests.txt 2>&1
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
