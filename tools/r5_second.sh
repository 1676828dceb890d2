#!/bin/bash
# round 5, second GPU session: which wave the chain's first barrier waits for (per-wave stamps), the hand-off words in uncached /
# fine-grained memory (speed and the dense stall), the CPU baseline's team on the box's cores
cd /root/repo
O=gpurun_out
GT_BITS=2 python tools/group_timeline.py BayesCpi 300 > $O/r5_group_timeline_waves.txt 2>&1
tail -22 $O/r5_group_timeline_waves.txt
python tools/cpu_baseline_probe.py 3000 > $O/r5_cpu_team.txt 2>&1; cat $O/r5_cpu_team.txt
for k in 0 1 2; do
  HB_HANDOFF_ALLOC=$k python bench.py --steps 100 --warmup 50 --no-cpu --tertiary BayesRR --burnin-converged 0 > $O/r5_bench_handoff$k.json 2> $O/r5_bench_handoff$k.err
  python - <<PY
import json
d=json.loads(open('$O/r5_bench_handoff$k.json').read().strip().splitlines()[-1])
print('handoff alloc $k: value', round(d['value'],1), 'int8', round(d['int8']['value'],1), 'mfma', round(d['mfma_ab']['value'],1), 'BayesR', round(d['secondary']['value'],1), [(t['model'], round(t['value'],1)) for t in d['all_move']])
PY
done
HB_HANDOFF_ALLOC=1 timeout 700 python tools/soak.py dense rr 10000 > $O/r5_dense_soak_uncached.txt 2>&1; tail -2 $O/r5_dense_soak_uncached.txt
