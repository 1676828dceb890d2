#!/bin/bash
# round 5, ninth GPU session: the sweep with the new default shape of k_dotq2m; the chain's phases accumulated over all rounds (stamped build), beside both 2-bit kernels and with the drift pre-check
cd /root/repo
O=gpurun_out
python bench.py --steps 100 --warmup 50 --no-cpu --secondary '' --tertiary '' > $O/r5_bench_q2m512.json 2> $O/r5_bench_q2m512.err
python - <<PY
import json
d=json.loads(open('$O/r5_bench_q2m512.json').read().strip().splitlines()[-1])
print('value %.1f (launch %.2f us) mfma %.1f (launch %.2f us in situ, %.2f isolated) int8 %.1f' % (d['value'], d['roofline']['avg_launch_ms']*1e3, d['mfma_ab']['value'], d['mfma_ab']['roofline']['avg_launch_ms']*1e3, d['mfma_ab']['roofline']['isolated']['avg_launch_ms']*1e3, d['int8']['value']))
PY
GT_BITS=2 python tools/group_timeline.py BayesCpi 300 > $O/r5_group_phases_k_dotq2.txt 2>&1; tail -16 $O/r5_group_phases_k_dotq2.txt
GT_BITS=2 HB_DOTQ2_KIND=2 python tools/group_timeline.py BayesCpi 300 > $O/r5_group_phases_k_dotq2m.txt 2>&1; tail -16 $O/r5_group_phases_k_dotq2m.txt
GT_BITS=2 HB_DOTQ2_KIND=2 HB_DRIFT=1 python tools/group_timeline.py BayesCpi 300 > $O/r5_group_phases_k_dotq2m_drift.txt 2>&1; tail -16 $O/r5_group_phases_k_dotq2m_drift.txt
tools/lost_store 150 1 0 2>&1 | tee $O/r5_lost_store_long.txt
