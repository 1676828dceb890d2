#!/bin/bash
# round 5, tenth GPU session: k_dotq2m as the library default for the 2-bit layout; the drift pre-check restricted to near markers (phases, parity, bench A/B)
cd /root/repo
O=gpurun_out
GT_BITS=2 HB_DRIFT=1 python tools/group_timeline.py BayesCpi 300 > $O/r5_group_phases_k_dotq2m_drift2.txt 2>&1; tail -14 $O/r5_group_phases_k_dotq2m_drift2.txt
HB_DRIFT=1 python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | tee $O/r5_drift2_tests.txt
for dr in 1 0; do
  HB_DRIFT=$dr python bench.py --steps 200 --warmup 100 --no-cpu --secondary '' --tertiary '' > $O/r5_default_q2m_drift$dr.json 2> $O/r5_default_q2m_drift$dr.err
  python - <<PY
import json
d=json.loads(open('$O/r5_default_q2m_drift$dr.json').read().strip().splitlines()[-1])
print('drift check $dr: value %.1f [%s] (redo %.1f, launch %.2f us in situ, %.2f isolated) vdot4 %.1f (redo %.1f, launch %.2f us) int8 %.1f (redo %.1f)' % (d['value'], d['roofline']['kernel'], d['config']['chain_rounds_rolled_back_per_sweep'],
      d['roofline']['avg_launch_ms']*1e3, d['roofline']['isolated']['avg_launch_ms']*1e3, d['vdot4_ab']['value'], d['vdot4_ab']['chain_rounds_rolled_back_per_sweep'], d['vdot4_ab']['roofline']['avg_launch_ms']*1e3, d['int8']['value'], d['int8']['chain_rounds_rolled_back_per_sweep']))
PY
done 2>&1 | tee $O/r5_default_q2m.txt
