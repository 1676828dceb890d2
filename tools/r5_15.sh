#!/bin/bash
# round 5, session 15: the certified violation check of k_chain_group: parity (bit for bit against the plain path), phases, sweeps/s on / off
cd /root/repo
O=gpurun_out
python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py -m gpu -x -q 2>&1 | tail -6 | tee $O/r5_cert_tests.txt
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3_bayescpi or stationary or config2 or config5 or config4" 2>&1 | tail -2 | tee -a $O/r5_cert_tests.txt
for g in 1 0; do
  GT_BITS=2 HB_CERT=$g python tools/group_timeline.py BayesCpi 300 > $O/r5_group_phases_cert_$g.txt 2>&1; tail -14 $O/r5_group_phases_cert_$g.txt | head -8
done
for g in 1 0; do
  HB_CERT=$g python bench.py --steps 200 --warmup 100 --no-cpu --secondary '' --tertiary '' > $O/r5_cert_$g.json 2> $O/r5_cert_$g.err
  python - <<PY
import json
d=json.loads(open('$O/r5_cert_$g.json').read().strip().splitlines()[-1])
print('cert $g: value %.1f [%s] (redo %.1f, launch %.2f us in situ, %.2f isolated) vdot4 %.1f (launch %.2f us) int8 %.1f (launch %.2f)' % (d['value'], d['roofline']['kernel'], d['config']['chain_rounds_rolled_back_per_sweep'],
      d['roofline']['avg_launch_ms']*1e3, d['roofline']['isolated']['avg_launch_ms']*1e3, d['vdot4_ab']['value'], d['vdot4_ab']['roofline']['avg_launch_ms']*1e3, d['int8']['value'], d['int8']['roofline']['avg_launch_ms']*1e3))
PY
done 2>&1 | tee $O/r5_cert.txt
