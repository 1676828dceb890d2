#!/bin/bash
# round 5, session 16: certificate with DPP scans and a candidate margin (0.9 built in; variants 1.0 and 0.8): parity, phases, sweeps/s
cd /root/repo
O=gpurun_out
python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py -m gpu -x -q 2>&1 | tail -3 | tee $O/r5_cert2_tests.txt
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3_bayescpi or stationary or config2 or config5 or config4" 2>&1 | tail -2 | tee -a $O/r5_cert2_tests.txt
GT_BITS=2 python tools/group_timeline.py BayesCpi 300 > $O/r5_group_phases_cert2.txt 2>&1; tail -14 $O/r5_group_phases_cert2.txt | head -6
for v in base cert_m100 cert_m80 base; do lib=build/variants/$v.so; [ $v = base ] && lib=hibayes_amd/libhibayes_gpu.so
  HIBAYES_GPU_LIB=$PWD/$lib python bench.py --steps 200 --warmup 100 --no-cpu --secondary '' --tertiary '' > $O/r5_cert2_$v.json 2> $O/r5_cert2_$v.err
  python - <<PY
import json
d=json.loads(open('$O/r5_cert2_$v.json').read().strip().splitlines()[-1])
print('$v: value %.1f [%s] (redo %.1f, launch %.2f us in situ, %.2f isolated) vdot4 %.1f (launch %.2f us) int8 %.1f (launch %.2f)' % (d['value'], d['roofline']['kernel'], d['config']['chain_rounds_rolled_back_per_sweep'],
      d['roofline']['avg_launch_ms']*1e3, d['roofline']['isolated']['avg_launch_ms']*1e3, d['vdot4_ab']['value'], d['vdot4_ab']['roofline']['avg_launch_ms']*1e3, d['int8']['value'], d['int8']['roofline']['avg_launch_ms']*1e3))
PY
done 2>&1 | tee $O/r5_cert2.txt
