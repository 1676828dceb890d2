#!/bin/bash
# round 5, session 17: candidate margin applied only when a round is repeated anyway (built in: none; variants 0.95, 0.9); the failing compact-band test in full
cd /root/repo
O=gpurun_out
python -m pytest tests/test_gpu_depth.py -m gpu -x -q -k "certified_check" 2>&1 | tail -30 | cut -c1-250 | tee $O/r5_cert3_tests.txt
for v in base cert_m95 cert_m90 base; do lib=build/variants/$v.so; [ $v = base ] && lib=hibayes_amd/libhibayes_gpu.so
  HIBAYES_GPU_LIB=$PWD/$lib python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_cert3_$v.json 2> $O/r5_cert3_$v.err
  python - <<PY
import json
d=json.loads(open('$O/r5_cert3_$v.json').read().strip().splitlines()[-1])
print('$v: value %.1f [%s] (redo %.1f, launch %.2f us in situ, %.2f isolated)' % (d['value'], d['roofline']['kernel'], d['config']['chain_rounds_rolled_back_per_sweep'],
      d['roofline']['avg_launch_ms']*1e3, d['roofline']['isolated']['avg_launch_ms']*1e3))
PY
done 2>&1 | tee $O/r5_cert3.txt
