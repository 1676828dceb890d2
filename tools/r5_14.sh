#!/bin/bash
# round 5, session 14: the compact band read one short per lane and row with exact reconstruction (no spills): parity, sweeps/s on / off
cd /root/repo
O=gpurun_out
HB_GRAM16=1 python -m pytest tests/test_gpu_depth.py -m gpu -x -q -k "default_geometry or long_chain or geometry_by_regime or two_bit" 2>&1 | tail -2 | tee $O/r5_g16c_tests.txt
for g in 1 0 1 0; do
  HB_GRAM16=$g python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_g16c_$g.json 2> $O/r5_g16c_$g.err
  python - <<PY
import json
d=json.loads(open('$O/r5_g16c_$g.json').read().strip().splitlines()[-1])
print('gram16 $g: value %.1f [%s] (redo %.1f, launch %.2f us in situ, %.2f isolated)' % (d['value'], d['roofline']['kernel'], d['config']['chain_rounds_rolled_back_per_sweep'],
      d['roofline']['avg_launch_ms']*1e3, d['roofline']['isolated']['avg_launch_ms']*1e3))
PY
done 2>&1 | tee $O/r5_g16c.txt
