cd /root/repo
O=gpurun_out
for w in 0 4 0 4 2 8; do for a in 14; do
  HB_WARM_G=$w HB_WARM_AHEAD=$a python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_warm.json 2> $O/r5_warm.err
  python - <<PY
import json
d=json.loads(open('$O/r5_warm.json').read().strip().splitlines()[-1])
print('HB_WARM_G=$w ahead $a: value %.1f (launch %.2f us in situ)' % (d['value'], d['roofline']['avg_launch_ms']*1e3))
PY
done; done
for a in 7 21; do
  HB_WARM_G=4 HB_WARM_AHEAD=$a python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_warm.json 2> $O/r5_warm.err
  python - <<PY
import json
d=json.loads(open('$O/r5_warm.json').read().strip().splitlines()[-1])
print('HB_WARM_G=4 ahead $a: value %.1f (launch %.2f us in situ)' % (d['value'], d['roofline']['avg_launch_ms']*1e3))
PY
done
