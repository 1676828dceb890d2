# headline under other look-ahead depths (the chain got 12 % faster in round 5: is three groups still the right depth?)
cd /root/repo
O=gpurun_out
for g in 1,3,7 1,2,7 1,3,7 1,2,7; do
  HB_BENCH_GEO_BayesCpi=$g python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_geo.json 2> $O/r5_geo.err
  python - <<PY
import json
d=json.loads(open('$O/r5_geo.json').read().strip().splitlines()[-1])
print('geometry $g: value %.1f (launch %.2f us in situ) %s' % (d['value'], d['roofline']['avg_launch_ms']*1e3, d['config'].get('geometry_of_the_timed_region')))
PY
done
