# the mat-vec launch (784 tiles + 196 update + 56 finalize blocks = 1036) does not fit the device's 1016 block slots (4 per compute unit at 37 KB of LDS,
# one compute unit gone to the chain, part of one to k_fwd; blocks are dealt round-robin over the XCDs): its last tiles start when the first ones end.
# Fewer, longer tiles:
cd /root/repo
O=gpurun_out
for tl in ${TILES_LIST:-900 730 680 730 900 620}; do
  HB_DOTQ2_TILES=$tl python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_tiles.json 2> $O/r5_tiles.err
  python - <<PY
import json
d=json.loads(open('$O/r5_tiles.json').read().strip().splitlines()[-1])
r=d['roofline']
print('HB_DOTQ2_TILES=$tl: value %.1f (launch %.2f us in situ, %.2f isolated; frac %.3f)' % (d['value'], r['avg_launch_ms']*1e3, r['isolated']['avg_launch_ms']*1e3, r['frac']))
PY
done
