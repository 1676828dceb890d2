cd /root/repo
O=gpurun_out
python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for tl in "" 900 ""; do
  HB_T=$tl; if [ -n "$tl" ]; then export HB_DOTQ2_TILES=$tl; else unset HB_DOTQ2_TILES; fi
  python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_tiles.json 2> $O/r5_tiles.err
  python - <<PY
import json
d=json.loads(open('$O/r5_tiles.json').read().strip().splitlines()[-1])
r=d['roofline']
print('HB_DOTQ2_TILES=${HB_T:-unset}: value %.1f (launch %.2f us in situ, %.2f isolated; frac %.3f)' % (d['value'], r['avg_launch_ms']*1e3, r['isolated']['avg_launch_ms']*1e3, r['frac']))
PY
done
