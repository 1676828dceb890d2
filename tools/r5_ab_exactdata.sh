cd /root/repo
O=gpurun_out
python -m pytest tests/test_gpu_depth.py -m gpu -x -q -k "default_geometry or long_chain or certified" 2>&1 | tail -2
for v in base head base head; do lib=build/variants/$v.so; [ $v = base ] && lib=hibayes_amd/libhibayes_gpu.so
  HIBAYES_GPU_LIB=$PWD/$lib python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_xd_$v.json 2> $O/r5_xd_$v.err
  python - <<PY
import json
d=json.loads(open('$O/r5_xd_$v.json').read().strip().splitlines()[-1])
print('$v: value %.1f (launch %.2f us in situ)' % (d['value'], d['roofline']['avg_launch_ms']*1e3))
PY
done
