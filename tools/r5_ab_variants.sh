# A/B of library variants under build/variants (HIBAYES_GPU_LIB): parity tests on the working build, then the headline of each variant, twice
# usage: tools/r5_ab_variants.sh v1 v2 ...   ("base" = hibayes_amd/libhibayes_gpu.so)
cd /root/repo
O=gpurun_out
python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2; do for v in "$@"; do lib=build/variants/$v.so; [ $v = base ] && lib=hibayes_amd/libhibayes_gpu.so
  HIBAYES_GPU_LIB=$PWD/$lib python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_v_$v.json 2> $O/r5_v_$v.err
  python - <<PY
import json
d=json.loads(open('$O/r5_v_$v.json').read().strip().splitlines()[-1])
print('$v: value %.1f (launch %.2f us in situ)' % (d['value'], d['roofline']['avg_launch_ms']*1e3))
PY
done; done
