# A/B of k_chain_group's prefetched opening (HB_PF=0/1, same library): parity tests, headline both ways, the phase timeline of a -DHB_STAMPS=1 variant
cd /root/repo
O=gpurun_out
python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py tests/test_gpu_recovery.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for v in 0 1 0 1; do
  HB_PF=$v python bench.py --steps 200 --warmup 100 --no-cpu --no-ab --secondary '' --tertiary '' > $O/r5_pf_$v.json 2> $O/r5_pf_$v.err
  python - <<PY
import json
d=json.loads(open('$O/r5_pf_$v.json').read().strip().splitlines()[-1])
print('HB_PF=$v: value %.1f (launch %.2f us in situ)' % (d['value'], d['roofline']['avg_launch_ms']*1e3))
PY
done
for v in 0 1; do
  HB_PF=$v HIBAYES_GPU_LIB=$PWD/build/variants/stamps.so GT_BITS=2 python tools/group_timeline.py BayesCpi 300 > $O/r5_group_phases_pf_$v.txt 2>&1; tail -14 $O/r5_group_phases_pf_$v.txt | head -8
done
