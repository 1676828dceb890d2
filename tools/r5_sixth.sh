#!/bin/bash
# round 5, sixth GPU session: (a) the matrix-core mat-vec without the non-temporal hint on its genotype pieces (does the second 64-byte half of a line
# come from L2 then?); (b) the drift pre-check of k_chain_group: parity, rolled-back rounds, sweeps/s
cd /root/repo
O=gpurun_out
( for v in base q2m_nt0; do lib=build/variants/$v.so; [ $v = base ] && lib=hibayes_amd/libhibayes_gpu.so
  for shape in "4 1" "4 2" "8 1"; do set -- $shape
   for tiles in 450 600 900; do
    echo -n "$v CT $1 G $2 tiles $tiles: "; HIBAYES_GPU_LIB=$PWD/$lib HB_Q2M_CT=$1 HB_Q2M_G=$2 HB_MV_BITS=2 HB_DOTQ2_KIND=2 HB_DOTQ2_TILES=$tiles python tools/matvec_only.py 50000 500000 2 3 2>&1 | tail -1 | sed 's/precise=2 bits=2: 140 launches of 3584 columns, //'
   done; done; done ) 2>&1 | tee $O/r5_q2m_nt.txt
( time python -m pytest tests/test_gpu_depth.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 ) 2>&1 | tee $O/r5_drift_tests.txt
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3_bayescpi or stationary or config2 or config5" 2>&1 | tail -2 | tee -a $O/r5_drift_tests.txt
for dr in 1 0; do
  HB_DRIFT=$dr python bench.py --steps 100 --warmup 50 --no-cpu --secondary '' --tertiary '' > $O/r5_drift_$dr.json 2> $O/r5_drift_$dr.err
  python - <<PY
import json
d=json.loads(open('$O/r5_drift_$dr.json').read().strip().splitlines()[-1])
print('drift check $dr: value %.1f (redo %.1f, moves %.0f) mfma %.1f (redo %.1f, launch %.2f us) int8 %.1f (redo %.1f)' % (d['value'], d['config']['chain_rounds_rolled_back_per_sweep'], d['config']['mean_changed_markers_per_sweep'],
      d['mfma_ab']['value'], d['mfma_ab']['chain_rounds_rolled_back_per_sweep'], d['mfma_ab']['roofline']['avg_launch_ms']*1e3, d['int8']['value'], d['int8']['chain_rounds_rolled_back_per_sweep']))
PY
done 2>&1 | tee $O/r5_drift.txt
