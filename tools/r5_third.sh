#!/bin/bash
# round 5, third GPU session: the chain's candidate margin (HB_CANDF: rolled-back rounds against extra candidates), and the matrix-core
# mat-vec's buffering depth x tile count in isolation
cd /root/repo
O=gpurun_out
for cf in 1.0 0.8 0.6 0.45 0.3; do
  HB_CANDF=$cf python bench.py --steps 100 --warmup 50 --no-cpu --secondary '' --tertiary '' > $O/r5_candf_$cf.json 2> $O/r5_candf_$cf.err
  python - <<PY
import json
d=json.loads(open('$O/r5_candf_$cf.json').read().strip().splitlines()[-1])
print('candf $cf: value %.1f (redo %.1f, moves %.0f) mfma %.1f (redo %.1f) int8 %.1f (redo %.1f)' % (d['value'], d['config']['chain_rounds_rolled_back_per_sweep'], d['config']['mean_changed_markers_per_sweep'],
      d['mfma_ab']['value'], d['mfma_ab']['chain_rounds_rolled_back_per_sweep'], d['int8']['value'], d['int8']['chain_rounds_rolled_back_per_sweep']))
PY
done 2>&1 | tee $O/r5_candf.txt
( for nb in 3 4 5 6; do
  for tiles in 600 800 1100 1400 1800 2400; do
    lib=build/variants/q2m_nbuf$nb.so; [ $nb = 3 ] && lib=hibayes_amd/libhibayes_gpu.so
    echo -n "NBUF $nb tiles $tiles: "; HIBAYES_GPU_LIB=$PWD/$lib HB_MV_BITS=2 HB_DOTQ2_KIND=2 HB_DOTQ2_TILES=$tiles python tools/matvec_only.py 50000 500000 2 3 2>&1 | tail -1
  done
done ) 2>&1 | tee $O/r5_q2m_depth.txt
