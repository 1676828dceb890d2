#!/bin/bash
# round 5, eighth GPU session: the 512-row matrix-core tile with bank-conflict-free DMA lane order (G = 3) against the plain one (G = 0), depth 3 / 4;
# the lost-store reproducer
cd /root/repo
O=gpurun_out
( for g in 3 0; do
    echo "== G $g"
    HB_Q2M_G=$g python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "two_bit_layout" 2>&1 | tail -1
    HB_Q2M_G=$g python -m pytest tests/test_gpu_depth.py -m gpu -x -q -k "matrix_core" 2>&1 | tail -1
    for v in base q2m_nbuf4; do lib=build/variants/$v.so; [ $v = base ] && lib=hibayes_amd/libhibayes_gpu.so
    for tiles in 700 900 1100; do
      echo -n "   $v tiles $tiles: "; HIBAYES_GPU_LIB=$PWD/$lib HB_Q2M_G=$g HB_MV_BITS=2 HB_DOTQ2_KIND=2 HB_DOTQ2_TILES=$tiles python tools/matvec_only.py 50000 500000 2 3 2>&1 | tail -1 | sed 's/precise=2 bits=2: 140 launches of 3584 columns, //'
    done; done
done ) 2>&1 | tee $O/r5_q2m_512b.txt
( tools/lost_store 20 1 0; tools/lost_store 20 0 0; tools/lost_store 20 1 1 ) 2>&1 | tee $O/r5_lost_store.txt
