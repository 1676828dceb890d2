#!/bin/bash
# round 5, seventh GPU session: the matrix-core mat-vec on 512-individual stages with whole-line DMA pieces (shape G = 0)
cd /root/repo
O=gpurun_out
( for shape in "0 1" "0 0" "1 1"; do set -- $shape
    echo "== G $1 SC $2"
    HB_Q2M_G=$1 HB_Q2M_SC=$2 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "two_bit_layout" 2>&1 | tail -1
    HB_Q2M_G=$1 HB_Q2M_SC=$2 python -m pytest tests/test_gpu_depth.py -m gpu -x -q -k "matrix_core" 2>&1 | tail -1
    for tiles in 300 450 600 900 1400; do
      echo -n "   tiles $tiles: "; HB_Q2M_G=$1 HB_Q2M_SC=$2 HB_MV_BITS=2 HB_DOTQ2_KIND=2 HB_DOTQ2_TILES=$tiles python tools/matvec_only.py 50000 500000 2 3 2>&1 | tail -1 | sed 's/precise=2 bits=2: 140 launches of 3584 columns, //'
    done
done ) 2>&1 | tee $O/r5_q2m_512.txt
