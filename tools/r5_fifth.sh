#!/bin/bash
# round 5, fifth GPU session: the matrix-core mat-vec's new shapes (column tiles per wave x paired stages x accumulator sets): exactness, then isolated timing
cd /root/repo
O=gpurun_out
( for shape in "4 1 1" "8 1 1" "8 1 0" "8 2 1" "8 2 0" "16 1 0" "16 1 1" "4 2 1" "4 1 0"; do set -- $shape
    echo "== CT $1 G $2 SC $3"
    HB_Q2M_CT=$1 HB_Q2M_G=$2 HB_Q2M_SC=$3 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "two_bit_layout" 2>&1 | tail -1
    HB_Q2M_CT=$1 HB_Q2M_G=$2 HB_Q2M_SC=$3 python -m pytest tests/test_gpu_depth.py -m gpu -x -q -k "matrix_core" 2>&1 | tail -1
    for tiles in 300 450 600 900; do
      echo -n "   tiles $tiles: "; HB_Q2M_CT=$1 HB_Q2M_G=$2 HB_Q2M_SC=$3 HB_MV_BITS=2 HB_DOTQ2_KIND=2 HB_DOTQ2_TILES=$tiles python tools/matvec_only.py 50000 500000 2 3 2>&1 | tail -1 | sed 's/precise=2 bits=2: 140 launches of 3584 columns, //'
    done
done ) 2>&1 | tee $O/r5_q2m_shapes.txt
