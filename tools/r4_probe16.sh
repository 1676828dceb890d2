#!/bin/bash
O=gpurun_out
export HIBAYES_GPU_LIB=$PWD/build/variants/stamps.so
GT_BITS=2 HB_DOTQ2_KIND=2 timeout 300 python tools/group_timeline.py BayesCpi 300 > $O/r4_group_timeline_mfma.txt 2>&1; tail -18 $O/r4_group_timeline_mfma.txt
GT_BITS=2 timeout 300 python tools/group_timeline.py BayesCpi 300 > $O/r4_group_timeline_valu.txt 2>&1; tail -8 $O/r4_group_timeline_valu.txt
