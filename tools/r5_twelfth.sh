#!/bin/bash
# round 5, twelfth GPU session: the chain's phases with the compact band on / off (stamped build)
cd /root/repo
O=gpurun_out
for g in 1 0; do
  GT_BITS=2 HB_GRAM16=$g python tools/group_timeline.py BayesCpi 300 > $O/r5_group_phases_g16_$g.txt 2>&1; tail -14 $O/r5_group_phases_g16_$g.txt
done
