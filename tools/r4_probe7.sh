#!/bin/bash
# round 4: where a launch's in-situ time goes (block roles) for the v_dot4 and the matrix-core kernel, the 128-row stage shape, counter passes
O=gpurun_out
timeout 300 python tools/launch_roles.py 2 3 > $O/r4_roles_valu.txt 2>&1; tail -7 $O/r4_roles_valu.txt
HB_DOTQ2_KIND=2 timeout 300 python tools/launch_roles.py 2 3 > $O/r4_roles_mfma.txt 2>&1; tail -7 $O/r4_roles_mfma.txt
for cfg in "0 256 3072" "0 128 3072" "0 128 6000" "0 128 1536"; do
  set -- $cfg
  echo "== kind=$1 rs=$2 tiles=$3"
  HB_MV_BITS=2 HB_DOTQ2_KIND=$1 HB_DOTQ2_RS=$2 HB_DOTQ2_TILES=$3 timeout 300 python tools/matvec_only.py 50000 100000 2 5 2>&1 | tail -1
done > $O/r4_matvec_variants4.log 2>&1
cat $O/r4_matvec_variants4.log
bash tools/r4_pmc.sh > $O/r4_pmc.log 2>&1; tail -40 $O/r4_pmc.log
