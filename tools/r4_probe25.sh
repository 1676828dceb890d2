#!/bin/bash
for cfg in "0 4" "4 2" "4 4" "4 8" "8 4" "2 3"; do
set -- $cfg
echo "== HB_WARM_R=$1 HB_WARM_AHEAD=$2"
HB_WARM_R=$1 HB_WARM_AHEAD=$2 timeout 300 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 60 2>&1 | tail -1
done
