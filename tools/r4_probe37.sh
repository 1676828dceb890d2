#!/bin/bash
for lib in "" "$PWD/build/variants/f48.so" "$PWD/build/variants/sb8.so" ""; do
echo "== $lib"
HIBAYES_GPU_LIB=$lib timeout 500 python tools/geo_sweep.py 50000 500000 BayesR 300 512 "2,1" 60 2>&1 | tail -1
done
