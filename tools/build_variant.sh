#!/bin/bash
# tools/build_variant.sh NAME "-DHB_X=.. ..."  ->  build/variants/NAME.so (git-ignored; delete it when the A/B is done: what sits under build/ travels to the GPU box) (A/B builds; select with HIBAYES_GPU_LIB)
set -e
cd "$(dirname "$0")/../hibayes_amd/csrc"
mkdir -p ../../build/variants /tmp/hbv_$1
for f in hb_ctx hb_kernels hb_gram hb_run hb_comm hb_sbayes; do
  if [ $f = hb_kernels ] || [ ! -f $f.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result $2 -c $f.hip -o /tmp/hbv_$1/$f.o
  else cp $f.o /tmp/hbv_$1/$f.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/$1.so /tmp/hbv_$1/*.o -ldl
echo built build/variants/$1.so
