#!/bin/bash
# round 5, fourth GPU session: what the closing atomics of a mat-vec tile cost (timing-only builds: -DQ2_DIAG=1 none, =3 plain stores), by kernel and tile count;
# the host's CPU quota (why 64 spinning threads of the CPU baseline run slower than 32)
cd /root/repo
O=gpurun_out
( for kind in 2 0; do for v in base q2diag1 q2diag3; do for tiles in 600 1000 2000 3000; do
    lib=build/variants/$v.so; [ $v = base ] && lib=hibayes_amd/libhibayes_gpu.so
    echo -n "kind $kind $v tiles $tiles: "; HIBAYES_GPU_LIB=$PWD/$lib HB_MV_BITS=2 HB_DOTQ2_KIND=$kind HB_DOTQ2_TILES=$tiles python tools/matvec_only.py 50000 500000 2 3 2>&1 | tail -1 | sed 's/precise=2 bits=2: 140 launches of 3584 columns, //'
done; done; done
for v in base q2diag1; do lib=build/variants/$v.so; [ $v = base ] && lib=hibayes_amd/libhibayes_gpu.so
  echo -n "int8 k_dotq $v: "; HIBAYES_GPU_LIB=$PWD/$lib python tools/matvec_only.py 50000 500000 2 3 2>&1 | tail -1; done ) 2>&1 | tee $O/r5_atomics_cost.txt
( cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpuset.cpus.effective; lscpu | grep -E "NUMA|Thread|Core|Socket|^CPU\(s\)"; nproc ) > $O/r5_host_cpu.txt 2>&1; cat $O/r5_host_cpu.txt
