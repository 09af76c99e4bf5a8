#!/bin/bash
# Where does an iteration of the C2 min-sum engine go?  Times the ablation builds (make -C sionna_amd/csrc ablation):
# parts of the kernel removed at compile time (wrong results), everything else unchanged.
TAG=${1:-r03c}
mkdir -p gpurun_out
OUT=gpurun_out/ablation_$TAG.txt
: > $OUT
for a in 0 1 2 4 6 8 14; do
  lib=$PWD/sionna_amd/lib/libsionna_amd_abl$a.so
  [ $a = 0 ] && lib=$PWD/sionna_amd/lib/libsionna_amd.so
  for grp in 0 1; do
    env="x:"; [ $grp = 1 ] && env="x:SAMD_MS_NOGROUP=1"
    r=$(SAMD_LIB=$lib timeout 300 python tools/ms_ab.py --cn minsum $env 2>&1 | tail -1)
    echo "abl=$a nogroup=$grp  $r" | tee -a $OUT
  done
done
