#!/bin/bash
TAG=${1:-r03s}
mkdir -p gpurun_out
for v in "" "SAMD_LY_ABL=1" "SAMD_LY_ABL=2" "SAMD_LY_ABL=3"; do
  echo "[$v] $(env $v timeout 300 python tools/layered_rate.py 16384 2>&1 | grep 'layered-10   minsum')"
done | tee gpurun_out/layered_abl_$TAG.txt
