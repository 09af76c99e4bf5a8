#!/bin/bash
# layered engine with register-resident item state: parity, rate, ablations, per-record timeline
TAG=${1:-r03x}
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layered" 2>&1 | tail -3
timeout 600 python tools/layered_rate.py 16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/layered_rate_$TAG.txt
for v in "SAMD_LY_ABL=1" "SAMD_LY_ABL=2" "SAMD_LY_ABL=3"; do
  echo "[$v] $(env $v timeout 300 python tools/layered_rate.py 16384 2>&1 | grep 'layered-10   minsum')"
done | tee gpurun_out/layered_abl_$TAG.txt
SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_lytrace.so timeout 300 python tools/ly_itrace.py minsum > gpurun_out/ly_itrace_${TAG}_new.txt 2>&1; tail -3 gpurun_out/ly_itrace_${TAG}_new.txt
