#!/bin/bash
TAG=${1:-r03x2}
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layered" 2>&1 | tail -3
for v in "" "SAMD_LY_NOSPLIT=1"; do
  echo "[$v] $(env $v timeout 300 python tools/layered_rate.py 16384 2>&1 | grep 'layered-10   ' | tr '\n' ' ')"
done | tee gpurun_out/layered_abl_$TAG.txt
