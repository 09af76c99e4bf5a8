#!/bin/bash
TAG=${1:-r03o}
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layered or minsum" 2>&1 | tail -3
echo "== layered rate"
for v in "" "SAMD_LY_CN_SPLIT=0" "SAMD_LY_VN_SINGLE_MAX=0" "SAMD_LY_VN_SINGLE_MAX=16" "SAMD_LY_VN_SINGLE_MAX=32" "SAMD_LY_NOGROUP=1"; do
  echo "[$v] $(env $v timeout 300 python tools/layered_rate.py 16384 2>&1 | grep 'layered-10   minsum')"
done | tee gpurun_out/layered_rate_$TAG.txt
echo "== flooding LDS-only barrier"; timeout 300 python tools/ms_ab.py --cn minsum base: ldsbar:SAMD_MS_LDSBAR=1 base2: ldsbar2:SAMD_MS_LDSBAR=1 2>&1 | tail -4 | tee gpurun_out/ms_ldsbar_$TAG.txt
