#!/bin/bash
# round 3, trip b: VALU issue rates (incl. v_bitop3), A/B of the grouped-dispatch kernel variants, parity of the new default
TAG=${1:-r03b}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== valu_rate"; timeout 120 tools/ubench/valu_rate 2>&1 | tee gpurun_out/valu_rate_$TAG.txt
echo "== A/B min-sum"; timeout 600 python tools/ms_ab.py --cn minsum --soft --out gpurun_out/ms_ab_$TAG.json \
  old:SAMD_MS_NOGROUP=1 grouped_generic:SAMD_MS_NOZ128=1 grouped_z128: grouped_z128_bitop3:SAMD_MS_VAR=1 \
  grouped_z128_norot:SAMD_MS_NOROT=1 grouped_z128_bitop3_norot:SAMD_MS_VAR=1,SAMD_MS_NOROT=1 grouped_z128_noprio:SAMD_MS_NOPRIO=1 2>&1 | tail -8
echo "== A/B phi"; timeout 600 python tools/ms_ab.py --cn boxplus-phi --soft --batch 16384 --out gpurun_out/phi_ab_$TAG.json \
  old:SAMD_MS_NOGROUP=1 grouped_z128: 2>&1 | tail -3
echo "== A/B phi fast"; timeout 600 python tools/ms_ab.py --cn boxplus-phi-fast --soft --batch 16384 --out gpurun_out/phifast_ab_$TAG.json \
  old:SAMD_MS_NOGROUP=1 grouped_z128: 2>&1 | tail -3
echo "== parity (ldpc)"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_idd.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -4
echo "== parity with bitop3"; SAMD_MS_VAR=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "minsum or random_codes" 2>&1 | tail -3
