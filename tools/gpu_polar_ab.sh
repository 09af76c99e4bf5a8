#!/bin/bash
# One GPU-box trip for the Polar list decoder: parity tests, then the C5 bench line with the register engine and
# with the generic engine (SAMD_SCL_GENERIC=1).  Outputs in gpurun_out/.
TAG=${1:-r02}
mkdir -p gpurun_out
echo "== pytest polar"; timeout 900 python -m pytest tests/test_gpu_polar.py -x -q 2>&1 | tail -15
for g in ${GS:-5 4 3}; do
echo "== c5 register engine, G=$g"; SAMD_SCL_GSTAGES=$g timeout 600 python bench.py --workload c5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | tee gpurun_out/c5_reg_g${g}_$TAG.json | cut -c1-400
done
echo "== c5 generic engine"; SAMD_SCL_GENERIC=1 timeout 600 python bench.py --workload c5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | tee gpurun_out/c5_generic_$TAG.json | cut -c1-400
