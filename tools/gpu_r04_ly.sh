#!/bin/bash
# round 4: layered engine A/B - bit-exactness tests, then the rate at C2
TAG=${1:-r04ly}
mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layered" 2>&1 | tail -3
timeout 300 python tools/layered_rate.py 65536 2>&1 | grep "layered-10" | tee gpurun_out/layered_rate_$TAG.txt
