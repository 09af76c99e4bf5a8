#!/bin/bash
# round 3, first GPU trip: full GPU suite (spec phi, advisor fixes), self-launched 2-rank bench over gloo, default bench
TAG=${1:-r03a}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -22 | tee gpurun_out/pytest_$TAG.txt
echo "== bench --gpus 2 self-launch over gloo"; SAMD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch 4096 --steps 5 --no-cpu-baseline --no-extra --also none 2>gpurun_out/dist2_$TAG.err | grep '^{' | tee gpurun_out/bench_dist2_$TAG.json | cut -c1-700
tail -3 gpurun_out/dist2_$TAG.err
echo "== bench"; timeout 900 python bench.py --no-extra 2>&1 | tail -1 | tee gpurun_out/bench_$TAG.json | cut -c1-2500
