#!/bin/bash
TAG=${1:-r03g}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ofdm.py -m gpu -x -q 2>&1 | tail -5
echo "== rocprof stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/prof_$TAG.log | cut -c1-300
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_$TAG.txt | head -30
