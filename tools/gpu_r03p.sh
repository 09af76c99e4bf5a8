#!/bin/bash
TAG=${1:-r03p}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocprof stats of the default bench"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_$TAG.txt | head -24
echo "== pmc"; bash tools/gpu_pmc.sh $TAG 2>&1 | tail -5
