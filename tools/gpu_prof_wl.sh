#!/bin/bash
# rocprofv3 kernel statistics of one bench workload: bash tools/gpu_prof_wl.sh <tag> <workload>
TAG=$1; WL=$2
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$WL -o s -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$WL.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_${TAG}_$WL -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_${TAG}_$WL.txt | head -24
