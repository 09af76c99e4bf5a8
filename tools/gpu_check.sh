#!/bin/bash
# One GPU-box trip: smoke, GPU parity tests, bench, rocprof kernel stats.  Outputs in gpurun_out/.
TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_$TAG.json | cut -c1-600
echo "== rocprof stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_$TAG.txt | head -24
