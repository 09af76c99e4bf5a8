#!/bin/bash
# round 3, final trip: full GPU suite, rocprofv3 kernel stats of the default bench, PMC passes, default bench line
TAG=${1:-r03w}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/pytest_$TAG.txt
echo "== rocprof stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_$TAG.txt | head -14
echo "== pmc"; bash tools/gpu_pmc.sh $TAG > /dev/null 2>&1; ls gpurun_out/pmc_$TAG
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_$TAG.json; head -c 400 gpurun_out/bench_$TAG.json
