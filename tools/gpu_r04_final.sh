#!/bin/bash
# round 4, last trip: smoke, the full GPU suite, the default bench line, kernel stats of the same command
TAG=${1:-r04zz}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/pytest_$TAG.txt
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_$TAG.json; head -c 300 gpurun_out/bench_$TAG.json; echo
echo "== rocprof stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_$TAG.txt | head -8
rm -rf gpurun_out/prof_$TAG
