#!/bin/bash
# First trip of the NEXT round (~12 GPU minutes): what the second half of round 4 could not afford.
#   1. the whole GPU suite over 8 workers (3 min; -v log survives a cut-off)
#   2. rocprofv3 --kernel-trace --stats of the default bench command -> kernel_stats_<tag>.txt
#   3. the full PMC passes (tools/gpu_pmc.sh: all seven kernels incl. polar_bp) -> counters.json, copied to profiles/ by hand
#   4. the default bench line reading the fresh counters
# usage: gpurun --timeout 900 -- 'bash tools/gpu_next_first_trip.sh r05a'
TAG=${1:-r05a}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (xdist)"
timeout 420 python -m pytest tests -m gpu -v -n 8 --dist load --tb=short -r fEx -p no:cacheprovider > gpurun_out/pytest_$TAG.txt 2>&1
grep -E "^FAILED|^ERROR|crashed" gpurun_out/pytest_$TAG.txt | head -20; tail -2 gpurun_out/pytest_$TAG.txt
echo "== rocprof stats"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_$TAG.txt | head -14
rm -rf gpurun_out/prof_$TAG
echo "== pmc"; bash tools/gpu_pmc.sh $TAG > /dev/null 2>&1; ls gpurun_out/pmc_$TAG; find gpurun_out/pmc_$TAG -name "*.db" -delete
cp gpurun_out/pmc_$TAG/counters.json profiles/counters.json      # so that the bench line below reads the fresh counters
echo "== bench"; timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_$TAG.json; head -c 500 gpurun_out/bench_$TAG.json; echo
