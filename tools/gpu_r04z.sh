#!/bin/bash
# round 4, verification trip on the final tree: full GPU suite, rocprofv3 kernel stats of the default bench, PMC passes
# (-> counters.json), the default bench line, and the overlay of every notebook BER table at 4x the notebook's statistics
TAG=${1:-r04z}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/pytest_$TAG.txt
echo "== rocprof stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_$TAG.txt | head -14
rm -rf gpurun_out/prof_$TAG
echo "== pmc"; bash tools/gpu_pmc.sh $TAG > /dev/null 2>&1; ls gpurun_out/pmc_$TAG; find gpurun_out/pmc_$TAG -name "*.db" -delete
cp gpurun_out/pmc_$TAG/counters.json profiles/counters.json      # so that the bench line below reads the fresh counters
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_$TAG.json; head -c 600 gpurun_out/bench_$TAG.json; echo
echo "== overlay"; timeout 1200 python tools/ber_vs_reference.py --mult 4 --out gpurun_out/ber_vs_reference_$TAG.json > gpurun_out/ber_vs_reference_$TAG.log 2>&1; tail -2 gpurun_out/ber_vs_reference_$TAG.log
du -sh gpurun_out
