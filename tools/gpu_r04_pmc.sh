#!/bin/bash
# round 4: PMC passes of the default bench on the final tree -> counters.json, then the bench line that reads them
TAG=${1:-r04zz}
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_pmc.sh $TAG > /dev/null 2>&1; ls gpurun_out/pmc_$TAG; find gpurun_out/pmc_$TAG -name "*.db" -delete; find gpurun_out/pmc_$TAG -name "*.csv" -delete
cp gpurun_out/pmc_$TAG/counters.json profiles/counters.json
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_${TAG}b.json; head -c 300 gpurun_out/bench_${TAG}b.json; echo
