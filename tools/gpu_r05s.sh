#!/bin/bash
# round 5, final trip: the whole GPU suite, generated-vs-generic A/B, traced bench, PMC passes -> counters.json, the bench line that reads them
# PMC passes of the default bench -> counters.json, then the bench line that reads them
TAG=${1:-r05s}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
timeout 300 python tools/jit_ab.py --out $OUT/jit_ab.json generic:SAMD_LDPC_JIT=0 jit_default: planar_layout:SAMD_JIT_LAYOUT=0 > $OUT/jit_ab.txt 2>&1; cat $OUT/jit_ab.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py > $OUT/bench_traced.json 2> $OUT/trace.log
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh $TAG > $OUT/pmc.log 2>&1
find gpurun_out/pmc_$TAG -name "*.db" -delete; find gpurun_out/pmc_$TAG -name "*_agent_info.csv" -delete; find $OUT -name "*.db" -delete
cp gpurun_out/pmc_$TAG/counters.json profiles/counters.json
timeout 900 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json; head -c 1500 $OUT/bench.json; echo
head -20 $OUT/trace/t_kernel_stats.csv
