#!/bin/bash
TAG=${1:-r03m}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layered or encoder or demapper or mapper or chain" 2>&1 | tail -3
echo "== layered rate"; timeout 600 python tools/layered_rate.py 16384 2>&1 | grep -v amdgpu.ids | grep -v "boxplus" | tee gpurun_out/layered_rate_$TAG.txt
echo "== stats"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --also none > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(find gpurun_out/prof_$TAG -name "*results.db" | head -1) | tee gpurun_out/kernel_stats_$TAG.txt | head -11
