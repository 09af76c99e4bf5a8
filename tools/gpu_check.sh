#!/bin/bash
# One GPU-box trip: smoke, GPU parity tests, bench (+ A/B of the two on-chip engines), rocprof.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu (v2 on-chip engine)"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== pytest gpu (v1 on-chip engine)"; SAMD_ONCHIP_V1=1 timeout 900 python -m pytest tests -m gpu -x -q -k "5g or c2" 2>&1 | tail -5
echo "== bench v2"; timeout 600 python bench.py --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_v2.json
echo "== bench v1"; SAMD_ONCHIP_V1=1 timeout 600 python bench.py --steps 5 --warmup 1 --also '' --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_v1.json
echo "== rocprof"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -20 $f; done
