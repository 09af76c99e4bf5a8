#!/bin/bash
# round 5, trip 17: OFDM / channel tests after the cir_to_ofdm and TDL changes, C4 bench line
TAG=${1:-r05q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ofdm.py tests/test_gpu_cdl.py tests/test_gpu_double.py -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout 600 python bench.py --workload c4 2>$OUT/bench_c4.err | tail -1 > $OUT/bench_c4.json; python -c "
import json;d=json.load(open('$OUT/bench_c4.json'));print(d['value'], d['ms_per_step']);print(json.dumps(d['channel_kernels'])[:1500])"
