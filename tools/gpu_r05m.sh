#!/bin/bash
# round 5, trip 13: precision="double" blocks (csrc/f64_ofdm.hip) + the generated-kernel tests after the phi default change
TAG=${1:-r05m}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_double.py -q -m gpu > $OUT/pytest_double_jit.txt 2>&1; tail -30 $OUT/pytest_double_jit.txt
