#!/bin/bash
# round 5, trip 18: several (batch, receiver) pairs per thread in the OFDM LMMSE kernels
TAG=${1:-r05r}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/lmmse_ab.py > $OUT/lmmse_ab.txt 2>&1; cat $OUT/lmmse_ab.txt
timeout 900 python -m pytest tests/test_gpu_ofdm.py tests/test_gpu_double.py -q -x > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
