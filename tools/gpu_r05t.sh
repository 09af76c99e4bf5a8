#!/bin/bash
# round 5, trip 20: maximum-likelihood detectors (mimo + OFDM) against the reference-executed fixture and the float64 oracle
TAG=${1:-r05t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_double.py tests/test_gpu_cdl.py -q > $OUT/pytest_ml.txt 2>&1; tail -40 $OUT/pytest_ml.txt
