#!/bin/bash
# round 5, trip 7: the cheaper defined phi with its table in LDS on the explicit-message engine
TAG=${1:-r05g}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/jit_ab.py --cn boxplus-phi --reps 3 --out $OUT/phi_rate.json table_in_lds: > $OUT/phi_rate.txt 2>&1; cat $OUT/phi_rate.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_idd.py -x -q -k "phi or boxplus or random_codes or reference_execution" > $OUT/pytest_phi.txt 2>&1; tail -4 $OUT/pytest_phi.txt
