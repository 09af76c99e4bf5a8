#!/bin/bash
# round 5, trip 6: the cheaper defined phi (table-driven log, table read from global memory by every engine) - parity of all
# boxplus-phi engines against the oracle, decode rate at C2
TAG=${1:-r05f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_idd.py tests/test_gpu_edge_cases.py -x -q -k "phi or boxplus or generic or schedule or layered or reference_execution or random_codes or idd" > $OUT/pytest_phi.txt 2>&1; tail -6 $OUT/pytest_phi.txt
timeout 300 python tools/jit_ab.py --cn boxplus-phi --reps 3 --out $OUT/phi_rate.json defined_phi: > $OUT/phi_rate.txt 2>&1; cat $OUT/phi_rate.txt
timeout 300 python tools/jit_ab.py --cn boxplus-phi-fast --reps 3 fast_phi: 2>&1 | tail -1
