#!/bin/bash
# round 5, trip 22: check-node comparisons issued ahead of the selections (volatile assembly order) in the generated kernel
TAG=${1:-r05v}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/jit_ab.py --reps 5 --out $OUT/jit_ab.json jit_default: cmp_ahead1:SAMD_JIT_CMP_AHEAD=1 cmp_ahead2:SAMD_JIT_CMP_AHEAD=2 cmp_ahead3:SAMD_JIT_CMP_AHEAD=3 jit_default_again: > $OUT/jit_ab.txt 2>&1; cat $OUT/jit_ab.txt
timeout 600 python tools/jit_ab.py --cn offset-minsum --reps 3 --out $OUT/jit_ab_off.json jit_default: cmp_ahead2:SAMD_JIT_CMP_AHEAD=2 > $OUT/jit_ab_off.txt 2>&1; cat $OUT/jit_ab_off.txt
timeout 600 python -m pytest tests/test_gpu_jit.py -q -x > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
