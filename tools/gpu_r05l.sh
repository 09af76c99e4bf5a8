#!/bin/bash
# round 5, trip 12: boxplus-phi on the generated kernel against the generic explicit-message kernel
TAG=${1:-r05l}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python tools/jit_ab.py --cn boxplus-phi --reps 3 --out $OUT/phi_ab.json generic_kernel:SAMD_LDPC_JIT_PHI=0 generated_kernel: generated_planar:SAMD_JIT_LAYOUT=0 > $OUT/phi_ab.txt 2>&1; cat $OUT/phi_ab.txt
