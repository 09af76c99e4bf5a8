#!/bin/bash
# round 5, trip 26: waves per workgroup of the generated kernel (own schedule, SAMD_JIT_WAVES) x load pipelining depth
TAG=${1:-r05v2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python tools/jit_ab.py --reps 5 --out $OUT/jit_waves.json jit_default: w8:SAMD_JIT_WAVES=8 w8_pipe2:SAMD_JIT_WAVES=8,SAMD_JIT_PIPE=2 w10:SAMD_JIT_WAVES=10 w12:SAMD_JIT_WAVES=12 w12_pipe2:SAMD_JIT_WAVES=12,SAMD_JIT_PIPE=2 w13:SAMD_JIT_WAVES=13 w14:SAMD_JIT_WAVES=14 w15:SAMD_JIT_WAVES=15 jit_default_again: > $OUT/jit_waves.txt 2>&1; cat $OUT/jit_waves.txt
