#!/bin/bash
# round 5, trip 3: cost model of the specialised kernel's own schedule against the generic kernel's lists
TAG=${1:-r05c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V="generic:SAMD_LDPC_JIT=0 oldsched:SAMD_JIT_SCHED=0 oldsched_nopre:SAMD_JIT_SCHED=0,SAMD_JIT_PREFETCH=0 oldsched_xor:SAMD_JIT_SCHED=0,SAMD_JIT_XOR128=1"
for co in 60 110 160 240; do for vo in 40 100 200; do V="$V c${co}_v${vo}:SAMD_JIT_CN_OVH=$co,SAMD_JIT_VN_OVH=$vo"; done; done
V="$V c110_v100_p1:SAMD_JIT_CN_OVH=110,SAMD_JIT_VN_OVH=100,SAMD_JIT_PIPE=1 c110_v100_norot:SAMD_JIT_CN_OVH=110,SAMD_JIT_VN_OVH=100,SAMD_JIT_ROTATE=0 c110_v100_novnrev:SAMD_JIT_CN_OVH=110,SAMD_JIT_VN_OVH=100,SAMD_JIT_VNREV=0"
timeout 1500 python tools/jit_ab.py --out $OUT/jit_ab.json $V > $OUT/jit_ab.txt 2>&1
cat $OUT/jit_ab.txt
