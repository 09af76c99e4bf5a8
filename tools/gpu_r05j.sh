#!/bin/bash
# round 5, trip 10: schedule cost model of the interleaved-layout kernel
TAG=${1:-r05j}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="SAMD_JIT_LAYOUT=1,SAMD_JIT_SCHED=1"
V="planar_default: il_oldlists:SAMD_JIT_LAYOUT=1"
for co in 40 60 80 110; do for vo in 10 40 80; do V="$V c${co}_v${vo}:$B,SAMD_JIT_CN_OVH=$co,SAMD_JIT_VN_OVH=$vo"; done; done
V="$V c60_v10_nopre:$B,SAMD_JIT_CN_OVH=60,SAMD_JIT_PREFETCH=0 c60_v10_pipe2:$B,SAMD_JIT_CN_OVH=60,SAMD_JIT_PIPE=2 c60_v10_pair16:$B,SAMD_JIT_CN_OVH=60,SAMD_JIT_VN_PAIR_MAX=16 c60_v10_pair30:$B,SAMD_JIT_CN_OVH=60,SAMD_JIT_VN_PAIR_MAX=30 c60_v10_norot:$B,SAMD_JIT_CN_OVH=60,SAMD_JIT_ROTATE=0 c60_v10_novnrev:$B,SAMD_JIT_CN_OVH=60,SAMD_JIT_VNREV=0 c60_v10_vs12:$B,SAMD_JIT_CN_OVH=60,SAMD_JIT_VN_SLOPE=12 c60_v10_vs5:$B,SAMD_JIT_CN_OVH=60,SAMD_JIT_VN_SLOPE=5"
timeout 1500 python tools/jit_ab.py --reps 4 --out $OUT/jit_ab.json $V > $OUT/jit_ab.txt 2>&1
cat $OUT/jit_ab.txt
