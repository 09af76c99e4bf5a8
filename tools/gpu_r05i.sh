#!/bin/bash
# round 5, trip 9: interleaved message layout of the specialised kernel (8-byte DS instructions in both phases)
TAG=${1:-r05i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_jit.py -x -q > $OUT/pytest_jit.txt 2>&1; tail -3 $OUT/pytest_jit.txt
timeout 900 python tools/jit_ab.py --out $OUT/jit_ab.json generic:SAMD_LDPC_JIT=0 planar_default: interleaved:SAMD_JIT_LAYOUT=1 \
  interleaved_pipe2:SAMD_JIT_LAYOUT=1,SAMD_JIT_PIPE=2 interleaved_noprefetch:SAMD_JIT_LAYOUT=1,SAMD_JIT_PREFETCH=0 \
  interleaved_sched1:SAMD_JIT_LAYOUT=1,SAMD_JIT_SCHED=1 interleaved_sched1_ovh60:SAMD_JIT_LAYOUT=1,SAMD_JIT_SCHED=1,SAMD_JIT_CN_OVH=60 \
  interleaved_novnrev:SAMD_JIT_LAYOUT=1,SAMD_JIT_VNREV=0 > $OUT/jit_ab.txt 2>&1; cat $OUT/jit_ab.txt
