#!/bin/bash
# round 5, trip 2: specialised LDPC kernel after the DS-offset fix and its own schedule - parity, knob sweep, ablation
# (development library), kernel trace + PMC
TAG=${1:-r05b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_jit.py -x -q > $OUT/pytest_jit.txt 2>&1; tail -3 $OUT/pytest_jit.txt
timeout 900 python tools/jit_ab.py --out $OUT/jit_ab.json \
  generic:SAMD_LDPC_JIT=0 jit_default: \
  pipe1:SAMD_JIT_PIPE=1 pipe1_noprefetch:SAMD_JIT_PIPE=1,SAMD_JIT_PREFETCH=0 xor128:SAMD_JIT_XOR128=1 xor128_pipe1:SAMD_JIT_XOR128=1,SAMD_JIT_PIPE=1 \
  noprefetch:SAMD_JIT_PREFETCH=0 oldsched:SAMD_JIT_SCHED=0 oldsched_pipe1:SAMD_JIT_SCHED=0,SAMD_JIT_PIPE=1 \
  novnrev:SAMD_JIT_VNREV=0 noprio:SAMD_JIT_PRIO=0 norotate:SAMD_JIT_ROTATE=0 \
  vnpair30:SAMD_JIT_VN_PAIR_MAX=30,SAMD_JIT_XOR128=1,SAMD_JIT_PIPE=1 vnpair16:SAMD_JIT_VN_PAIR_MAX=16,SAMD_JIT_PIPE=1 \
  ovh40:SAMD_JIT_CN_OVH=40 ovh20:SAMD_JIT_CN_OVH=20 vnslope12:SAMD_JIT_VN_SLOPE=12 vnslope5:SAMD_JIT_VN_SLOPE=5 > $OUT/jit_ab.txt 2>&1
cat $OUT/jit_ab.txt
SAMD_LIB=$GRAFT_REPO_ROOT/sionna_amd/lib/libsionna_amd_jitdev.so timeout 600 python tools/jit_ab.py --out $OUT/jit_abl.json \
  full: no_cn_arith:SAMD_JIT_ABL=1 no_vn_arith:SAMD_JIT_ABL=2 no_arith:SAMD_JIT_ABL=3 no_barriers:SAMD_JIT_ABL=4 \
  no_arith_no_barriers:SAMD_JIT_ABL=7 > $OUT/jit_abl.txt 2>&1
cat $OUT/jit_abl.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
  python $GRAFT_REPO_ROOT/tools/jit_ab.py --reps 3 jit_default: > $OUT/trace.log 2>&1
run() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- \
    python $GRAFT_REPO_ROOT/tools/jit_ab.py --reps 1 jit_default: > $OUT/pmc_$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
run sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
grep -h "samd_ldpc5g_jit" $OUT/trace/*kernel_stats.csv 2>/dev/null | head -3
find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
grep -A28 "^samd_ldpc5g_jit" $OUT/pmc_summary.txt
