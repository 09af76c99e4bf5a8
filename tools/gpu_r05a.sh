#!/bin/bash
# round 5, trip 1: the specialised LDPC kernel on hardware - parity, A/B against the generic kernel over the generator's
# knobs, kernel trace and PMC passes of the specialised kernel
TAG=${1:-r05a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_jit.py -x -q > $OUT/pytest_jit.txt 2>&1; tail -5 $OUT/pytest_jit.txt
timeout 600 python tools/jit_ab.py --out $OUT/jit_ab.json \
  generic:SAMD_LDPC_JIT=0 jit_default: \
  pipe1:SAMD_JIT_PIPE=1 xor128:SAMD_JIT_XOR128=1 xor128_pipe1:SAMD_JIT_XOR128=1,SAMD_JIT_PIPE=1 \
  noprefetch:SAMD_JIT_PREFETCH=0 xor128_noprefetch:SAMD_JIT_XOR128=1,SAMD_JIT_PREFETCH=0 \
  novnrev:SAMD_JIT_VNREV=0 noprio:SAMD_JIT_PRIO=0 pipe3:SAMD_JIT_PIPE=3,SAMD_JIT_XOR128=1 > $OUT/jit_ab.txt 2>&1
cat $OUT/jit_ab.txt
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
run sq4 SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
grep -h "samd_ldpc5g_jit\|ldpc5g_decode_msg" $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head
find $OUT -name "*.db" -delete; find $OUT -name "*_agent_info.csv" -delete
tail -40 $OUT/pmc_summary.txt
