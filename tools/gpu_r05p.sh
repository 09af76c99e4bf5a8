#!/bin/bash
# round 5, trip 16: PMC counters of the pass-based cir_to_ofdm kernel alone (tools/c2o_ab.py, default variant)
TAG=${1:-r05p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/c2o_ab.py default: > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
run sq3 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_ACCUM_PREV
cd $GRAFT_REPO_ROOT

python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -i "cir_to_ofdm" -A14 $OUT/summary.txt | head -80
