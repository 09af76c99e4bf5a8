#!/bin/bash
# PMC passes (separate runs, --pmc with --kernel-trace only) over tools/pmc_mini.py: the C4 front end and the Polar BP
# decoder - the two kernels whose counters are missing / flagged stale after the second half of round 4.
# usage: bash tools/gpu_pmc_mini.sh <tag>     -> gpurun_out/pmc_<tag>/{summary.txt,counters.json}
TAG=${1:-r04mini}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- \
    python $GRAFT_REPO_ROOT/tools/pmc_mini.py > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
python tools/pmc_counters.py $OUT --tag $TAG polar_bp=32768 ofdm_lmmse=6291456 > $OUT/counters.json
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -size +2M -delete
head -c 1200 $OUT/counters.json; tail -1 $OUT/sq1.log | head -c 300
