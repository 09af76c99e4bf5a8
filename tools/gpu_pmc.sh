#!/bin/bash
# PMC passes for the dominant kernels of the default bench command (separate runs, --pmc with --kernel-trace only),
# then the per-unit counters file that bench.py reads.
# usage: bash tools/gpu_pmc.sh <tag>     -> gpurun_out/pmc_<tag>/{summary.txt,counters.json}
TAG=${1:-r02}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
run() {  # name, counters...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "${EXTRA[@]}" > $OUT/$name.log 2>&1
}
EXTRA=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
run sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
python tools/pmc_counters.py $OUT --tag $TAG ldpc5g_jit=65536 ldpc5g_ms=65536 ldpc5g_bp=65536 ldpc5g_bp_fast=65536 ldpc5g_layered=65536 polar_scl=32768 polar_bp=32768 ofdm_lmmse=6291456 ofdm_lsnn_lmmse=6291456 cir_to_ofdm=69730304 tdl_cir=21102592 > $OUT/counters.json
head -c 1500 $OUT/counters.json
