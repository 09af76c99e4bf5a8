#!/bin/bash
TAG=${1:-r03e}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQC?_[A-Z0-9_]+)" | sort -u | grep -i -E "ICACHE|IFETCH|INST_CACHE|DCACHE|SQC_|SQ_INST_LEVEL|SQ_WAIT|SQ_ACTIVE|SQ_INSTS|SQ_INST_CYC|SQ_BUSY|SQ_VALU" | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/counters_avail_$TAG.txt
cat $GRAFT_REPO_ROOT/gpurun_out/counters_avail_$TAG.txt | head -c 3000; echo
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/ms_ab.py --cn minsum --batch 16384 g: > $OUT/$name.log 2>&1; tail -2 $OUT/$name.log; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run ic2 SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
run ic3 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_LDS
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT 2>&1 | tee $OUT/summary.txt | head -60
