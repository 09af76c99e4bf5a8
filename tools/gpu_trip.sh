#!/bin/bash
# One parametrised GPU trip (replaces the per-trip tools/gpu_r0*.sh scripts of rounds 3-5).
#   bash tools/gpurun.sh --timeout S -- 'bash tools/gpu_trip.sh TAG step [step ...]'
# steps: jit_tests | parity | all_tests | sweep | sweep_quick | bench | bench_prof | pmc | idd_rate | pic_rate | ubench2 | sh:<command>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for step in "$@"; do
  echo "=== $step" | tee -a $OUT/steps.txt
  case "$step" in
    jit_tests) timeout 1500 python -m pytest tests/test_gpu_jit.py -x -q 2>&1 | tail -15 | tee $OUT/jit_tests.txt ;;
    parity) timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15 | tee $OUT/parity.txt ;;
    all_tests) timeout 3400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/all_tests.txt ;;
    sweep) timeout 1500 python tools/ldpc_size_sweep.py --out $OUT/ldpc_size_sweep.json 2>&1 | tee $OUT/ldpc_size_sweep.txt ;;
    sweep_quick) timeout 900 python tools/ldpc_size_sweep.py --quick --out $OUT/ldpc_size_sweep_quick.json 2>&1 | tee $OUT/ldpc_size_sweep_quick.txt ;;
    bench) timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 6000 $OUT/bench.json ;;
    bench_prof) (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 > $OUT/bench_prof.json 2> $OUT/bench_prof.err); python tools/prof_summary.py $OUT/prof/bench_results.db > $OUT/kernel_stats.txt 2>&1; head -40 $OUT/kernel_stats.txt ;;
    pmc)  # PMC passes for the kernels of the default bench command (separate runs, --pmc with --kernel-trace only), then the
          # per-unit counters file bench.py reads (copy $OUT/pmc/counters.json to profiles/counters.json, summary to profiles/)
      mkdir -p $OUT/pmc
      runpmc() { local name=$1; shift; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc/$name -o p -- \
                 python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc/$name.log 2>&1); }
      runpmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
      runpmc sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
      runpmc sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
      runpmc tcc1 FETCH_SIZE
      runpmc tcc2 WRITE_SIZE
      python tools/pmc_summary.py $OUT/pmc > $OUT/pmc/summary.txt 2>&1
      python tools/pmc_counters.py $OUT/pmc --tag $TAG ldpc5g_jit=65536 ldpc5g_jit_phi=65536 ldpc5g_jit_c4=16384 ldpc5g_ms=65536 ldpc5g_bp=65536 ldpc5g_bp_fast=65536 \
        ldpc5g_layered=65536 polar_scl=32768 polar_bp=32768 ofdm_lmmse=6291456 ofdm_lsnn_lmmse=6291456 cir_to_ofdm=69730304 tdl_cir=21102592 > $OUT/pmc/counters.json
      head -c 600 $OUT/pmc/counters.json ;;
    idd_rate) timeout 600 python tools/idd_rate.py 2>&1 | tee $OUT/idd_rate.txt ;;          # return_state / msg_v2c chain
    pic_rate) timeout 600 python tools/pic_rate.py 2>&1 | tee $OUT/pic_rate.txt ;;          # MMSE-PIC against the LMMSE detector
    ubench2) hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rate2 tools/ubench/valu_rate2.hip && timeout 300 /tmp/valu_rate2 2>&1 | tee $OUT/valu_rate2.txt ;;   # issue classes
    sh:*) bash -c "${step#sh:}" 2>&1 | tee -a $OUT/sh.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
