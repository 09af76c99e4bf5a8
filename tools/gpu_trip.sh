#!/bin/bash
# One parametrised GPU trip (replaces the per-trip tools/gpu_r0*.sh scripts of rounds 3-5).
#   bash tools/gpurun.sh --timeout S -- 'bash tools/gpu_trip.sh TAG step [step ...]'
# steps: jit_tests | parity | all_tests | sweep | sweep_quick | bench | bench_prof | pmc | c4 | sh:<command>
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
    bench_prof) (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 > $OUT/bench_prof.json 2> $OUT/bench_prof.err); python tools/prof_summary.py $OUT/prof > $OUT/kernel_stats.txt 2>&1; head -40 $OUT/kernel_stats.txt ;;
    sh:*) bash -c "${step#sh:}" 2>&1 | tee -a $OUT/sh.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
