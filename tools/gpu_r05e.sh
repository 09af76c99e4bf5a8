#!/bin/bash
# round 5, trip 5: the C4 kernels of the round (register-staged cir_to_ofdm, two resource elements per lane in the LMMSE
# kernels) - parity, A/B, C4 bench line; plus the failed launcher test of trip 4 and the extended reference-execution twins
TAG=${1:-r05e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ofdm.py tests/test_gpu_rccl.py tests/test_gpu_jit.py "tests/test_gpu_parity.py::test_5g_chain_matches_reference_execution" tests/test_gpu_idd.py tests/test_gpu_cdl.py -x -q > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
timeout 300 python tools/c4_ab.py $OUT/c4_ab.json > $OUT/c4_ab.txt 2>&1; cat $OUT/c4_ab.txt
timeout 600 python bench.py --workload c4 2>$OUT/bench_c4.err | tail -1 > $OUT/bench_c4.json; head -c 600 $OUT/bench_c4.json; echo
timeout 300 python tools/jit_ab.py --out $OUT/jit_ab.json generic:SAMD_LDPC_JIT=0 jit_default: pipe2:SAMD_JIT_PIPE=2 > $OUT/jit_ab.txt 2>&1; cat $OUT/jit_ab.txt
# eight processes sharing the one GPU, collectives over gloo: bootstrap, LOCAL_RANK binding and teardown with 8 ranks
SAMD_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --batch 4096 --steps 2 --warmup 1 --also none --no-cpu-baseline --no-extra > $OUT/bench_8ranks_gloo.json 2> $OUT/bench_8ranks_gloo.err; tail -c 700 $OUT/bench_8ranks_gloo.json; echo; tail -3 $OUT/bench_8ranks_gloo.err
