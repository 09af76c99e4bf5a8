#!/bin/bash
# round-2 trip A: new parity bars (LMMSE f32 oracle, C4 chain, SCL-8 bit exact), phi at scale, C5/C2 speed
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== ofdm+polar tests"; timeout 1500 python -m pytest tests/test_gpu_ofdm.py tests/test_gpu_polar.py -q -m gpu -x 2>&1 | tail -25
echo "== phi scale"; timeout 600 python tools/phi_scale_check.py 2048 > gpurun_out/phi_scale_r02a.json 2> gpurun_out/phi_scale_r02a.err; tail -3 gpurun_out/phi_scale_r02a.err; cat gpurun_out/phi_scale_r02a.json
echo "== bench c5"; timeout 600 python bench.py --workload c5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_c5_r02a.json
echo "== bench c2"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_c2_r02a.json
