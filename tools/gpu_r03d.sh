#!/bin/bash
TAG=${1:-r03d}
mkdir -p gpurun_out
echo "== item trace"; SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_trace.so timeout 300 python tools/ms_itrace.py 2>&1 | tail -24 | tee gpurun_out/ms_itrace_$TAG.txt
echo "== iterations sweep"; for it in 0 1 5 10 20 40; do echo "iters $it: $(timeout 200 python tools/ms_ab.py --cn minsum --iters $it x: 2>&1 | tail -1)"; done | tee gpurun_out/ms_iters_$TAG.txt
echo "== variants again"; timeout 300 python tools/ms_ab.py --cn minsum old:SAMD_MS_NOGROUP=1 g: gz_bitop3:SAMD_MS_VAR=1 old2:SAMD_MS_NOGROUP=1 g2: gz_bitop3_2:SAMD_MS_VAR=1 2>&1 | tail -6 | tee gpurun_out/ms_ab2_$TAG.txt
