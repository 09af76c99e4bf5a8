#!/bin/bash
TAG=${1:-r03r}
mkdir -p gpurun_out
echo "== parity (min-sum / phi at scale, random codes)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "minsum or at_scale or random_codes or boxplus_phi_bit" 2>&1 | tail -3
echo "== A/B fused start"; timeout 300 python tools/ms_ab.py --cn minsum --batch 65536 nofused:SAMD_MS_NOFUSEDINIT=1 fused: nofused2:SAMD_MS_NOFUSEDINIT=1 fused2: 2>&1 | tail -4 | tee gpurun_out/ms_fusedinit_$TAG.txt
echo "== iterations 0 / 20"; for it in 0 20; do echo "iters $it: $(timeout 200 python tools/ms_ab.py --cn minsum --iters $it nofused:SAMD_MS_NOFUSEDINIT=1 fused: 2>&1 | tail -2 | tr '\n' ' ')"; done | tee -a gpurun_out/ms_fusedinit_$TAG.txt
echo "== prefetch build"; SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_pre.so timeout 300 python tools/ms_ab.py --cn minsum --batch 65536 pre: pre2: 2>&1 | tail -2 | tee -a gpurun_out/ms_fusedinit_$TAG.txt
