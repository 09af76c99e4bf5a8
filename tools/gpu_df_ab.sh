#!/bin/bash
# A/B of the dataflow-readiness variant of the headline kernel (SAMD_MS_DATAFLOW=1) against the two-barrier form.
# The dataflow form is compiled into the development library only (`make -C sionna_amd/csrc dataflow`, -DSAMD_MS_DF).
# Staged so that a deadlock cannot occupy the box: a tiny batch first (bounded by the kernel's spin limit), the parity
# tests, then the full bench lines.  Output: gpurun_out/<tag>_df_ab.txt
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_df_ab.txt
{
echo "== small batch, dataflow"; SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_df.so SAMD_MS_DATAFLOW=1 timeout 120 python bench.py --batch 2048 --steps 3 --warmup 1 --also none --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ber"], d["bler"])' || { echo "FAILED small batch"; exit 0; }
echo "== small batch, barriers"; timeout 120 python bench.py --batch 2048 --steps 3 --warmup 1 --also none --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ber"], d["bler"])'
echo "== parity tests under dataflow"; SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_df.so SAMD_MS_DATAFLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "minsum or bit_exact or random_codes or c2 or state or onchip" 2>&1 | tail -4
for i in 1 2; do
echo "== full batch, dataflow"; SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_df.so SAMD_MS_DATAFLOW=1 timeout 300 python bench.py --steps 10 --warmup 2 --also none --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ber"], d["bler"])'
echo "== full batch, barriers"; timeout 300 python bench.py --steps 10 --warmup 2 --also none --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ber"], d["bler"])'
done
} 2>&1 | tee $OUT
