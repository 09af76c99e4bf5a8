#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_idd.py tests/test_gpu_double.py -q -m gpu 2>&1 | tail -5
for cn in minsum boxplus-phi; do
  for np in 0 1; do
    if [ $np = 1 ]; then export SAMD_MS_NOPRIO=1; else unset SAMD_MS_NOPRIO; fi
    timeout 900 python tools/sweep_ldpc.py $cn > gpurun_out/sweep_${cn}_noprio${np}.json 2> gpurun_out/sweep_${cn}_noprio${np}.log
  done
done
unset SAMD_MS_NOPRIO
python - <<'PY'
import json
for cn in ("minsum","boxplus-phi"):
    a=json.load(open(f"gpurun_out/sweep_{cn}_noprio0.json"))["rows"]; b=json.load(open(f"gpurun_out/sweep_{cn}_noprio1.json"))["rows"]
    print(cn)
    for x,y in zip(a,b):
        print(f"  k={x['k']:5d} n={x['n']:5d} z={x['z']:3d} {x['engine'][:34]:34s} prio {x['decodes_per_s']:9d}/s  noprio {y['decodes_per_s']:9d}/s  x{x['decodes_per_s']/y['decodes_per_s']:.3f}")
PY
echo "== bench c2"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_c2_r02e.json | cut -c1-300
