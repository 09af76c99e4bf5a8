#!/bin/bash
# C5 bench line of the Polar register engine over (L2 stages G, workgroups per CU).  Outputs in gpurun_out/.
mkdir -p gpurun_out
for g in ${GS:-5 4}; do for pc in ${PCS:-32 24 16}; do
  echo "== G=$g per_cu=$pc $(SAMD_SCL_GSTAGES=$g SAMD_SCL_PER_CU=$pc timeout 300 python bench.py --workload c5 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done; done 2>&1 | tee gpurun_out/c5_sweep.txt
