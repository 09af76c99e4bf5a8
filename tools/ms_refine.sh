#!/bin/bash
run() { echo "[$1] $(env $1 timeout 300 python bench.py --steps 10 --warmup 2 --no-extra --also none --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ber"])')"; }
run "A=0"
for v in "SAMD_MS_VN_REFINE=2" "SAMD_MS_VN_REFINE=4" "SAMD_MS_VN_REFINE=8" "SAMD_MS_VN_REFINE=16" "SAMD_MS_VN_REFINE=8 SAMD_MS_VN_REFINE_COST=20" "SAMD_MS_VN_REFINE=8 SAMD_MS_VN_REFINE_COST=100"; do run "$v"; done
run "A=0"
