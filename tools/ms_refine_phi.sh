#!/bin/bash
run() { echo "[$1] $(env $1 timeout 300 python bench.py --cn-update boxplus-phi --steps 4 --warmup 1 --no-extra --also none --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ber"])')"; }
run "A=0"
for v in "$@"; do run "$v"; done
run "A=0"
