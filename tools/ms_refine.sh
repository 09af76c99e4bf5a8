#!/bin/bash
# decode-only C2 bench (min-sum) under the item-refinement knobs of the explicit-message engine (one box)
run() { echo "[$1] $(env $1 timeout 300 python bench.py --steps 10 --warmup 2 --no-extra --also none --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ber"])')"; }
run "A=0"
for v in "$@"; do run "$v"; done
run "A=0"
