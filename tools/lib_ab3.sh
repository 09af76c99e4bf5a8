#!/bin/bash
# like lib_ab.sh for any number of builds; checks that the decisions agree (ber)
for r in 1 2 3; do
  for L in "$@"; do
    echo "$(basename $L) $(SAMD_LIB=$PWD/$L timeout 300 python bench.py --steps 10 --warmup 2 --no-extra --also none --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ber"])')"
  done
done
