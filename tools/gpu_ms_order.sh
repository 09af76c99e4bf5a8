#!/bin/bash
# C2 min-sum decode time for the item-order modes of the LPT lists (SAMD_MS_ORDER, ldpc5g.h)
for m in ${MODES:-0 1 2 3 4}; do echo "== order $m: $(SAMD_MS_ORDER=$m timeout 300 python tools/ms_sweep.py 400,200 2>&1 | tail -1)"; done 2>&1 | tee gpurun_out/ms_order.txt
