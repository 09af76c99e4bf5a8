#!/bin/bash
TAG=${1:-r03q}
mkdir -p gpurun_out
echo "== random parity"; timeout 900 python tools/ldpc_random_parity.py 80 > gpurun_out/ldpc_random_parity_$TAG.json 2>gpurun_out/rp.err; python -c "import json; d=json.load(open('gpurun_out/ldpc_random_parity_$TAG.json')); print(d['codes'], 'codes', d['failures'], 'failures')"
echo "== BER curve"; timeout 1500 python tools/ber_curve.py --out gpurun_out/ber_c2_$TAG.json 2>/dev/null | tail -1 | cut -c1-200
python -c "
import json; d=json.load(open('gpurun_out/ber_c2_$TAG.json'))
print(d['ebno_db_at_bler']); print(d['gap_db_to_boxplus_phi'])
for k,v in d['rules'].items(): print(k, [x for x in v.values() if isinstance(x, list) and len(x) and not isinstance(x[0], float) or True][3] if False else (v.get('oracle_bit_exact_on_sample') or v.get('oracle_hard_decisions_equal_on_sample')))
"
