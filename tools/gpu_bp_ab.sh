#!/bin/bash
# boxplus rules at C2: the explicit-message engine with the boxplus node update vs the first boxplus kernel
# (SAMD_BP_ENGINE=1), same process each; parity tests of the boxplus rules first.
echo "== pytest boxplus"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_idd.py tests/test_gpu_double.py -q -x -k "boxplus or phi or tanh or idd or state or scale" 2>&1 | tail -3
for e in 0 1; do
  echo "== SAMD_BP_ENGINE=$e"
  ( [ $e = 1 ] && export SAMD_BP_ENGINE=1; python - <<PY
import torch, time
import sionna_amd.phy as phy
from sionna_amd import _ffi
_ffi.device()
k, n, m, B = 2816, 8448, 6, 32768
phy.config.seed = 3
enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
no = phy.utils.ebnodb2no(4.5, m, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
for cn in ("boxplus-phi", "boxplus"):
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=20, hard_out=False)
    out = dec(llr); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): dec(llr)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    print(cn, round(ms, 2), "ms /", B, "->", round(B / ms / 1e3, 1), "k decodes/s", "checksum", float(out.double().abs().sum()))
PY
  )
done
