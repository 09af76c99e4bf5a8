#!/usr/bin/env python3
"""Stress check of the on-chip decoders on random 5G code sizes (evidence / development aid, uses oracle/): min-sum
and (since round 3: phi on the defined float32 exp / log) boxplus-phi soft outputs must equal the C oracle bit for bit.
Draws small and medium codes on purpose - lifting sizes below and between multiples of 64 exercise the packed-tail
items of the explicit-message engine.
    python tools/ldpc_random_parity.py 80 > gpurun_out/ldpc_random_parity.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi
from oracle.ldpc5g import LDPC5GCode
from oracle import ldpc_bp as obp, cbind

_ffi.device()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2026)
rows, bad = [], 0
for i in range(N):
    k = int(rng.choice([rng.integers(12, 200), rng.integers(200, 1200), rng.integers(1200, 4000)]))
    rate = float(rng.uniform(0.22, 0.9))
    n = int(min(max(k / rate, k + 8), 20000))
    try:
        code = LDPC5GCode(k, n)
    except ValueError:
        continue
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    B = 48
    u = rng.integers(0, 2, (B, k)).astype(np.float32)
    c = enc(u).cpu().numpy()
    sigma = 0.9
    llr = (2 * ((2 * c - 1) + sigma * rng.normal(size=c.shape)) / sigma ** 2).astype(np.float32)
    row = {"k": k, "n": n, "z": code.z, "bg": code.bg}
    for cn, it in (("minsum", 6), ("boxplus-phi", 4)):
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=it, hard_out=False)
        got = dec(llr).cpu().numpy()
        odec = obp.LDPC5GDecoder(code, cn_update=cn, num_iter=it, hard_out=False)
        ref = cbind.bp_decode(odec, odec.rate_recover(llr))[:, :k]
        # round 3: boxplus-phi evaluates phi on the defined exp / log -> soft outputs bit for bit as well
        ok = bool(np.array_equal(got, ref.astype(np.float32)))
        row["minsum_bit_exact" if cn == "minsum" else "phi_bit_exact"] = ok
        bad += 0 if ok else 1
    rows.append(row)
print(json.dumps({"codes": len(rows), "failures": bad, "rows": rows}, indent=1))
