#!/usr/bin/env python3
"""Stress check of the on-chip LAYERED decoder (csrc/ldpc5g_onchip_ly.hip; evidence / development aid, uses oracle/):
random 5G code sizes (first argument: how many; "any" as second argument: every lifting size, default: multiples of
64), both base graphs, rates 1/5 ... 0.9; cn_schedule="layered"
soft outputs for min-sum and boxplus-phi must equal the oracle's literal form (check-node update of the layer, then every
variable node) bit for bit - on the on-chip engine where it takes the code (reported per row), else on the scheduled
HBM-resident engine.
    python tools/ldpc_layered_parity.py 40 > gpurun_out/ldpc_layered_parity.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi
from oracle.ldpc5g import LDPC5GCode
from oracle import ldpc_bp as obp

_ffi.device()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ANY = len(sys.argv) > 2 and sys.argv[2] == "any"
ALL_Z = sorted({a * 2 ** j for a in (2, 3, 5, 7, 9, 11, 13, 15) for j in range(8) if a * 2 ** j <= 384})
rng = np.random.default_rng(2027)
rows, bad, tried = [], 0, 0
while len(rows) < N and tried < 40 * N:
    tried += 1
    z = int(rng.choice(ALL_Z)) if ANY else int(rng.choice([64, 128, 192, 256, 320, 384]))
    bg = str(rng.choice(["bg1", "bg2"]))
    kb = 22 if bg == "bg1" else 10
    k = kb * z - int(rng.integers(0, 16))                      # (a few filler bits)
    rate = float(rng.uniform(0.2 if bg == "bg2" else 0.34, 0.92))
    n = int(k / rate)
    try:
        code = LDPC5GCode(k, n, None, bg)
    except ValueError:
        continue
    if code.z != z:
        continue
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, bg=bg)
    B = 5
    u = rng.integers(0, 2, (B, k)).astype(np.float32)
    c = enc(u).cpu().numpy()
    sigma = 0.8
    llr = (2 * ((2 * c - 1) + sigma * rng.normal(size=c.shape)) / sigma ** 2).astype(np.float32)
    row = {"k": k, "n": n, "z": code.z, "bg": code.bg}
    for cn in ("minsum", "boxplus-phi"):
        kw = dict(cn_update=cn, cn_schedule="layered", num_iter=2, hard_out=False, return_infobits=False)
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, **kw)
        row["on_chip"] = bool(_ffi.lib().samd_ldpc5g_decode_layered_supported(enc._handle(dec._nb_pruned_nodes), dec._cn_mode))
        got = dec(llr).cpu().numpy()
        ref = obp.LDPC5GDecoder(code, **kw).decode5g(llr)
        ok = bool(np.array_equal(got, ref))
        row[cn + "_bit_exact"] = ok
        bad += 0 if ok else 1
    rows.append(row)
    print(row, file=sys.stderr, flush=True)
print(json.dumps({"codes": len(rows), "on_chip": sum(r["on_chip"] for r in rows), "failures": bad, "rows": rows}, indent=1))
