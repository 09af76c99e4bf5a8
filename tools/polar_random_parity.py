#!/usr/bin/env python3
"""Stress check of the Polar list decoders on random configurations (evidence / development aid, uses oracle/): hard
decisions and CRC status must equal the C oracle (float32 specification arithmetic) bit for bit.
    python tools/polar_random_parity.py 120 > gpurun_out/polar_random_parity.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi
from oracle import polar as op, polar_c as pc

_ffi.device()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(77)
rows, bad = [], 0
for i in range(N):
    n = int(rng.choice([32, 64, 128, 256, 512, 1024]))
    L = int(rng.choice([1, 2, 4, 8, 16, 32]))
    crc = [None, "CRC6", "CRC11", "CRC16", "CRC24C"][int(rng.integers(0, 5))]
    kc = op.CRC_POLYS[crc][0] if crc else 0
    k = int(rng.integers(kc + 1, max(kc + 2, n - 1)))
    fast = bool(rng.integers(0, 2))
    frozen, info = phy.fec.polar.generate_5g_ranking(k, n)
    B = 64
    u = rng.integers(0, 2, (B, k - kc)).astype(np.float32)
    uc = op.crc_encode(u, crc) if crc else u
    c = op.polar_encode(uc, info, n)
    sigma = float(rng.uniform(0.5, 1.2))
    logits = (2 * ((2 * c - 1) + sigma * rng.normal(size=c.shape)) / sigma ** 2).astype(np.float32)
    dec = phy.fec.polar.PolarSCLDecoder(frozen, n, list_size=L, crc_degree=crc, use_fast_scl=fast, return_crc_status=crc is not None)
    out = dec(logits)
    got, status = (out if crc else (out, None))
    ref, ref_status = pc.SCLDecoder(frozen, n, L, crc, fast).decode(logits)
    ok = bool(np.array_equal(got.cpu().numpy(), ref)) and (crc is None or bool(np.array_equal(status.cpu().numpy().astype(bool), ref_status)))
    r = _ffi.lib().samd_polar_scl_register_stages(n, L, 0)
    rows.append({"n": n, "k": k, "L": L, "crc": crc, "fast": fast, "sigma": round(sigma, 3), "engine": "register" if r >= 1 else "generic",
                 "bit_exact": ok})
    bad += 0 if ok else 1
# SC decoder
for n in (64, 128, 512, 1024):
    for k in (n // 4, n // 2, (3 * n) // 4):
        frozen, info = phy.fec.polar.generate_5g_ranking(k, n)
        u = rng.integers(0, 2, (64, k)).astype(np.float32)
        c = op.polar_encode(u, info, n)
        logits = (2 * ((2 * c - 1) + 0.8 * rng.normal(size=c.shape)) / 0.64).astype(np.float32)
        got = phy.fec.polar.PolarSCDecoder(frozen, n)(logits).cpu().numpy()
        ref = pc.sc_decode(logits, frozen, n)
        ok = bool(np.array_equal(got, ref))
        rows.append({"n": n, "k": k, "L": "SC", "bit_exact": ok})
        bad += 0 if ok else 1
print(json.dumps({"configs": len(rows), "failures": bad, "rows": rows}, indent=1))
