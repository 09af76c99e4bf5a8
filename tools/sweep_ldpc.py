#!/usr/bin/env python3
"""Decode throughput of LDPC5GDecoder (min-sum, 20 iterations) over a set of 5G code sizes: which
engine runs, lifting size, lane utilisation of the on-chip engine (Z / (64 ceil(Z/64))) and the rate.
    python tools/sweep_ldpc.py > gpurun_out/ldpc_sweep.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi

_ffi.device()
phy.config.seed = 1
CN = sys.argv[1] if len(sys.argv) > 1 else "minsum"          # python tools/sweep_ldpc.py boxplus-phi
CN_ID = {"boxplus": 0, "boxplus-phi": 1, "minsum": 2, "offset-minsum": 3}[CN]
rows = []
for k, n, bg in [(512, 1024, None), (768, 1536, None), (1024, 2048, "bg1"), (1500, 3000, None), (2048, 6144, "bg1"),
                 (2816, 8448, "bg1"), (3840, 7680, None), (4096, 6144, None), (4224, 12672, "bg1"), (5632, 11264, None),
                 (6144, 9216, None), (7040, 14080, None), (8448, 12672, None), (8448, 16896, None), (8448, 25344, None), (3840, 19200, None)]:
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, bg=bg)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=CN, num_iter=20)
    B = int(min(65536, max(4096, 2 ** int(np.log2(5e8 / n)))))
    if CN_ID < 2:
        B = max(4096, B // 4)                                 # boxplus rules: 3-10x more work per decode
    u = phy.mapping.BinarySource()([B, k])
    c = enc(u).as_subclass(torch.Tensor)
    llr = (4.0 * (2 * c - 1) + 2.0 * torch.randn_like(c)).contiguous()
    out = dec(llr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        dec(llr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    z = enc.z
    ws = _ffi.lib().samd_ldpc5g_decode_workspace_bytes(enc._handle(dec._nb_pruned_nodes), B, CN_ID) if dec._onchip_ok else 0
    eng = int(_ffi.lib().samd_ldpc5g_decode_engine(enc._handle(dec._nb_pruned_nodes), CN_ID))
    rows.append({"k": k, "n": n, "bg": enc._bg, "z": z,
                 "engine": {0: "generic-hbm", 1: "on-chip compressed state" + (" (part of it in L2)" if ws else ""),
                            2: "on-chip explicit messages" + (" (channel LLRs in L2)" if ws else ""),
                            3: "on-chip explicit messages, last rows in L2"}[eng if dec._onchip_ok else 0],
                 "lane_utilisation": round(z / (64 * -(-z // 64)), 3), "batch": B, "ms": round(dt * 1e3, 2),
                 "decodes_per_s": round(B / dt), "coded_gbit_per_s": round(B * n / dt / 1e9, 2),
                 "ber": float((out != u).float().mean())})
    print(rows[-1], file=sys.stderr)
print(json.dumps({"sweep": f"LDPC5GDecoder {CN} 20 iterations", "rows": rows}))
