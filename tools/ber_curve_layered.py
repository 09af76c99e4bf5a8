#!/usr/bin/env python3
"""BLER versus Eb/N0 of config C2 with cn_schedule="layered" (10 and 20 iterations) next to flooding (20 iterations), on the
GPU engines, min-sum and boxplus-phi; at every point the oracle's literal layered form (oracle/ldpc_bp.py: check-node update
of the layer, then every variable node) decodes a sample of the same device LLRs and must give the same soft outputs bit
for bit.  Test / evidence tooling (uses oracle/).
    python tools/ber_curve_layered.py > gpurun_out/ber_layered_c2.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from oracle.ldpc5g import LDPC5GCode
from oracle import ldpc_bp as obp

k, n, m, B, S = 2816, 8448, 6, 16384, 48
phy.config.seed = 20260924
enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
code = LDPC5GCode(k, n, m, "bg1")
src, mapper = phy.mapping.BinarySource(), phy.mapping.Mapper("qam", m)
demap, chan = phy.mapping.Demapper("app", "qam", m), phy.channel.AWGN()
ebnos = [3.5, 4.0, 4.5, 5.0]
res = {"config": "C2: LDPC5G BG1 k=2816 n=8448 (num_bits_per_symbol=6), 64-QAM, AWGN", "batch": B, "oracle_sample": S,
       "ebno_db": ebnos, "rules": {}}
t0 = time.time()
for cn in ("minsum", "boxplus-phi"):
    rows = {"flooding-20": [], "layered-10": [], "layered-20": [], "layered_soft_outputs_equal_oracle": []}
    for e in ebnos:
        no = phy.utils.ebnodb2no(e, m, k / n)
        u = src([B, k])
        llr = demap(chan(mapper(enc(u)), no), no)
        for name, kw in (("flooding-20", dict(num_iter=20)), ("layered-10", dict(num_iter=10, cn_schedule="layered")),
                         ("layered-20", dict(num_iter=20, cn_schedule="layered"))):
            out = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, **kw)(llr)
            rows[name].append(float((out != u).any(dim=1).float().mean()))
        kw = dict(cn_update=cn, cn_schedule="layered", num_iter=10, hard_out=False)
        got = phy.fec.ldpc.LDPC5GDecoder(enc, **kw)(llr[:S]).cpu().numpy()
        ref = obp.LDPC5GDecoder(code, **kw).decode5g(llr[:S].cpu().numpy())
        rows["layered_soft_outputs_equal_oracle"].append(bool(np.array_equal(got, ref)))
        print(cn, e, {kk: v[-1] for kk, v in rows.items()}, f"{time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    res["rules"][cn] = rows
print(json.dumps(res, indent=1))
