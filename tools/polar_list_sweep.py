#!/usr/bin/env python3
"""Decode rate of the Polar list decoders over list size and code length (development / evidence aid): Polar5G uplink
codes at rate 1/2, QPSK AWGN LLRs at 2.5 dB, the engine each configuration gets (register engine: list sizes 4..32,
generic engine: list size 2 and SC), bit-exact check of a sample against the C oracle.

    python tools/polar_list_sweep.py --out profiles/r02c_polar_list_sweep.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "polar_list_sweep.json"))
    ap.add_argument("--batch", type=int, default=32768)
    args = ap.parse_args()
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    from oracle import polar as op, polar_c as pc
    phy.config.seed = 7
    rows = []
    for n, k in ((1024, 512), (512, 256), (256, 128), (128, 64)):
        enc = phy.fec.polar.Polar5GEncoder(k, n)
        ocode = op.Polar5GCode(k, n)
        no = phy.utils.ebnodb2no(2.5, 2, k / n)
        u = phy.mapping.BinarySource()([args.batch, k])
        x = phy.mapping.Mapper("qam", 2)(enc(u))
        llr = phy.mapping.Demapper("app", "qam", 2)(phy.channel.AWGN()(x, no), no)
        for dec_type, L in (("SC", 1), ("SCL", 2), ("SCL", 4), ("SCL", 8), ("SCL", 16), ("SCL", 32)):
            dec = phy.fec.polar.Polar5GDecoder(enc, dec_type, list_size=max(L, 1), return_crc_status=True)
            out, st = dec(llr)
            torch.cuda.synchronize()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                dec(llr)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            ns = 512
            ref, rst = pc.polar5g_decode(ocode, llr[:ns].cpu().numpy(), list_size=max(L, 1), precision="f32",
                                         return_crc_status=True, dec_type=dec_type)
            r = _ffi.lib().samd_polar_scl_register_stages(ocode.n_polar, max(L, 1), 1 if dec_type == "SC" else 0)
            row = {"n": n, "k": k, "decoder": dec_type, "list_size": L, "engine": "register" if r >= 1 else "generic",
                   "register_stages": r, "batch": args.batch, "ms": round(ms, 3),
                   "decodes_per_s": round(args.batch / ms * 1e3, 1),
                   "bler": float((out != u).any(-1).float().mean()),
                   "bit_exact_vs_oracle_sample": bool(np.array_equal(out[:ns].cpu().numpy(), ref) and
                                                      np.array_equal(st[:ns].cpu().numpy().astype(bool), rst.astype(bool)))}
            rows.append(row)
            print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"sweep": "Polar5G uplink rate 1/2, QPSK AWGN 2.5 dB", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
