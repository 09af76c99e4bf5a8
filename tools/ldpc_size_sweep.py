"""Decode rate of LDPC5GDecoder (min-sum, 20 iterations) over 5G code sizes: the kernel generated for the code
(csrc/ldpc5g_jit.cpp) against the generic on-chip engines on the same LLRs, decisions compared.

python tools/ldpc_size_sweep.py --out profiles/r06_ldpc_size_sweep.json [--cn minsum] [--quick]
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

CODES = [(100, 200, None), (512, 1024, None), (768, 1536, None), (1024, 2048, "bg1"), (1024, 2048, None), (1500, 3000, None),
         (2048, 6144, "bg1"), (2816, 5632, "bg1"), (2816, 8448, "bg1"), (3840, 7680, "bg1"), (4096, 6144, "bg1"),
         (5632, 8448, "bg1"), (5632, 11264, "bg1"), (6144, 9216, "bg1"), (8448, 12672, "bg1")]


def rate(dec, llr, reps):
    out = dec(llr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = dec(llr)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--cn", default="minsum")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--codes", default=None, help="k:n[:bg],...")
    a = ap.parse_args()
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    codes = CODES
    if a.codes:
        codes = []
        for s in a.codes.split(","):
            f = s.split(":")
            codes.append((int(f[0]), int(f[1]), f[2] if len(f) > 2 else None))
    rows = []
    for k, n, bg in codes:
        batch = max(4096, min(65536, (1 << 29) // (4 * n) // 4096 * 4096))
        if a.quick:
            batch = min(batch, 16384)
        enc = phy.fec.ldpc.LDPC5GEncoder(k, n, bg=bg)
        g = torch.Generator(device="cuda").manual_seed(k + n)
        u = (torch.rand((batch, k), device="cuda", generator=g) < 0.5).float()
        c = enc(u)
        sigma = 0.62 if k / n <= 0.4 else (0.72 if k / n <= 0.55 else 0.55)
        y = (2 * c - 1) + sigma * torch.randn(c.shape, device="cuda", generator=g)
        llr = (2 * y / sigma ** 2).float().contiguous()
        res = {}
        for tag, jit in (("generated", "1"), ("generic", "0")):
            with _ffi.option("SAMD_LDPC_JIT", jit):
                e2 = phy.fec.ldpc.LDPC5GEncoder(k, n, bg=bg)
                dec = phy.fec.ldpc.LDPC5GDecoder(e2, cn_update=a.cn, num_iter=a.iters, hard_out=True)
                dec(llr[:1024])                                            # build + compile outside the timing
                t, out = rate(dec, llr, 3 if a.quick else 5)
                h = e2._handle(dec._nb_pruned_nodes)
                res[tag] = {"ms": t * 1e3, "out": out, "launches": int(_ffi.lib().samd_ldpc5g_jit_launches(h))}
        same = bool(torch.equal(res["generated"]["out"], res["generic"]["out"]))
        ber = float((res["generated"]["out"] != u).float().mean())
        row = {"k": k, "n": n, "bg": enc._bg, "z": int(enc._z), "batch": batch, "generated_kernel_ran": res["generated"]["launches"] > 0,
               "ms": round(res["generated"]["ms"], 3), "decodes_per_s": int(batch / res["generated"]["ms"] * 1e3),
               "coded_gbit_per_s": round(batch * n / res["generated"]["ms"] / 1e6, 2),
               "generic_ms": round(res["generic"]["ms"], 3), "generic_decodes_per_s": int(batch / res["generic"]["ms"] * 1e3),
               "generic_coded_gbit_per_s": round(batch * n / res["generic"]["ms"] / 1e6, 2),
               "speedup": round(res["generic"]["ms"] / res["generated"]["ms"], 3), "same_decisions": same, "ber": ber}
        print(json.dumps(row), flush=True)
        rows.append(row)
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"sweep": f"LDPC5GDecoder {a.cn} {a.iters} iterations, hard decisions, generated kernel vs generic engines",
                       "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
