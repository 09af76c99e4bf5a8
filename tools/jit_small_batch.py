"""Generated LDPC kernel against the generic engines at SMALL batches (where the 1024-codeword threshold of round 5 came from):
python tools/jit_small_batch.py --out profiles/r06_jit_small_batch.json"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    rows = []
    for k, n, bg in ((768, 1536, None), (1024, 2048, "bg1"), (100, 200, None), (2816, 8448, "bg1"), (4096, 6144, "bg1")):
        enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, bg=bg)
        g = torch.Generator(device="cuda").manual_seed(k)
        llr_all = (4.0 * torch.randn((4096, n), device="cuda", generator=g) + 4.0).contiguous()
        for batch in (16, 64, 128, 256, 512, 1024, 2048, 4096):
            llr = llr_all[:batch].contiguous()
            res = {}
            for tag, jit in (("generic", "0"), ("generated", "2")):
                with _ffi.option("SAMD_LDPC_JIT", jit):
                    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, bg=bg)
                    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20, hard_out=True)
                    out = dec(llr)
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(7):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); out = dec(llr); e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1))
                    res[tag] = (float(np.median(ts)), out.as_subclass(torch.Tensor).clone())
            row = {"k": k, "n": n, "z": int(enc0._z), "batch": batch, "generic_ms": round(res["generic"][0], 4),
                   "generated_ms": round(res["generated"][0], 4), "speedup": round(res["generic"][0] / res["generated"][0], 3),
                   "same": bool(torch.equal(res["generic"][1], res["generated"][1]))}
            print(json.dumps(row), flush=True)
            rows.append(row)
    if a.out:
        json.dump({"what": "LDPC5GDecoder min-sum BP-20, one call, median of 7 (HIP events around the Block call)", "rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
