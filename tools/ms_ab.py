"""Development aid: A/B of kernel variants of the on-chip LDPC engine at config C2 in ONE process.

    python tools/ms_ab.py [--cn minsum] [--batch 32768] name:ENV=V,ENV=V ...

Every variant gets its environment (read by the C library when the code handle is created / at launch), a fresh handle,
its output is compared bit for bit with the first variant's, and its decode time is the median of 5 launches measured
with HIP events on the launch stream."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cn", default="minsum")
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--soft", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import sionna_amd.phy as phy
    k, n, m, B = 2816, 8448, 6, a.batch
    phy.config.seed = 1
    enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    no = phy.utils.ebnodb2no(4.5, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc0(u)), no), no)
    ref, rows = None, []
    for spec in a.variants:
        name, _, envs = spec.partition(":")
        env = dict(e.split("=") for e in envs.split(",") if e)
        saved = {kk: os.environ.get(kk) for kk in env}
        os.environ.update(env)
        try:
            enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=a.cn, num_iter=a.iters, hard_out=not a.soft)
            out = dec(llr).as_subclass(torch.Tensor)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref))
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); dec(llr); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = float(np.median(ts))
        finally:
            for kk, vv in saved.items():
                if vv is None:
                    os.environ.pop(kk, None)
                else:
                    os.environ[kk] = vv
        rows.append({"variant": name, "env": env, "ms": round(ms, 3), "M_decodes_per_s": round(B / ms / 1e3, 3), "same_bits_as_first": same})
        print(f"{name:28s} {ms:8.3f} ms / {B} -> {B / ms / 1e3:6.3f} M decodes/s   same bits: {same}", flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"cn": a.cn, "batch": B, "iters": a.iters, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
