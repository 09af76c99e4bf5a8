"""A/B of the specialised 5G LDPC decoder (csrc/ldpc5g_jit.cpp) against the generic kernel at config C2, in ONE process.

    python tools/jit_ab.py [--batch 65536] [--cn minsum] name:SAMD_X=V,SAMD_Y=V ...

Every variant gets its development options (samd_debug_set_option; handles capture them at creation), a fresh handle,
its SOFT outputs are compared bit for bit with the first variant's, and its decode time is the median of `--reps`
launches measured with HIP events on the launch stream."""
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
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ebno", type=float, default=4.5)
    ap.add_argument("--out", default=None)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    k, n, m, B = 2816, 8448, 6, a.batch
    phy.config.seed = 1
    enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    no = phy.utils.ebnodb2no(a.ebno, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc0(u)), no), no)
    ref, rows = None, []
    for spec in a.variants:
        name, _, envs = spec.partition(":")
        env = dict(e.split("=") for e in envs.split(",") if e)
        for kk, vv in env.items():
            _ffi.set_option(kk, vv)
        try:
            enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=a.cn, num_iter=a.iters, hard_out=False)
            out = dec(llr).as_subclass(torch.Tensor)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref))
            jit = int(_ffi.lib().samd_ldpc5g_jit_launches(enc._handle(dec._nb_pruned_nodes)))
            ts = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); dec(llr); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = float(np.median(ts))
        finally:
            for kk in env:
                _ffi.set_option(kk, None)
        rows.append({"variant": name, "options": env, "ms": round(ms, 3), "M_decodes_per_s": round(B / ms / 1e3, 3),
                     "same_bits_as_first": same, "specialised_kernel_ran": jit > 0})
        print(f"{name:34s} {ms:8.3f} ms / {B} -> {B / ms / 1e3:6.3f} M decodes/s   same bits: {same}   jit: {jit > 0}", flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"cn": a.cn, "batch": B, "iters": a.iters, "ebno_db": a.ebno, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
