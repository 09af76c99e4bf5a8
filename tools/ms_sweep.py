"""Development aid: decode time of the explicit min-sum engine at C2 for a grid of LPT per-item overhead constants
(SAMD_MS_CN_OVH / SAMD_MS_VN_OVH are read when the code handle is created)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import sionna_amd.phy as phy
    from sionna_amd import _ffi   # switches reach the library through samd_debug_set_option (no environment reads after load)
    k, n, m, B = 2816, 8448, 6, 32768
    phy.config.seed = 1
    enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    no = phy.utils.ebnodb2no(4.5, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc0(u)), no), no)
    ref = None
    grid = [(c, v) for c in (40, 100, 160, 240, 400) for v in (40, 100, 200)]
    caps = [None]
    if len(sys.argv) > 1:
        grid = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:] if not a.startswith(("cap=", "env="))]
        caps = [a[4:] for a in sys.argv[1:] if a.startswith("cap=")] or [None]
        grid = [g for g in grid if len(g) == 2]
    for a in sys.argv[1:]:
        if a.startswith("env="):
            kk, vv = a[4:].split(":")
            _ffi.set_option(kk, vv)
    for cap in caps:
      if cap:
        _ffi.set_option("SAMD_MS_CAP", cap)
        print("cap", cap)
      for cn, vn in grid:
        _ffi.set_option("SAMD_MS_CN_OVH", cn); _ffi.set_option("SAMD_MS_VN_OVH", vn)
        enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20)
        out = dec(llr)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        assert torch.equal(out.as_subclass(torch.Tensor), ref.as_subclass(torch.Tensor))
        t0 = time.perf_counter()
        for _ in range(4):
            dec(llr)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        print(f"cn_ovh {cn:4d} vn_ovh {vn:4d}: {ms:7.3f} ms / {B} -> {B / ms / 1e3:6.3f} M decodes/s", flush=True)


if __name__ == "__main__":
    main()
