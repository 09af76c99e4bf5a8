"""Flooding min-sum decode rate (20 iterations) of a few 5G code sizes with and without the item refinement
(SAMD_MS_VN_REFINE=0 SAMD_MS_CN_REFINE=0); handles are created per setting."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi   # switches reach the library through samd_debug_set_option (it never reads the environment after load)

phy.config.seed = 1
for k, n, bg in ((2816, 8448, "bg1"), (2816, 5632, "bg1"), (1408, 4224, "bg1"), (4096, 6144, "bg1"), (1920, 5760, "bg2"), (768, 1536, "bg2"), (5632, 8448, "bg1")):
    B = 16384
    res = []
    for env in ("0", None):
        for name in ("SAMD_MS_VN_REFINE", "SAMD_MS_CN_REFINE"):
            if env is None:
                _ffi.set_option(name, None)
            else:
                _ffi.set_option(name, env)
        enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=2, bg=bg)
        u = phy.mapping.BinarySource()([B, k])
        no = phy.utils.ebnodb2no(2.0, 2, k / n)
        llr = phy.mapping.Demapper("app", "qam", 2)(phy.channel.AWGN()(phy.mapping.Mapper("qam", 2)(enc(u)), no), no)
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20)
        out = dec(llr); torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 4
        for _ in range(reps): dec(llr)
        torch.cuda.synchronize()
        res.append(B * reps / (time.perf_counter() - t0) / 1e3)
    print(f"k={k} n={n} {bg}: without {res[0]:8.1f}  with {res[1]:8.1f} k decodes/s  ({(res[1]/res[0]-1)*100:+.1f} %)", flush=True)
