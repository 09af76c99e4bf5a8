"""Layered (10 iterations) decode rate of one 5G code on the on-chip layered engine and on the scheduled HBM-resident
engine (SAMD_NO_ONCHIP_LAYERED=1).  python tools/layered_rate_code.py k n [batch]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi   # switches reach the library through samd_debug_set_option (it never reads the environment after load)

k, n = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
phy.config.seed = 1
enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=2)
no = phy.utils.ebnodb2no(2.0, 2, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", 2)(phy.channel.AWGN()(phy.mapping.Mapper("qam", 2)(enc(u)), no), no)
for cn in ("minsum", "boxplus-phi"):
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=10, cn_schedule="layered")
    for env in ("", "1"):
        if env:
            _ffi.set_option("SAMD_NO_ONCHIP_LAYERED", env)
        else:
            _ffi.set_option("SAMD_NO_ONCHIP_LAYERED", None)
        out = dec(llr); torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 2
        for _ in range(reps): dec(llr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"k={k} n={n} Z={enc._z if hasattr(enc, '_z') else '?'} {cn:12s} {'HBM-resident' if env else 'on-chip     '}: {dt*1e3:8.2f} ms / {B} = {B/dt/1e3:8.1f} k decodes/s  BLER {float((out != u).any(dim=1).float().mean()):.4f}", flush=True)
_ffi.set_option("SAMD_NO_ONCHIP_LAYERED", None)
