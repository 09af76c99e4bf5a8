"""Decode rate of cn_schedule="layered" (10 iterations) next to flooding (20 iterations) at config C2, min-sum."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy

k, n, m = 2816, 8448, 6
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
phy.config.seed = 1
enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
no = phy.utils.ebnodb2no(4.5, m, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
for name, kw in (("flooding-20", dict(num_iter=20)), ("layered-10", dict(num_iter=10, cn_schedule="layered")), ("layered-20", dict(num_iter=20, cn_schedule="layered"))):
    for cn in ("minsum", "boxplus-phi"):
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, **kw)
        out = dec(llr); torch.cuda.synchronize()
        t0 = time.perf_counter(); reps = 2
        for _ in range(reps): dec(llr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:12s} {cn:12s}: {dt*1e3:9.2f} ms / {B} = {B/dt/1e3:9.1f} k decodes/s   BLER {float((out != u).any(dim=1).float().mean()):.4f}", flush=True)
