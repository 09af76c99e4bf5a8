"""One development knob of the layered engine's host scheduler (csrc/ldpc5g_onchip_ly.hip) swept in ONE process at C2
(layered, 10 iterations, batch 65536): `python tools/ly_knob.py SAMD_LY_SIMD_ALPHA 0 40 60 80 100 [KEY=VALUE ...] [--cn boxplus-phi]`.
Options reach the library through samd_debug_set_option; a decoder handle captures them when it is created."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi

k, n, m, B = 2816, 8448, 6, 65536
bg = "bg1"
cn = sys.argv[sys.argv.index("--cn") + 1] if "--cn" in sys.argv else "minsum"
args = [a for a in sys.argv[1:] if a not in ("--cn", cn)]
if "--kn" in args:                                              # another code: --kn k n (base graph chosen by the encoder)
    i = args.index("--kn")
    k, n, bg, B = int(args[i + 1]), int(args[i + 2]), None, 32768
    del args[i:i + 3]
for a in [a for a in args if "=" in a]:                         # fixed options beside the swept one: KEY=VALUE
    _ffi.set_option(*a.split("="))
args = [a for a in args if "=" not in a]
knob, values = args[0], [int(v) for v in args[1:]]
phy.config.seed = 1
enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
no = phy.utils.ebnodb2no(4.5, m, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc0(u)), no), no)
ref = None
for rep in range(2):
    for v in values:
        _ffi.set_option(knob, v)
        enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, cn_schedule="layered", num_iter=10, hard_out=False)
        out = dec(llr); torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out.as_subclass(torch.Tensor), ref.as_subclass(torch.Tensor)))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(3): dec(llr)
        ev[1].record(); torch.cuda.synchronize()
        t = ev[0].elapsed_time(ev[1]) / 3
        print(f"{knob}={v:4d}: {t:8.2f} ms / {B} = {B / t:7.1f} k decodes/s   soft outputs identical to the first: {same}", flush=True)
