"""Grid over the cost model of the layered engine's scheduler (development knobs SAMD_LY_*), every point measured in ONE
process at C2 (layered, 10 iterations, min-sum, batch 16384); prints the best points and the default's rank."""
import itertools, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi   # switches reach the library through samd_debug_set_option (it never reads the environment after load)

k, n, m, B = 2816, 8448, 6, 16384
phy.config.seed = 1
enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
no = phy.utils.ebnodb2no(4.5, m, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc0(u)), no), no)
KNOBS = {"SAMD_LY_CN_SLOPE": [30, 60, 100], "SAMD_LY_CN_OVH": [150, 300, 600], "SAMD_LY_VN_SLOPE": [9, 18, 30],
         "SAMD_LY_VN_OVH": [120, 250, 500], "SAMD_LY_SIMD_ALPHA": [30, 45, 60]}
DEFAULT = (60, 300, 18, 250, 45)


def measure(cfg, reps=3):
    for name, v in zip(KNOBS, cfg):
        _ffi.set_option(name, str(v))
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", cn_schedule="layered", num_iter=10)
    dec(llr); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): dec(llr)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


res = sorted((measure(cfg), cfg) for cfg in itertools.product(*KNOBS.values()))
print("knobs:", list(KNOBS))
for t, cfg in res[:10]:
    print(f"{t:.3f} ms {cfg}")
print(f"default {DEFAULT}: rank {[c for _, c in res].index(DEFAULT)} of {len(res)}, {dict((c, t) for t, c in res)[DEFAULT]:.3f} ms; worst {res[-1][0]:.3f} ms")
for t, cfg in res[:3]:
    print(f"re-measured {cfg}: {measure(cfg, 6):.3f} ms; default {measure(DEFAULT, 6):.3f} ms")
