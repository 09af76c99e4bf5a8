"""Grid over the cost-model / item-cut knobs of the flooding engine, every point measured in ONE process at C2
(decode-only, batch 32768); prints the best points and the default's rank (development)."""
import itertools, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi   # switches reach the library through samd_debug_set_option (it never reads the environment after load)

k, n, m, B = 2816, 8448, 6, 32768
phy.config.seed = 1
enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
no = phy.utils.ebnodb2no(4.5, m, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc0(u)), no), no)
KNOBS = {"SAMD_MS_CN_SLOPE": [32, 36, 40], "SAMD_MS_CN_OVH": [350, 400, 450], "SAMD_MS_VN_OVH": [180, 200, 220],
         "SAMD_MS_VN_REFINE_COST": [50, 60, 70, 80], "SAMD_MS_CN_REFINE_COST": [120, 150, 180]}
DEFAULT = (36, 400, 200, 70, 150)


def measure(cfg, reps=6):
    for name, v in zip(KNOBS, cfg):
        _ffi.set_option(name, str(v))
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20)
    dec(llr); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): dec(llr)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


res = sorted((measure(cfg), cfg) for cfg in itertools.product(*KNOBS.values()))
print("knobs:", list(KNOBS))
for t, cfg in res[:10]:
    print(f"{t:.4f} ms {cfg}")
rank = [cfg for _, cfg in res].index(DEFAULT)
print(f"default {DEFAULT}: rank {rank} of {len(res)}, {dict((c, t) for t, c in res)[DEFAULT]:.4f} ms; worst {res[-1][0]:.4f} ms")
for t, cfg in res[:3]:
    print(f"re-measured {cfg}: {measure(cfg, 12):.4f} ms; default {measure(DEFAULT, 12):.4f} ms")
