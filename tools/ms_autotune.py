"""Search over LPT assignments of the flooding engine near the cost model's (development; SAMD_MS_PERTURB = seed of a
+-8 % perturbation of the item costs): decode-only time at C2 for seeds 0..N-1 in ONE process, best seeds re-measured."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
from sionna_amd import _ffi   # switches reach the library through samd_debug_set_option (it never reads the environment after load)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
k, n, m, B = 2816, 8448, 6, 32768
phy.config.seed = 1
enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
no = phy.utils.ebnodb2no(4.5, m, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc0(u)), no), no)


def measure(seed, reps=6):
    if seed is None:
        _ffi.set_option("SAMD_MS_PERTURB", None)
    else:
        _ffi.set_option("SAMD_MS_PERTURB", str(seed))
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20)
    dec(llr); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): dec(llr)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


base = [measure(None) for _ in range(3)]
print(f"model: {min(base):.4f} ms per {B} ({[round(x, 4) for x in base]})", flush=True)
res = []
for s in range(N):
    res.append((measure(s), s))
res.sort()
print("best:", [(round(t, 4), s) for t, s in res[:8]], flush=True)
print("worst:", [(round(t, 4), s) for t, s in res[-3:]], flush=True)
for t, s in res[:4]:
    print(f"seed {s}: re-measured {measure(s, 12):.4f} ms, model again {measure(None, 12):.4f} ms", flush=True)
