"""Development aid: LDPC5GEncoder time at C2 (batch 65536) for the debug phases of the packed kernel (SAMD_ENC_DBG)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sionna_amd.phy as phy
k, n, m, B = 2816, 8448, 6, 65536
enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
u = phy.mapping.BinarySource()([B, k])
for name, env in [("packed", {}), ("no input", {"SAMD_ENC_DBG": "1"}), ("no rows", {"SAMD_ENC_DBG": "2"}), ("no output", {"SAMD_ENC_DBG": "4"}),
                  ("only output", {"SAMD_ENC_DBG": "3"}), ("nothing", {"SAMD_ENC_DBG": "7"}), ("bytes kernel", {"SAMD_ENC_BYTES": "1"})]:
    for kk in ("SAMD_ENC_DBG", "SAMD_ENC_BYTES"): os.environ.pop(kk, None)
    os.environ.update(env)
    enc(u); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): enc(u)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:14s} {e0.elapsed_time(e1) / 5:8.3f} ms", flush=True)
