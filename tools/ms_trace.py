"""Per-wave phase timeline of the explicit-message min-sum engine at config C2 (development aid).
Needs the trace build: make -C sionna_amd/csrc trace; run on the GPU box:
    SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_trace.so python tools/ms_trace.py > gpurun_out/ms_trace.txt
Prints, per iteration 2..5 of workgroup 0: for every wave the cycles spent in its CN items, waiting at the first
barrier, in its VN items and waiting at the second barrier (s_memtime, 100 MHz constant clock -> shown in ns)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    lib = _ffi.lib()
    lib.samd_debug_set_ms_trace.argtypes = [C.c_void_p]
    k, n, m, B = 2816, 8448, 6, 4096
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20)
    phy.config.seed = 1
    no = phy.utils.ebnodb2no(4.5, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
    dec(llr)
    NW = 16
    trace = torch.zeros(4 * 5 * NW, dtype=torch.int64, device="cuda")
    assert lib.samd_debug_set_ms_trace(C.c_void_p(trace.data_ptr())) == 0
    dec(llr)
    torch.cuda.synchronize()
    assert lib.samd_debug_set_ms_trace(None) == 0
    t = trace.cpu().numpy().reshape(4, 5, NW).astype(np.float64)
    tick_ns = 10.0          # s_memtime / readcyclecounter: 100 MHz on gfx9
    for it in range(4):
        a = t[it]
        cn, w1, vn, w2 = a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3]
        print(f"iteration {it + 2}: total {(a[4].max() - a[0].min()) * tick_ns:.0f} ns")
        for name, v in (("CN items", cn), ("barrier 1 wait", w1), ("VN items", vn), ("barrier 2 wait", w2)):
            print(f"  {name:15s} per wave [ns]: min {v.min() * tick_ns:7.0f}  mean {v.mean() * tick_ns:7.0f}  max {v.max() * tick_ns:7.0f}   "
                  + " ".join(f"{x * tick_ns:.0f}" for x in v))
    if len(t) > 1:
        print(f"iteration period: {(t[1:, 0].mean(axis=1) - t[:-1, 0].mean(axis=1)).mean() * tick_ns:.0f} ns")


if __name__ == "__main__":
    main()
