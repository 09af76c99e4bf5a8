"""Item-level timeline of the grouped explicit-message engine at config C2 (development aid; trace build:
make -C sionna_amd/csrc trace; SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_trace.so python tools/ms_itrace.py).
Prints, for iteration 3 of workgroup 0, every wave's phase durations and per-item times in shader-clock cycles
(s_memtime), and the fitted cycles per edge of the CN / VN bodies."""
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
    trace = torch.zeros(320 + NW * 24 * 2, dtype=torch.int64, device="cuda")
    assert lib.samd_debug_set_ms_trace(C.c_void_p(trace.data_ptr())) == 0
    dec(llr)
    torch.cuda.synchronize()
    assert lib.samd_debug_set_ms_trace(None) == 0
    t = trace.cpu().numpy()[320:].reshape(NW, 24, 2)
    t0 = min(int(t[w, 0, 1]) for w in range(NW) if t[w, 0, 0] == 1000)
    rows = []
    for w in range(NW):
        rec = [(int(a), int(b) - t0) for a, b in t[w] if a != 0]
        line, prev = [], None
        for tag, tm in rec:
            if tag in (1000, 2000, 3000):
                line.append(f"|{ {1000: 'CN', 2000: 'VN', 3000: 'end'}[tag]}@{tm}")
            else:
                ph, key = ("V", tag - 2000) if tag >= 2000 else ("C", tag)
                d, flag = key & 31, key >> 5
                line.append(f"{ph}{d}{'f' if (ph == 'C' and flag) else ('p' if (ph == 'V' and flag) else '')}:{tm - prev}")
                rows.append((ph, d, flag, tm - prev))
            prev = tm
        print(f"wave {w:2d} (SIMD {w % 4}, age {w // 4}): " + " ".join(line))
    for ph in "CV":
        for flag in (0, 1):
            xs = np.array([(d, c) for p_, d, f, c in rows if p_ == ph and f == flag], float)
            if len(xs) >= 3 and len(set(xs[:, 0])) >= 2:
                a, b = np.polyfit(xs[:, 0], xs[:, 1], 1)
                print(f"{'CN' if ph == 'C' else 'VN'} flag={flag}: cycles ~ {a:.0f} * degree + {b:.0f}   ({len(xs)} items)")


if __name__ == "__main__":
    main()
