"""Per-record timeline of the layered on-chip engine at config C2 (development aid; trace build:
make -C sionna_amd/csrc lytrace; SAMD_LIB=$PWD/sionna_amd/lib/libsionna_amd_lytrace.so python tools/ly_itrace.py).
For iteration 3 of workgroup 0: every step (the records between two barriers) with its duration, the slowest wave's items
and the time the waves spent waiting at the barrier; then totals per kind of step."""
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
    lib.samd_debug_set_ly_trace.argtypes = [C.c_void_p]
    k, n, m, B = 2816, 8448, 6, 2048
    cn = sys.argv[1] if len(sys.argv) > 1 else "minsum"
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, cn_schedule="layered", num_iter=10)
    phy.config.seed = 1
    no = phy.utils.ebnodb2no(4.5, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
    dec(llr)
    NW = 16
    trace = torch.zeros(NW * 512 * 2 + NW * 512 * 4, dtype=torch.int64, device="cuda")
    assert lib.samd_debug_set_ly_trace(C.c_void_p(trace.data_ptr())) == 0
    dec(llr)
    torch.cuda.synchronize()
    assert lib.samd_debug_set_ly_trace(None) == 0
    full = trace.cpu().numpy()
    t = full[:NW * 512 * 2].reshape(NW, 512, 2)
    sub = full[NW * 512 * 2:].reshape(NW, 512, 4)
    t0 = min(int(t[w, 0, 1]) for w in range(NW))
    # record word: bits 0-1 kind (2 CN, 3 VN), 2-7 key, 8-13 barriers after the body; a 0xF0 record closes a body and
    # carries four inner times: body start (after the dispatch), operands arrived, node update done, stores issued
    agg, steps = {}, {}
    end = 0
    for w in range(NW):
        recs = [(int(a), int(b) - t0, i) for i, (a, b) in enumerate(t[w]) if b != 0]
        end = max(end, recs[-1][1])
        for n, (tag, ts, _) in enumerate(recs[:-1]):
            if tag == 0xF0 or (tag & 3) < 2:
                continue
            tag2, te, i2 = recs[n + 1]
            assert tag2 == 0xF0
            tm = [int(x) - t0 for x in sub[w, i2]]
            nxt = recs[n + 2][1] if n + 2 < len(recs) else te
            key = ("CN" if (tag & 3) == 2 else "VN", (tag >> 2) & 15 if (tag & 3) == 3 else (tag >> 2) & 31, (tag >> 6) & 1 if (tag & 3) == 3 else (tag >> 7) & 1)
            mid = tm[2] if tm[2] > 0 else tm[1]
            agg.setdefault(key, []).append((tm[0] - ts, tm[1] - tm[0], mid - tm[1], tm[3] - mid, te - tm[3], nxt - te, (tag >> 8) & 63))
    print(f"{cn}: iteration 3 of workgroup 0 took {end} cycles")
    # steps: a record's barriers (bits 8-13) close steps; bodies per step and wave
    steps = {}
    for w in range(NW):
        recs = [(int(a), int(b) - t0, i) for i, (a, b) in enumerate(t[w]) if b != 0]
        st = 0
        for n, (tag, ts, _) in enumerate(recs[:-1]):
            if tag == 0xF0:
                continue
            if (tag & 3) >= 1:
                te = recs[n + 1][1]
                kind = {1: "P", 2: "C", 3: "V"}[tag & 3]
                key = (tag >> 2) & 31 if kind == "C" else (tag >> 2) & 7 if kind == "P" else 4 * ((tag >> 2) & 15)
                flag = ("f" if (tag >> 7) & 1 else "") if kind == "C" else ("f" if (tag >> 5) & 1 else "") if kind == "P" else ("p" if (tag >> 6) & 1 else "")
                steps.setdefault(st, {}).setdefault(w, []).append((f"{kind}{key}{flag}", ts, te))
            st += ((tag >> 8) & 63) + (1 if (tag & 3) == 1 else 0)            # (a split check-node part has a barrier inside)
    prev_end = 0
    for st in sorted(steps):
        ws = steps[st]
        first = min(b[1] for l in ws.values() for b in l)
        last = max(b[2] for l in ws.values() for b in l)
        desc = " ".join(f"w{w}:" + ",".join(f"{n}={te - ts}" for n, ts, te in l) for w, l in sorted(ws.items()))
        print(f"step {st:3d}: bodies from {first - prev_end:5d} to {last - prev_end:5d} after the previous step's last body | {desc}")
        prev_end = last
    print("body (kind, key, fused/pair): n x mean cycles [dispatch, operands, update, stores, tail, barriers + next record]")
    tot = np.zeros(6)
    for key in sorted(agg):
        a = np.array(agg[key], float)
        tot += a[:, :6].sum(axis=0)
        print(f"  {key}: {len(a):3d} x {np.round(a[:, :6].mean(axis=0)).astype(int).tolist()}")
    print("sum over all bodies of all waves:", np.round(tot).astype(int).tolist())


if __name__ == "__main__":
    main()
