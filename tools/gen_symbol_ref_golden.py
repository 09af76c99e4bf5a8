#!/usr/bin/env python3
"""Generates tests/golden/symbol_ref_golden.npz by EXECUTING the reference's own code under the NumPy stand-in for
TensorFlow (tools/ref_exec) for everything that works on SYMBOL logits / indices instead of bit LLRs:

    mapping.py           LLRs2SymbolLogits :969-1058, SymbolLogits2Moments :1061-1138, SymbolInds2Bits :1141-1178,
                         QAM2PAM :1181-1231, PAM2QAM :1234-1314
    mimo/detection.py    EPDetector(output="symbol") soft and hard :1272-1295, KBestDetector(output="symbol",
                         hard_out=True) :1001-1019, MMSEPICDetector(output="symbol") with logits as priors :1523-1524,
                         1636-1637, LinearDetector(output="symbol")
    ofdm/detection.py    the same three detectors behind OFDMDetector / OFDMDetectorWithPrior (symbol output layout
                         [batch, num_tx, num_streams, num_data_symbols(, num_points)] :289-317, 531-560) on the "c4" link of
                         tests/golden/ofdm_rx_ref_golden.npz (inputs are read from that fixture)

Run here (needs /root/reference); the fixture travels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "symbol_ref_golden.npz")
MIMO = [(4, 2, 2), (4, 2, 4), (8, 4, 4), (4, 4, 6), (2, 2, 6)]      # (rx antennas, streams, bits per symbol)


def cn(rng, shape, var=1.0):
    return ((rng.normal(size=shape) + 1j * rng.normal(size=shape)) * np.sqrt(var / 2)).astype(np.complex64)


def main():
    from tools.gen_ofdm_rx_ref_golden import load, LINKS
    mp, mimo, ofdm, od, ce, eq = load()
    rng = np.random.default_rng(20260925)
    out = {"mimo_cases": np.array(MIMO, np.int32)}

    # ---- mapping.py utilities
    for m in (1, 2, 4, 6):
        llrs = (rng.normal(size=(3, 7, m)) * 4).astype(np.float32)
        llrs[0, 0] = 0.
        llrs[0, 1] = 40.
        out[f"l2s{m}_llrs"] = llrs
        out[f"l2s{m}_logits"] = np.asarray(mp.LLRs2SymbolLogits(m)(llrs))
        out[f"l2s{m}_hard"] = np.asarray(mp.LLRs2SymbolLogits(m, hard_out=True)(llrs)).astype(np.int32)
        ind = rng.integers(0, 1 << m, (4, 5)).astype(np.int32)
        out[f"i2b{m}_ind"], out[f"i2b{m}_bits"] = ind, np.asarray(mp.SymbolInds2Bits(m)(ind))
    for m in (2, 4, 6):
        logits = (rng.normal(size=(3, 6, 1 << m)) * 3).astype(np.float32)
        mean, var = mp.SymbolLogits2Moments("qam", m)(logits)
        out[f"mom{m}_logits"], out[f"mom{m}_mean"], out[f"mom{m}_var"] = logits, np.asarray(mean), np.asarray(var)
        q = rng.integers(0, 1 << m, (5, 3)).astype(np.int32)
        p1, p2 = mp.QAM2PAM(m)(q)
        out[f"q2p{m}_q"], out[f"q2p{m}_p1"], out[f"q2p{m}_p2"] = q, np.asarray(p1).astype(np.int32), np.asarray(p2).astype(np.int32)
        out[f"p2q{m}_q"] = np.asarray(mp.PAM2QAM(m)(np.asarray(p1), np.asarray(p2))).astype(np.int32)
        a = (rng.normal(size=(4, 3, 1 << (m // 2))) * 2).astype(np.float32)
        b = (rng.normal(size=(4, 3, 1 << (m // 2))) * 2).astype(np.float32)
        out[f"p2q{m}_a"], out[f"p2q{m}_b"] = a, b
        out[f"p2q{m}_logits"] = np.asarray(mp.PAM2QAM(m, hard_in_out=False)(a, b))

    # ---- mimo detectors with symbol output
    for ci, (M, K, m) in enumerate(MIMO):
        n = 24
        h = cn(rng, (n, M, K))
        bits = rng.integers(0, 2, (n, K, m)).astype(np.float32)
        x = np.asarray(mp.Mapper("qam", m)(bits.reshape(n, K * m)))
        a = cn(rng, (n, M, M))
        s = (0.1 * np.eye(M) + 0.05 * a @ np.conj(np.swapaxes(a, -1, -2))).astype(np.complex64)
        w = (np.linalg.cholesky(s.astype(np.complex128)) @ cn(rng, (n, M, 1)).astype(np.complex128))[..., 0]
        y = ((h.astype(np.complex128) @ x[..., None].astype(np.complex128))[..., 0] + w).astype(np.complex64)
        o = dict(y=y, h=h, s=s)
        o["ep_logits"] = np.asarray(mimo.EPDetector("symbol", m, hard_out=False, l=6)(y, h, s))
        o["ep_hard"] = np.asarray(mimo.EPDetector("symbol", m, hard_out=True, l=6)(y, h, s)).astype(np.int32)
        kk = min(16, (1 << m) ** K)
        o["kbest_hard"] = np.asarray(mimo.KBestDetector("symbol", K, kk, "qam", m, hard_out=True)(y, h, s)).astype(np.int32)
        o["kbest_k"] = np.int32(kk)
        prior = (rng.normal(size=(n, K, 1 << m)) * 2).astype(np.float32)
        o["pic_prior"] = prior
        for meth in ("app", "maxlog"):
            o[f"pic_logits_{meth}"] = np.asarray(mimo.MMSEPICDetector("symbol", meth, 2, "qam", m)(y, h, s, prior))
        o["pic_hard"] = np.asarray(mimo.MMSEPICDetector("symbol", "maxlog", 1, "qam", m, hard_out=True)(y, h, s, prior)).astype(np.int32)
        o["lin_logits"] = np.asarray(mimo.LinearDetector("lmmse", "symbol", "app", "qam", m)(y, h, s))
        o["lin_hard"] = np.asarray(mimo.LinearDetector("lmmse", "symbol", "app", "qam", m, hard_out=True)(y, h, s)).astype(np.int32)
        for kname, v in o.items():
            out[f"m{ci}/{kname}"] = v
        print("mimo", (M, K, m), {kname: np.asarray(v).shape for kname, v in o.items() if kname not in ("y", "h", "s")}, flush=True)

    # ---- the OFDM wrappers on the c4 link of the receiver fixture
    G = np.load(os.path.join(ROOT, "tests", "golden", "ofdm_rx_ref_golden.npz"))
    L = LINKS["c4"]
    T, S, F, m = L["num_tx"], L["spt"], L["fft"], L["m"]
    rg = ofdm.ResourceGrid(num_ofdm_symbols=14, fft_size=F, subcarrier_spacing=15e3, num_tx=T, num_streams_per_tx=S,
                           cyclic_prefix_length=6, num_guard_carriers=list(L["guards"]), dc_null=True,
                           pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    sm = mimo.StreamManagement(np.ones([1, T]), S)
    y, hh, ev, no = G["c4/y"], G["c4/h_hat_lin"], G["c4/err_var_lin"], G["c4/no"]
    nd = int(rg.num_data_symbols)
    o = {}
    o["ep_logits"] = np.asarray(od.EPDetector("symbol", rg, sm, m, l=6, hard_out=False)(y, hh, ev, no))
    o["ep_hard"] = np.asarray(od.EPDetector("symbol", rg, sm, m, l=6, hard_out=True)(y, hh, ev, no)).astype(np.int32)
    o["kbest_hard"] = np.asarray(od.KBestDetector("symbol", T * S, L["kbest"], rg, sm, constellation_type="qam", num_bits_per_symbol=m,
                                                  hard_out=True)(y, hh, ev, no)).astype(np.int32)
    prior = (2.0 * rng.normal(size=(y.shape[0], T, S, nd, 1 << m))).astype(np.float32)
    o["pic_prior"] = prior
    const = mp.Constellation("qam", m)
    o["pic_logits"] = np.asarray(od.MMSEPICDetector(output="symbol", resource_grid=rg, stream_management=sm, demapping_method="maxlog",
                                                    constellation=const, num_iter=2, hard_out=False)(y, hh, prior, ev, no))
    o["pic_hard"] = np.asarray(od.MMSEPICDetector(output="symbol", resource_grid=rg, stream_management=sm, demapping_method="app",
                                                  constellation=const, num_iter=1, hard_out=True)(y, hh, prior, ev, no)).astype(np.int32)
    for kname, v in o.items():
        out[f"c4/{kname}"] = v
        print("c4", kname, v.shape, v.dtype, flush=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
