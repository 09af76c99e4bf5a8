#!/usr/bin/env python3
"""Generates tests/golden/precoding_ref_golden.npz by EXECUTING the reference's own ``rzf_precoding_matrix`` /
``rzf_precoder`` (mimo/precoding.py:12-244) and ``RZFPrecoder`` with its effective channel (ofdm/precoding.py:15-177) under
the NumPy stand-in for TensorFlow: the downlink link of MIMO_OFDM_Transmissions_over_CDL.ipynb in small (one base station with
8 antennas precoding 4 streams to a 4-antenna terminal; guard carriers and DC null), with and without regularisation.
Run here (needs /root/reference); the fixture travels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "precoding_ref_golden.npz")


def cn(rng, shape):
    return ((rng.normal(size=shape) + 1j * rng.normal(size=shape)) / np.sqrt(2)).astype(np.complex64)


def main():
    from tools.gen_ofdm_rx_ref_golden import load
    from tools.ref_exec.loader import reference
    mp, mimo, ofdm, od, ce, eq = load()
    ref = reference()
    pm = ref.load("sionna.phy.mimo.precoding")
    for k, v in vars(pm).items():
        if not k.startswith("_"):
            setattr(mimo, k, v)
    po = ref.load("sionna.phy.ofdm.precoding")
    rng = np.random.default_rng(909)
    out = {}
    for i, (K, M, alpha) in enumerate([(2, 4, 0.0), (4, 8, 0.0), (4, 8, 0.3), (1, 2, 0.0), (4, 4, 0.05)]):
        h, x = cn(rng, (6, 3, K, M)), cn(rng, (6, 3, K))
        al = np.float32(alpha) if i != 2 else (alpha * (1 + rng.random((6, 3)))).astype(np.float32)
        xp, g = pm.rzf_precoder(x, h, alpha=al, return_precoding_matrix=True)
        out[f"m{i}/h"], out[f"m{i}/x"], out[f"m{i}/alpha"] = h, x, np.asarray(al)
        out[f"m{i}/x_precoded"], out[f"m{i}/g"] = np.asarray(xp), np.asarray(g)
    # OFDM level: 1 transmitter (8 antennas, 4 streams) -> 1 receiver with 4 antennas, fft 38 with guards and DC null
    rg = ofdm.ResourceGrid(num_ofdm_symbols=14, fft_size=38, subcarrier_spacing=15e3, num_tx=1, num_streams_per_tx=4,
                           cyclic_prefix_length=6, num_guard_carriers=[3, 2], dc_null=True, pilot_pattern="kronecker",
                           pilot_ofdm_symbol_indices=[2, 11])
    sm = mimo.StreamManagement(np.array([[1]]), 4)
    B = 2
    x_rg = cn(rng, (B, 1, 4, 14, 38))
    h = cn(rng, (B, 1, 4, 1, 8, 14, 38))
    for tag, alpha in (("zf", 0.0), ("rzf", 0.2)):
        xp, heff = po.RZFPrecoder(rg, sm, return_effective_channel=True)(x_rg, h, alpha=np.float32(alpha))
        out[f"o_{tag}/x_precoded"], out[f"o_{tag}/h_eff"] = np.asarray(xp), np.asarray(heff)
    out["o/x_rg"], out["o/h"] = x_rg, h
    out["o/precoding_ind"] = np.asarray(sm.precoding_ind).astype(np.int32)
    out["o/effective_subcarrier_ind"] = np.asarray(rg.effective_subcarrier_ind).astype(np.int32)
    for k, v in out.items():
        print(k, np.asarray(v).shape, np.asarray(v).dtype)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
