#!/usr/bin/env python3
"""Generates tests/golden/lmmse_interp_ref_golden.npz by EXECUTING the reference's own ``LMMSEInterpolator`` (with its
``LMMSEInterpolator1D`` and ``SpatialChannelFilter``), ``tdl_freq_cov_mat`` and ``tdl_time_cov_mat`` from the unmodified
source file ofdm/channel_estimation.py:736-2070 under the NumPy stand-in for TensorFlow (tools/ref_exec): a Kronecker pilot
pattern with two transmitters (so every stream sees the other's pilots as zero-power slots) and one with two streams per
transmitter, five interpolation orders, inputs as ``LSChannelEstimator`` delivers them (zeros at the zero-power slots).
``tf.linalg.lstsq(fast=False)`` is NumPy's pseudo-inverse there (float32 / complex64).  Run here (needs /root/reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "lmmse_interp_ref_golden.npz")
GRIDS = [dict(fft=24, num_tx=2, spt=1, ra=4, B=3), dict(fft=16, num_tx=1, spt=2, ra=2, B=2),
         # the pilot patterns of the reference's own test (test/unit/ofdm/test_ofdm_channel_estimation.py:829-883): one stream
         # with two pilots and three with one on a single OFDM symbol (rows with different pilot counts, many zero-power
         # slots), and a Kronecker pattern with a single pilot symbol
         dict(fft=12, num_tx=4, spt=1, ra=2, B=2, sparse=True), dict(fft=16, num_tx=2, spt=1, ra=2, B=2, pilot_symbols=[3])]
ORDERS = ["t-f", "f-t", "t-f-s", "s-f-t", "f-s-t"]


def main():
    import warnings
    warnings.simplefilter("ignore")
    from tools.gen_ofdm_rx_ref_golden import load
    from tools.ref_exec import tf_numpy
    mp, mimo, ofdm, od, ce, eq = load()
    # the power delay profiles tdl_*_cov_mat read (TDL-*.json next to the reference's TDL model)
    ce.models.__path__ = ["/root/reference/src/sionna/phy/channel/tr38901/models"]
    out = {"orders": np.array(ORDERS)}
    for gi, G in enumerate(GRIDS):
        rng = np.random.default_rng(77 + gi)
        if G.get("sparse"):
            m = np.zeros([4, 1, 14, 12], bool)
            m[..., 5, :] = True
            p = np.zeros([4, 1, 12], np.complex64)
            p[0, 0, [0, 11]], p[1, 0, 1], p[2, 0, 5], p[3, 0, 10] = 1, 1, 1, 1
            pp = ofdm.PilotPattern(m, p)
        else:
            rg = ofdm.ResourceGrid(num_ofdm_symbols=14, fft_size=G["fft"], subcarrier_spacing=30e3, num_tx=G["num_tx"],
                                   num_streams_per_tx=G["spt"], pilot_pattern="kronecker",
                                   pilot_ofdm_symbol_indices=G.get("pilot_symbols", [2, 11]))
            pp = rg.pilot_pattern
        pilots, mask = np.asarray(pp.pilots).astype(np.complex64), np.asarray(pp.mask).astype(np.uint8)
        F, T, ra, B = G["fft"], 14, G["ra"], G["B"]
        cf = np.asarray(ce.tdl_freq_cov_mat("A", 30e3, F, 300e-9)).astype(np.complex64)
        ct = np.asarray(ce.tdl_time_cov_mat("A", 30., 3.5e9, 1. / 30e3 * (1 + 0.07), T)).astype(np.complex64)
        i = np.arange(ra)
        cs = (0.7 ** np.abs(i[:, None] - i[None, :]) * np.exp(0.3j * (i[:, None] - i[None, :]))).astype(np.complex64)
        shape = (B, 1, ra) + pilots.shape
        live = (np.abs(pilots) > 0)[None, None, None]
        h = ((rng.normal(size=shape) + 1j * rng.normal(size=shape)) / np.sqrt(2)).astype(np.complex64) * live
        no = np.array([0.02, 0.2, 1.0])[:B].reshape(B, 1, 1, 1, 1, 1)
        ev = (no / np.where(live, np.abs(pilots) ** 2, 1) * live).astype(np.float32) * np.ones(shape, np.float32)
        o = dict(pilots=pilots, mask=mask, cov_freq=cf, cov_time=ct, cov_space=cs, h=h, err_var=ev)
        for order in ORDERS:
            itp = ce.LMMSEInterpolator(pp, tf_numpy._t(ct), tf_numpy._t(cf), tf_numpy._t(cs), order=order)
            hh, ee = itp(h, ev)
            o[f"h_{order}"], o[f"e_{order}"] = np.asarray(hh), np.asarray(ee)
            print(gi, order, np.asarray(hh).shape, float(np.abs(np.asarray(hh)).mean()), float(np.asarray(ee).mean()), flush=True)
        for k, v in o.items():
            out[f"g{gi}/{k}"] = v
    for model in ("A", "C", "D", "E"):
        out[f"fcov_{model}"] = np.asarray(ce.tdl_freq_cov_mat(model, 15e3, 12, 100e-9, precision="double"))
        out[f"tcov_{model}"] = np.asarray(ce.tdl_time_cov_mat(model, 10., 2.6e9, 71.4e-6, 14, precision="double"))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
