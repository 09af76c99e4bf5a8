#!/usr/bin/env python3
"""Probe for the open CP-2 discrepancy: the HIP CDL generator against oracle/cdl.py on the same Philox draws in EXACTLY
the configuration of MIMO_OFDM_Transmissions_over_CDL.ipynb cell 76 (CDL-C, 100 ns, uplink 4 -> 8, 3 m/s, 1052 time
steps at 1.08 MHz), which tests/test_gpu_cdl.py covers only at 33 x <= 14 steps of 210 kHz; then second-order statistics
of the product's own time-domain channel over many realisations."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import sionna_amd.phy as phy
    from oracle import cdl as oc
    fc, T, fs = 2.6e9, 1052, 72 * 15e3
    t38 = phy.channel.tr38901
    ut = t38.AntennaArray(num_rows=1, num_cols=2, polarization="dual", polarization_type="cross", antenna_pattern="38.901", carrier_frequency=fc)
    bs = t38.AntennaArray(num_rows=1, num_cols=4, polarization="dual", polarization_type="cross", antenna_pattern="38.901", carrier_frequency=fc)
    cdl = t38.CDL(model="C", delay_spread=100e-9, carrier_frequency=fc, ut_array=ut, bs_array=bs, direction="uplink", min_speed=3.0)
    ref = oc.CDL("C", 100e-9, fc, oc.AntennaArray(1, 2, "dual", "cross", "38.901", fc), oc.AntennaArray(1, 4, "dual", "cross", "38.901", fc),
                 "uplink", min_speed=3.0)
    for B in (6, 1024):
        phy.config.seed = 99
        a, tau = cdl(B, T, fs)
        a = a.cpu().numpy()
        ar, taur = ref(99, 0, min(B, 6), T, fs)
        if B > 6:                                   # the last examples of a big batch: oracle on the same draws needs the full batch layout
            ar_full_first = ar
        scale = np.sqrt(np.mean(np.abs(ar) ** 2))
        print(f"B={B}: first {ar.shape[0]} examples vs oracle: max |diff| / rms {np.abs(a[:ar.shape[0]] - ar).max() / scale:.3e}"
              f"  (tau {np.abs(tau.cpu().numpy()[:ar.shape[0]] - taur).max():.2e})", flush=True)
        if B > 6:
            p = np.mean(np.sum(np.abs(a[..., 0]) ** 2, axis=(2, 4)), axis=0)[0, 0]
            print("   per-cluster power, batch mean:", np.round(p[:6], 4), "... total", float(p.sum()))
            v = a[:, 0, :, 0, :, :, 0].reshape(B, 32, -1)
            cov = np.einsum("bin,bjn->ij", v, np.conj(v)) / B
            rho_t = np.sum(a[..., -1] * np.conj(a[..., 0])) / np.sum(np.abs(a[..., 0]) ** 2)
            print("   correlation first / last time step:", complex(rho_t), " cov diag mean", float(np.real(np.diag(cov)).mean()))
            # are the examples of a big batch independent?  correlation between neighbouring examples
            x = a[:, 0, 0, 0, 0, :, 0]
            print("   |corr| between examples b and b+1:", float(np.abs(np.sum(x[1:] * np.conj(x[:-1]))) / np.sum(np.abs(x) ** 2)),
                  " b and b+512:", float(np.abs(np.sum(x[512:] * np.conj(x[:-512]))) / np.sum(np.abs(x[512:]) ** 2)))


if __name__ == "__main__":
    main()
