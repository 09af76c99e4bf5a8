#!/usr/bin/env python3
"""Generates tests/golden/ep_steps_ref_golden.npz by EXECUTING the four public step methods of the reference's
``EPDetector`` (mimo/detection.py:1166-1227: compute_sigma_mu, compute_v_x_obs, compute_v_x, update_lam_gam) under the NumPy
stand-in for TensorFlow, two iterations chained on a random real-valued 8 x 4 problem per constellation size."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "ep_steps_ref_golden.npz")


def main():
    from tools.gen_ofdm_rx_ref_golden import load
    mp, mimo, ofdm, od, ce, eq = load()
    out = {}
    for nb in (2, 4, 6):
        rng = np.random.default_rng(nb)
        det = mimo.EPDetector("bit", nb, l=2, beta=0.7)
        B, m, n = 5, 8, 4
        h = rng.normal(size=(B, m, n)).astype(np.float32) / np.sqrt(2)
        y = rng.normal(size=(B, m, 1)).astype(np.float32)
        hth = np.matmul(np.swapaxes(h, -1, -2), h)
        hty = np.matmul(np.swapaxes(h, -1, -2), y)
        no = np.full((1, 1, 1), 0.5, np.float32)
        lam = np.full((B, n), 1. / float(np.asarray(det._es)), np.float32)
        gam = np.zeros((B, n), np.float32)
        o = dict(hth=hth, hty=hty, no=no, lam_init=lam, gam_init=gam)
        for it in range(2):
            sigma, mu = det.compute_sigma_mu(hth, hty, no, lam, gam)
            v_obs, x_obs = det.compute_v_x_obs(sigma, mu, lam, gam)
            v, x, logits = det.compute_v_x(v_obs, x_obs)
            lam, gam = det.update_lam_gam(v, v_obs, x, x_obs, lam, gam)
            for name, val in dict(sigma=sigma, mu=mu, v_obs=v_obs, x_obs=x_obs, v=v, x=x, logits=logits, lam=lam, gam=gam).items():
                o[f"{name}{it}"] = np.asarray(val)
            lam, gam = np.asarray(lam), np.asarray(gam)
        for k, v_ in o.items():
            out[f"nb{nb}/{k}"] = v_
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
