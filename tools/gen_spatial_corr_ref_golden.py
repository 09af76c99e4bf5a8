#!/usr/bin/env python3
"""Generates tests/golden/spatial_corr_ref_golden.npz by EXECUTING the reference's ``exp_corr_mat`` / ``one_ring_corr_mat``
(channel/utils.py:1490-1652) and ``KroneckerModel`` / ``PerColumnModel`` (channel/spatial_correlation.py:41-195) under the
NumPy stand-in for TensorFlow: the correlation matrices and the models' outputs on fixed channel matrices (4 x 16 as in
Simple_MIMO_Simulation.ipynb cell 44, and 3 x 5 with complex correlation).  Run here (needs /root/reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "spatial_corr_ref_golden.npz")


def main():
    from tools.ref_exec import tf_numpy
    from tools.ref_exec.loader import reference
    ref = reference()
    ref.load_utils()
    tf = ref.tf

    class LinearOperatorToeplitz:                                  # dense Toeplitz matrix from first column / first row
        def __init__(self, col, row):
            self.col, self.row = np.asarray(col), np.asarray(row)

        def to_dense(self):
            n = self.col.shape[-1]
            i = np.arange(n)
            d = i[:, None] - i[None, :]
            return tf_numpy._t(np.where(d >= 0, self.col[..., np.abs(d)], self.row[..., np.abs(d)]))
    tf.linalg.LinearOperatorToeplitz = LinearOperatorToeplitz
    tf.linalg.matrix_transpose = lambda a, **k: tf_numpy._t(np.swapaxes(np.asarray(a), -1, -2))
    chan = sys.modules["sionna.phy.channel"]
    cu = ref.load("sionna.phy.channel.utils")
    for k, v in vars(cu).items():
        if not k.startswith("_"):
            setattr(chan, k, v)
    sc = ref.load("sionna.phy.channel.spatial_correlation")
    rng = np.random.default_rng(8)
    out = {}
    out["exp_04_4"] = np.asarray(cu.exp_corr_mat(0.4, 4))
    out["exp_07_16"] = np.asarray(cu.exp_corr_mat(0.7, 16))
    out["exp_c_5"] = np.asarray(cu.exp_corr_mat(np.complex64(0.5 + 0.3j), 5))
    out["exp_batch"] = np.asarray(cu.exp_corr_mat(np.array([0.0, 0.2, -0.6], np.complex64), 3))
    out["ring_30_8"] = np.asarray(cu.one_ring_corr_mat(30.0, 8))
    out["ring_m45_4"] = np.asarray(cu.one_ring_corr_mat(-45.0, 4, d_h=0.7, sigma_phi_deg=8))
    h = ((rng.normal(size=(6, 16, 4)) + 1j * rng.normal(size=(6, 16, 4))) / np.sqrt(2)).astype(np.complex64)
    out["h_16x4"] = h
    out["kron_16x4"] = np.asarray(sc.KroneckerModel(out["exp_04_4"], out["exp_07_16"])(h))
    out["kron_rx_only"] = np.asarray(sc.KroneckerModel(None, out["exp_07_16"])(h))
    out["kron_tx_only"] = np.asarray(sc.KroneckerModel(out["exp_04_4"], None)(h))
    h2 = ((rng.normal(size=(7, 5, 3)) + 1j * rng.normal(size=(7, 5, 3))) / np.sqrt(2)).astype(np.complex64)
    r_tx3 = np.asarray(cu.exp_corr_mat(np.complex64(0.3 - 0.5j), 3))
    out["h_5x3"], out["r_tx3"] = h2, r_tx3
    out["kron_5x3"] = np.asarray(sc.KroneckerModel(r_tx3, out["exp_c_5"])(h2))
    r_cols = np.stack([np.asarray(cu.exp_corr_mat(np.complex64(a), 5)) for a in (0.2, 0.5 + 0.3j, -0.7j)])
    out["r_cols"] = r_cols
    out["percol_5x3"] = np.asarray(sc.PerColumnModel(r_cols)(h2))
    np.savez_compressed(OUT, **out)
    print({k: v.shape for k, v in out.items()})
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
