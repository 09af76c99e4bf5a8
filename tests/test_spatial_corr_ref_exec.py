"""Spatial correlation of the flat-fading channel (Simple_MIMO_Simulation.ipynb cell 44) against the reference's OWN
``exp_corr_mat`` / ``one_ring_corr_mat`` / ``KroneckerModel`` / ``PerColumnModel`` executed under the NumPy stand-in for
TensorFlow (tests/golden/spatial_corr_ref_golden.npz, tools/gen_spatial_corr_ref_golden.py).  CPU: the correlation
matrices, and the [M K, M K] matrices this build hands to ``samd_spatial_corr_c64`` applied in NumPy; the kernel itself:
tests/test_gpu_ofdm.py::test_spatial_correlation_models_match_reference_execution."""
import os

import numpy as np

import sionna_amd.phy as phy

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "spatial_corr_ref_golden.npz"))
ch = phy.channel


def close(a, b, tol=2e-6):
    return np.abs(np.asarray(a) - b).max() <= tol * max(1.0, np.abs(b).max())


def test_correlation_matrices():
    assert close(ch.exp_corr_mat(0.4, 4), G["exp_04_4"]) and close(ch.exp_corr_mat(0.7, 16), G["exp_07_16"])
    assert close(ch.exp_corr_mat(0.5 + 0.3j, 5), G["exp_c_5"])
    assert close(ch.exp_corr_mat(np.array([0.0, 0.2, -0.6]), 3), G["exp_batch"])          # a = 0: the identity (0 ** 0 = 1)
    assert close(ch.one_ring_corr_mat(30.0, 8), G["ring_30_8"], 1e-5)
    assert close(ch.one_ring_corr_mat(-45.0, 4, d_h=0.7, sigma_phi_deg=8), G["ring_m45_4"], 1e-5)


def _applied(model, h, mats=[]):
    """what the model would launch: its [M K, M K] matrix on the rx-major vector of every h"""
    got = {}
    model._apply = lambda hh, mat: got.setdefault("y", np.einsum("ij,bj->bi", mat, hh.reshape(hh.shape[0], -1)).reshape(hh.shape))
    model(h)
    return got["y"]


def test_kronecker_and_per_column_models():
    h, h2 = G["h_16x4"], G["h_5x3"]
    assert close(_applied(ch.KroneckerModel(G["exp_04_4"], G["exp_07_16"]), h), G["kron_16x4"], 1e-5)
    assert close(_applied(ch.KroneckerModel(None, G["exp_07_16"]), h), G["kron_rx_only"], 1e-5)
    assert close(_applied(ch.KroneckerModel(G["exp_04_4"], None), h), G["kron_tx_only"], 1e-5)
    assert close(_applied(ch.KroneckerModel(G["r_tx3"], G["exp_c_5"]), h2), G["kron_5x3"], 1e-5)      # complex: the conjugate matters
    assert close(_applied(ch.PerColumnModel(G["r_cols"]), h2), G["percol_5x3"], 1e-5)
    assert ch.KroneckerModel()(h) is h
