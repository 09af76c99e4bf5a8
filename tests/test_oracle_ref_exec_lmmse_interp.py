"""Pins the LMMSE channel interpolator (oracle/lmmse_interp.py) to the reference's OWN ``LMMSEInterpolator`` /
``LMMSEInterpolator1D`` / ``SpatialChannelFilter`` and ``tdl_freq_cov_mat`` / ``tdl_time_cov_mat`` executed here
(tests/golden/lmmse_interp_ref_golden.npz from tools/gen_lmmse_interp_ref_golden.py: ofdm/channel_estimation.py:736-2070
under the NumPy stand-in for TensorFlow, complex64 with NumPy's pseudo-inverse behind ``tf.linalg.lstsq``).  The oracle
works in complex128 with a plain solve; covariance matrices: 1e-12; interpolated estimates and error variances: 2e-4 of scale
(the reference side inverts TDL covariance matrices with condition numbers ~1e4 in single precision)."""
import os

import numpy as np
import pytest

from oracle import lmmse_interp as li

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "lmmse_interp_ref_golden.npz"))
ORDERS = [str(o) for o in GOLD["orders"]]


@pytest.mark.parametrize("gi", [0, 1, 2, 3])
@pytest.mark.parametrize("order", ORDERS)
def test_lmmse_interpolator_matches_reference_execution(gi, order):
    g = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith(f"g{gi}/")}
    h, e = li.lmmse_interpolate(g["mask"], g["pilots"], g["h"], g["err_var"], g["cov_time"], g["cov_freq"], g["cov_space"], order)
    href, eref = g[f"h_{order}"], g[f"e_{order}"]
    assert h.shape == href.shape and e.shape == eref.shape
    assert np.abs(h - href).max() <= 2e-4 * np.abs(href).max(), np.abs(h - href).max()
    assert np.abs(e - eref).max() <= 2e-4 * max(np.abs(eref).max(), 1.0), np.abs(e - eref).max()
    assert e.min() >= 0.0


@pytest.mark.parametrize("model", ["A", "C", "D", "E"])
def test_tdl_covariance_matrices(model):
    f = li.tdl_freq_cov_mat(model, 15e3, 12, 100e-9)
    t = li.tdl_time_cov_mat(model, 10., 2.6e9, 71.4e-6, 14)
    assert np.abs(f - GOLD[f"fcov_{model}"]).max() < 1e-12 and np.abs(t - GOLD[f"tcov_{model}"]).max() < 1e-12
    assert np.allclose(np.diagonal(f), 1.0) and np.allclose(f, np.conj(f.T))


def test_interpolation_recovers_a_channel_drawn_from_the_model():
    """property: at high SNR the estimate of a channel drawn from the covariance model is close to it everywhere, and the
    reported error variance is of the order of the observed error"""
    g = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith("g1/")}
    rng = np.random.default_rng(3)
    T, F = g["mask"].shape[2:]
    def root(c):                                                      # (the TDL covariance matrices are nearly singular)
        c = c.astype(np.complex128)
        w_, v_ = np.linalg.eigh((c + c.conj().T) / 2)
        return v_ * np.sqrt(np.clip(w_, 0, None))
    lt, lf = root(g["cov_time"]), root(g["cov_freq"])
    w = (rng.normal(size=(200, T, F)) + 1j * rng.normal(size=(200, T, F))) / np.sqrt(2)
    chan = np.einsum("ts,bsf->btf", lt, np.einsum("fg,btg->btf", lf, w))              # [200, T, F], covariance time x frequency
    pm = li.build_pilot_mask(g["mask"], g["pilots"])
    no = 1e-3
    hp = np.zeros((200, 1, 1) + g["pilots"].shape, np.complex128)
    ep = np.zeros((200, 1, 1) + g["pilots"].shape)
    for st in range(pm.shape[1]):
        pos = np.flatnonzero(pm[0, st].reshape(-1) != 0)
        live = pm[0, st].reshape(-1)[pos] == 1
        noise = np.sqrt(no / 2) * (rng.normal(size=(200, live.sum())) + 1j * rng.normal(size=(200, live.sum())))
        hp[:, 0, 0, 0, st, np.flatnonzero(live)] = chan.reshape(200, -1)[:, pos[live]] + noise
        ep[:, 0, 0, 0, st, np.flatnonzero(live)] = no
    h, e = li.lmmse_interpolate(g["mask"], g["pilots"], hp, ep, g["cov_time"], g["cov_freq"], None, "f-t")
    mse = np.mean(np.abs(h[:, 0, 0, 0, 0] - chan) ** 2)
    assert mse < 0.05 and 0.2 < np.mean(e[:, 0, 0, 0, 0]) / mse < 5.0, (mse, np.mean(e[:, 0, 0, 0, 0]))
