"""``sionna_amd.phy.ofdm.LMMSEInterpolator`` / ``tdl_freq_cov_mat`` / ``tdl_time_cov_mat`` against the reference's own classes
executed under the NumPy stand-in (tests/golden/lmmse_interp_ref_golden.npz) and against the oracle.  The interpolator is
library algebra (batched solves + matrix products through torch.linalg / matmul on the device the estimates live on), so
its arithmetic is checked HERE on host tensors through the internal ``_interpolate`` (test infrastructure: the public call
moves its inputs to the MI355X first and fails without one)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import lmmse_interp as li

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "lmmse_interp_ref_golden.npz"))
ORDERS = [str(o) for o in GOLD["orders"]]


def _pattern(g):
    return types.SimpleNamespace(mask=g["mask"], pilots=g["pilots"])


@pytest.mark.parametrize("gi", [0, 1, 2, 3])
@pytest.mark.parametrize("order", ORDERS)
def test_interpolator_matches_reference_execution_and_oracle(gi, order):
    from sionna_amd.phy.ofdm import LMMSEInterpolator
    g = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith(f"g{gi}/")}
    itp = LMMSEInterpolator(_pattern(g), g["cov_time"], torch.from_numpy(g["cov_freq"]), g["cov_space"], order=order)
    h, e = itp._interpolate(torch.from_numpy(g["h"]), torch.from_numpy(g["err_var"]))
    h, e = h.numpy(), e.numpy()
    href, eref = g[f"h_{order}"], g[f"e_{order}"]
    assert h.shape == href.shape and e.shape == eref.shape
    assert np.abs(h - href).max() <= 2e-4 * np.abs(href).max() and np.abs(e - eref).max() <= 2e-4 * max(np.abs(eref).max(), 1.0)
    ho, eo = li.lmmse_interpolate(g["mask"], g["pilots"], g["h"], g["err_var"], g["cov_time"], g["cov_freq"], g["cov_space"], order)
    assert np.abs(h - ho).max() <= 1e-9 * np.abs(ho).max() and np.abs(e - eo).max() <= 1e-9
    # err_var broadcastable like the estimator's (no leading dims)
    h2, _ = itp._interpolate(torch.from_numpy(g["h"][:1]), torch.from_numpy(g["err_var"][:1, :, :1]))
    assert np.abs(h2.numpy() - h[:1]).max() <= 1e-9 * np.abs(h).max()


@pytest.mark.parametrize("model", ["A", "C", "D", "E"])
def test_tdl_covariance_matrices(model):
    from sionna_amd.phy.ofdm import tdl_freq_cov_mat, tdl_time_cov_mat
    f = tdl_freq_cov_mat(model, 15e3, 12, 100e-9, precision="double").numpy()
    t = tdl_time_cov_mat(model, 10., 2.6e9, 71.4e-6, 14, precision="double").numpy()
    assert np.abs(f - GOLD[f"fcov_{model}"]).max() < 1e-12 and np.abs(t - GOLD[f"tcov_{model}"]).max() < 1e-12
    assert tdl_freq_cov_mat(model, 15e3, 12, 100e-9).dtype == torch.complex64


def test_constructor_checks():
    from sionna_amd.phy.ofdm import LMMSEInterpolator
    g = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith("g1/")}
    pp = _pattern(g)
    for bad in ("t", "t-t", "f-s", "t-f-x", "t-f-s-t"):
        with pytest.raises(AssertionError):
            LMMSEInterpolator(pp, g["cov_time"], g["cov_freq"], g["cov_space"], order=bad)
    with pytest.raises(AssertionError):
        LMMSEInterpolator(pp, g["cov_time"], g["cov_freq"], None, order="t-f-s")
