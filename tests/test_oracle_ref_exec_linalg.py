"""oracle/linalg.py (the specification of csrc/mimo_linalg.hip) against the reference's own inv_cholesky / matrix_pinv /
whiten_channel / lmmse_matrix and its complex <-> real-valued representation helpers EXECUTED under the NumPy stand-in for
TensorFlow (tests/golden/linalg_ref_golden.npz, tools/gen_linalg_ref_golden.py)."""
import os

import numpy as np
import pytest

from oracle import linalg as ol

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "linalg_ref_golden.npz"))
SIZES = [tuple(int(v) for v in r) for r in G["sizes"]]


def _close(a, b, tol=1e-10):
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert np.allclose(a, b, rtol=tol, atol=tol), float(np.max(np.abs(a - b)))


@pytest.mark.parametrize("i", range(len(SIZES)))
def test_linalg_oracle_vs_executed_reference(i):
    m, k = SIZES[i]
    y, h, s = G[f"y{i}"], G[f"h{i}"], G[f"s{i}"]
    _close(ol.inv_cholesky(s), G[f"inv_chol{i}"])
    _close(ol.inv_cholesky(s.real + np.eye(m)), G[f"inv_chol_real{i}"])
    if k <= m:
        _close(ol.matrix_pinv(h), G[f"pinv{i}"])
    yw, hw = ol.whiten_channel(y, h, s)
    _close(yw, G[f"yw{i}"])
    _close(hw, G[f"hw{i}"])
    _close(ol.lmmse_matrix(h, s), G[f"g{i}"])
    _close(ol.lmmse_matrix(h), G[f"g_white{i}"])


def test_linalg_identities():
    """what the outputs mean: L^-1 S L^-H = I, pinv(A) A = I, G = H^H (H H^H + S)^-1"""
    for i, (m, k) in enumerate(SIZES):
        h, s = G[f"h{i}"], G[f"s{i}"]
        li = ol.inv_cholesky(s)
        assert np.allclose(li @ s @ np.conj(np.swapaxes(li, -1, -2)), np.eye(m), atol=1e-9)
        assert np.allclose(np.triu(li, 1), 0)
        if k <= m:
            assert np.allclose(ol.matrix_pinv(h) @ h, np.eye(k), atol=1e-9)
        hh = np.conj(np.swapaxes(h, -1, -2))
        assert np.allclose(ol.lmmse_matrix(h, s) @ (h @ hh + s), hh, atol=1e-9)
