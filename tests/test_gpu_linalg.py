"""csrc/mimo_linalg.hip through the package's inv_cholesky / matrix_pinv / whiten_channel / lmmse_matrix (and the complex <->
real-valued representation helpers) against oracle/linalg.py and the reference-executed fixture: complex128 at 1e-10,
complex64 at 3e-6 x the condition number of the factorised matrix x the largest entry."""
import os

import numpy as np
import pytest
import torch

from oracle import linalg as ol

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "linalg_ref_golden.npz"))
SIZES = [tuple(int(v) for v in r) for r in G["sizes"]]


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    return p


def _np(t):
    return t.detach().cpu().numpy()


def _chk(got, ref, dbl, cond=1.0):
    """complex128: 1e-10; complex64: 3e-6 x the condition number of the matrix the function factorises x the largest entry"""
    got = _np(got)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert got.dtype == (ref.dtype if dbl else {np.dtype("complex128"): np.dtype("complex64"), np.dtype("float64"): np.dtype("float32")}[ref.dtype])
    tol = 1e-10 if dbl else 3e-6 * max(10.0, cond) * max(1.0, float(np.max(np.abs(ref))))
    assert np.allclose(got, ref, rtol=0, atol=tol), (float(np.max(np.abs(got - ref))), tol)


def _cond(a):
    return float(np.max(np.linalg.cond(a)))


@pytest.mark.parametrize("dbl", [True, False])
@pytest.mark.parametrize("i", range(len(SIZES)))
def test_helpers_vs_reference_fixture(phy, i, dbl):
    m, k = SIZES[i]
    cd, rd = (np.complex128, np.float64) if dbl else (np.complex64, np.float32)
    y, h, s = G[f"y{i}"].astype(cd), G[f"h{i}"].astype(cd), G[f"s{i}"].astype(cd)
    h64, s64 = G[f"h{i}"], G[f"s{i}"]
    hh = np.conj(np.swapaxes(h64, -1, -2))
    _chk(phy.utils.inv_cholesky(s), G[f"inv_chol{i}"], dbl, _cond(s64))
    _chk(phy.utils.inv_cholesky((s64.real + np.eye(m)).astype(rd)), G[f"inv_chol_real{i}"], dbl, _cond(s64.real + np.eye(m)))
    if k <= m:
        _chk(phy.utils.matrix_pinv(h), G[f"pinv{i}"], dbl, _cond(hh @ h64))
    yw, hw, sw = phy.mimo.whiten_channel(y, h, s)
    _chk(yw, G[f"yw{i}"], dbl, _cond(s64))
    _chk(hw, G[f"hw{i}"], dbl, _cond(s64))
    assert np.array_equal(_np(sw), np.broadcast_to(np.eye(m), s.shape))
    assert len(phy.mimo.whiten_channel(y, h, s, return_s=False)) == 2
    prec = "double" if dbl else "single"
    _chk(phy.mimo.lmmse_matrix(h, s, precision=prec), G[f"g{i}"], dbl, _cond(h64 @ hh + s64))
    _chk(phy.mimo.lmmse_matrix(h, precision=prec), G[f"g_white{i}"], dbl, _cond(hh @ h64 + np.eye(k)))
    yr, hr, sr = phy.mimo.complex2real_channel(torch.from_numpy(y), torch.from_numpy(h), torch.from_numpy(s))
    for got, key in ((yr, "yr"), (hr, "hr"), (sr, "sr")):
        assert np.array_equal(_np(got), G[f"{key}{i}"].astype(rd))
    yc, hc, sc = phy.mimo.real2complex_channel(yr, hr, sr)
    assert np.array_equal(_np(yc), y) and np.array_equal(_np(hc), h) and np.array_equal(_np(sc), s)


def test_batch_shapes_and_limits(phy):
    rng = np.random.default_rng(3)
    h = (rng.normal(size=(5, 7, 4, 2)) + 1j * rng.normal(size=(5, 7, 4, 2))).astype(np.complex64)
    s = np.eye(4, dtype=np.complex64) * 0.5                              # broadcast over the leading dimensions
    g = phy.mimo.lmmse_matrix(h, s)
    assert tuple(g.shape) == (5, 7, 2, 4)
    assert np.allclose(_np(g), ol.lmmse_matrix(h, np.broadcast_to(s, (5, 7, 4, 4))), atol=1e-4)
    assert tuple(phy.utils.inv_cholesky(np.zeros((0, 3, 3), np.complex64)).shape) == (0, 3, 3)
    with pytest.raises(ValueError):
        phy.utils.inv_cholesky(np.broadcast_to(np.eye(17, dtype=np.complex64), (2, 17, 17)))
    with pytest.raises(ValueError):
        phy.utils.matrix_pinv(np.ones((2, 2, 3), np.complex64))           # K > M: not of full column rank
    # the equaliser built from the helpers is the fused equaliser (mimo/equalization.py:205-231)
    y = (rng.normal(size=(5, 7, 4)) + 1j * rng.normal(size=(5, 7, 4))).astype(np.complex64)
    sb = np.broadcast_to(s, (5, 7, 4, 4))
    yw, hw = phy.mimo.whiten_channel(y, h, sb, return_s=False)
    gw = _np(phy.mimo.lmmse_matrix(hw))
    gy = np.einsum("...km,...m->...k", gw, _np(yw))
    d = np.einsum("...km,...mk->...k", gw, _np(hw))
    x_hat, no_eff = phy.mimo.lmmse_equalizer(y, h, sb)
    assert np.allclose(gy / d, _np(x_hat), atol=2e-4) and np.allclose(np.real(1 / d - 1), _np(no_eff), rtol=2e-3, atol=1e-4)
