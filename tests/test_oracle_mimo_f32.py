"""CPU tests of oracle/mimo_f32.py (the float32, operation-order-defined restatement of the linear MIMO
equalisers): it must agree with the complex128 restatement of the reference formulas (oracle/ofdm.py) within the
float32 conditioning bound, and satisfy the invariants the reference's own tests assert
(test/unit/mimo/test_mimo_equalizers.py:55-102 error statistics; test_mimo_utils.py whitening)."""
import numpy as np
import pytest

from oracle import mimo_f32 as f32, ofdm as o, mapping as omap

EPS = 2.0 ** -24


def _problem(rng, n, m, k, coloured=True, no=0.05):
    h = ((rng.normal(size=(n, m, k)) + 1j * rng.normal(size=(n, m, k))) / np.sqrt(2)).astype(np.complex64)
    q = (rng.normal(size=(n, m, m)) + 1j * rng.normal(size=(n, m, m))).astype(np.complex64)
    s = ((0.1 * q @ np.conj(np.swapaxes(q, -1, -2)) if coloured else 0) + no * np.eye(m)).astype(np.complex64)
    x = omap.qam(4)[rng.integers(0, 16, (n, k))]
    w = 0.1 * (rng.normal(size=(n, m)) + 1j * rng.normal(size=(n, m)))
    y = ((h @ x[..., None])[..., 0] + w).astype(np.complex64)
    return y, h, s, x


@pytest.mark.parametrize("m,k", [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2), (8, 4)])
@pytest.mark.parametrize("whiten", [True, False])
def test_lmmse_f32_within_conditioning_of_complex128(m, k, whiten):
    rng = np.random.default_rng(m * 10 + k)
    y, h, s, _ = _problem(rng, 2000, m, k)
    xf, nf = f32.lmmse_equalizer(y, h, s, whiten)
    xr, nr = o.lmmse_equalizer(y, h, s, whiten)
    assert xf.dtype == np.complex64 and nf.dtype == np.float32
    # forward error of a backward-stable float32 solve: ~ cond * eps; the two solves involve S and H^H S^-1 H + I
    s64, h64 = s.astype(np.complex128), h.astype(np.complex128)
    a = np.conj(np.swapaxes(h64, -1, -2)) @ np.linalg.solve(s64, h64) + np.eye(k)
    cond = np.linalg.cond(s64) * np.linalg.cond(a)
    bound = 8 * (m + k) * EPS * cond
    ex = np.max(np.abs(xf - xr), axis=-1) / np.max(np.abs(xr), axis=-1)
    # no_eff = 1/d - 1 with d = (G H)_kk in (0, 1): the subtraction amplifies the error of d by 1/(1 - d) = 1 + 1/no_eff
    en = np.max(np.abs(nf - nr) / np.abs(nr) / (1 + 1 / np.abs(nr)), axis=-1)
    assert np.all(ex <= bound), float(np.max(ex / bound))
    assert np.all(en <= bound), float(np.max(en / bound))
    # and in absolute terms the float32 results are far inside the old 2e-3 test bar
    assert np.max(ex) < 5e-4 and np.allclose(nf, nr, rtol=5e-4)


@pytest.mark.parametrize("m,k", [(1, 1), (2, 1), (4, 2), (4, 4), (8, 4)])
def test_zf_mf_f32_vs_complex128(m, k):
    rng = np.random.default_rng(m * 3 + k)
    y, h, s, _ = _problem(rng, 1000, m, k, no=0.2)
    for fn, ref in ((f32.zf_equalizer, o.zf_equalizer), (f32.mf_equalizer, o.mf_equalizer)):
        x, ne = fn(y, h, s)
        xr, nr = ref(y, h, s)
        h64 = h.astype(np.complex128)
        cond = np.linalg.cond(np.conj(np.swapaxes(h64, -1, -2)) @ h64) if fn is f32.zf_equalizer else np.ones(len(y))
        bound = 16 * (m + k) * EPS * cond
        assert np.all(np.max(np.abs(x - xr), -1) / np.max(np.abs(xr), -1) <= bound)
        assert np.all(np.max(np.abs(ne - nr) / np.abs(nr), -1) <= 4 * bound)


def test_lmmse_f32_error_statistics():
    """test_mimo_equalizers.py:55-102: the estimate is unbiased and its error variance equals no_eff."""
    rng = np.random.default_rng(3)
    n, m, k, no = 100000, 8, 4, 0.3
    h = ((rng.normal(size=(n, m, k)) + 1j * rng.normal(size=(n, m, k))) / np.sqrt(2)).astype(np.complex64)
    x = omap.qam(2)[rng.integers(0, 4, (n, k))]
    w = np.sqrt(no / 2) * (rng.normal(size=(n, m)) + 1j * rng.normal(size=(n, m)))
    y = ((h @ x[..., None])[..., 0] + w).astype(np.complex64)
    s = np.broadcast_to((no * np.eye(m)).astype(np.complex64), (n, m, m))
    for whiten in (True, False):
        xh, ne = f32.lmmse_equalizer(y, h, s, whiten)
        err = xh - x
        assert abs(np.mean(err)) < 5e-3
        assert abs(np.mean(np.abs(err) ** 2) - np.mean(ne)) / np.mean(ne) < 2e-2


def test_whitened_and_unwhitened_forms_agree():
    """mimo/equalization.py:175-193: both branches evaluate the same estimator."""
    rng = np.random.default_rng(5)
    y, h, s, _ = _problem(rng, 500, 4, 2)
    a, an = f32.lmmse_equalizer(y, h, s, True)
    b, bn = f32.lmmse_equalizer(y, h, s, False)
    assert np.allclose(a, b, rtol=2e-3, atol=2e-4) and np.allclose(an, bn, rtol=2e-3)


@pytest.mark.parametrize("cfg", ["c4", "two_tx", "two_rx"])
def test_ofdm_equalize_f32_vs_complex128(cfg):
    kw = dict(cyclic_prefix_length=6, dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    if cfg == "c4":
        rg, assoc, ns = o.ResourceGrid(14, 76, 15e3, num_tx=1, num_streams_per_tx=2, num_guard_carriers=[5, 6], **kw), [[1]], 2
    elif cfg == "two_tx":
        rg, assoc, ns = o.ResourceGrid(14, 76, 15e3, num_tx=2, num_streams_per_tx=1, num_guard_carriers=[5, 6], **kw), [[1, 1]], 1
    else:
        rg, assoc, ns = o.ResourceGrid(14, 72, 15e3, num_tx=2, num_streams_per_tx=2, num_guard_carriers=[3, 4], **kw), [[1, 0], [0, 1]], 2
    sm = o.StreamManagement(assoc, ns)
    rng = np.random.default_rng(4)
    B, nrx, ntx, nra = 3, len(assoc), len(assoc[0]), 4
    feff = rg.num_effective_subcarriers
    y = (rng.normal(size=(B, nrx, nra, 14, rg.fft_size)) + 1j * rng.normal(size=(B, nrx, nra, 14, rg.fft_size))).astype(np.complex64)
    h = ((rng.normal(size=(B, nrx, nra, ntx, ns, 14, feff)) + 1j * rng.normal(size=(B, nrx, nra, ntx, ns, 14, feff))) / np.sqrt(2)).astype(np.complex64)
    ev = rng.uniform(0, 0.05, (1, 1, 1, ntx, ns, 14, feff)).astype(np.float32)
    no = rng.uniform(0.02, 0.1, (B,)).astype(np.float32)
    for whiten in (True, False):
        x, ne = f32.ofdm_equalize(rg, sm, y, h, ev, no, "lmmse", whiten)
        xr, nr = o.ofdm_lmmse_equalize(rg, sm, y, h, ev, no, whiten)
        assert x.shape == xr.shape and ne.shape == nr.shape
        assert np.allclose(x, xr, rtol=3e-4, atol=3e-5) and np.allclose(ne, nr, rtol=3e-4)
    for kind in ("zf", "mf"):
        x, ne = f32.ofdm_equalize(rg, sm, y, h, ev, no, kind)
        xr, nr = o.ofdm_linear_equalize(rg, sm, y, h, ev, no, kind)
        assert np.allclose(x, xr, rtol=1e-3, atol=1e-3 * np.abs(xr).max()) and np.allclose(ne, nr, rtol=1e-3)
