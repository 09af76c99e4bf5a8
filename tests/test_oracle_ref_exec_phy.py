"""Pins oracle/mapping.py, oracle/mimo_f32.py (+ the complex128 witness in oracle/ofdm.py) and oracle/utils.py to outputs of
the reference's OWN mapping.py / mimo/equalization.py / mimo/utils.py / utils/linalg.py / utils/misc.py, executed
unmodified under the NumPy stand-in for TensorFlow (tools/gen_phy_ref_golden.py -> tests/golden/phy_ref_golden.npz).
Exact pieces bit for bit; float pipelines at the north star's 1e-5 relative bar (scaled by the problem's conditioning
for the single-precision linear algebra: float32 results are only defined up to cond(S) 2^-24 on either side)."""
import os

import numpy as np
import pytest

from oracle import mapping as om, mimo_f32, ofdm as oofdm, utils as outil

GOLD = os.path.join(os.path.dirname(__file__), "golden", "phy_ref_golden.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


@pytest.mark.parametrize("m", (2, 4, 6, 8))
def test_constellation_and_mapper_bit_exact(g, m):
    assert np.array_equal(om.qam(m), g[f"qam{m}_points"])
    assert np.array_equal(om.mapper(g[f"qam{m}_bits"], om.qam(m)), g[f"qam{m}_x"])


@pytest.mark.parametrize("m", (2, 4, 6, 8))
@pytest.mark.parametrize("meth", ("app", "maxlog"))
def test_demapper_llrs_within_1e5(g, m, meth):
    pts = om.qam(m)
    for no in (0.5, 0.05):
        got = om.demapper(g[f"qam{m}_y_no{no}"], np.float32(no), pts, meth)
        ref = g[f"qam{m}_{meth}_no{no}"]
        assert got.dtype == ref.dtype == np.float32
        # LLR = difference of two (log-sum-)max terms of size |y-c|^2/no: the 1e-5 bar is relative to those terms
        scale = np.maximum(np.abs(ref), np.abs(g[f"qam{m}_y_no{no}"]).max() ** 2 / no)
        assert np.all(np.abs(got - ref) <= 1e-5 * scale.max()), np.abs(got - ref).max()
        assert np.mean(np.isclose(got, ref, rtol=1e-5, atol=1e-5)) >= 0.99
    got = om.demapper(g[f"qam{m}_y_t"], g[f"qam{m}_no_t"], pts, meth, prior=g[f"qam{m}_prior"])
    ref = g[f"qam{m}_{meth}_prior"]
    assert np.allclose(got, ref, rtol=1e-5, atol=2e-4), np.abs(got - ref).max()


@pytest.mark.parametrize("m", (2, 4, 6, 8))
def test_demapper_hard_decisions(g, m):
    for no in (0.5, 0.05):
        got = om.demapper(g[f"qam{m}_y_no{no}"], np.float32(no), om.qam(m), "app", hard_out=True)
        ref = g[f"qam{m}_hard_no{no}"]
        soft = g[f"qam{m}_app_no{no}"]
        sure = np.abs(soft) > 1e-3                                  # a sign can only differ where the LLR is ~0
        assert np.array_equal(got.astype(np.uint8)[sure], ref[sure])


def test_custom_constellation(g):
    pts = g["custom3_in"]
    pts = pts - pts.mean()
    pts = (pts / np.sqrt(np.mean(np.abs(pts) ** 2))).astype(np.complex64)
    assert np.allclose(pts, g["custom3_points"], rtol=1e-6, atol=1e-7)
    # the reference's Demapper measures against the RAW stored points (mapping.py:667-668), not against constellation()
    got = om.demapper(g["custom3_y"], np.float32(0.3), g["custom3_in"], "app")
    assert np.allclose(got, g["custom3_app"], rtol=1e-5, atol=1e-4)
    wrong = om.demapper(g["custom3_y"], np.float32(0.3), g["custom3_points"], "app")
    assert not np.allclose(wrong, g["custom3_app"], rtol=1e-2, atol=1e-2)


SHAPES = ((4, 2), (2, 1), (1, 1), (8, 4), (16, 4), (4, 4))


def _cond(s, h):
    c = np.linalg.cond(s.astype(np.complex128))
    hw = np.linalg.solve(np.linalg.cholesky(s.astype(np.complex128)), h.astype(np.complex128))
    return c * np.linalg.cond(hw)


@pytest.mark.parametrize("mk", SHAPES, ids=[f"{m}x{k}" for m, k in SHAPES])
@pytest.mark.parametrize("noise", ("col", "wht"))
def test_lmmse_zf_mf_equalizers(g, mk, noise):
    """float32 oracle (the kernels' bit-level spec) and complex128 witness against the reference-executed complex64
    results: within 1e-5 x conditioning, and the witness sits between the two float32 results at that scale."""
    M, K = mk
    p = f"mimo{M}x{K}_{noise}_"
    y, h, s = g[p + "y"], g[p + "h"], g[p + "s"]
    cond = _cond(s, h)                                              # [B]
    for wi in (1, 0):
        xr, nr = g[p + f"lmmse_w{wi}_x"], g[p + f"lmmse_w{wi}_no"]
        xo, no_ = mimo_f32.lmmse_equalizer(y, h, s, whiten_interference=bool(wi))
        xw, nw = oofdm.lmmse_equalizer(y, h, s, whiten_interference=bool(wi))
        tol = 1e-5 * np.maximum(cond, 10.)[:, None] if wi else 1e-5 * np.maximum(np.linalg.cond((h @ np.conj(np.swapaxes(h, -1, -2)) + s).astype(np.complex128)), 10.)[:, None]
        sx = np.maximum(np.abs(xw), 1.0)
        assert np.all(np.abs(xr - xw) <= tol * sx), (wi, np.max(np.abs(xr - xw) / (tol * sx)))
        assert np.all(np.abs(xo - xw) <= tol * sx), (wi, np.max(np.abs(xo - xw) / (tol * sx)))
        sn = np.maximum(np.abs(nw), 1e-2)
        assert np.all(np.abs(nr - nw) <= tol * sn) and np.all(np.abs(no_ - nw) <= tol * sn), wi
    for kind, fo, fw in (("zf", mimo_f32.zf_equalizer, oofdm.zf_equalizer), ("mf", mimo_f32.mf_equalizer, oofdm.mf_equalizer)):
        xr, nr = g[p + f"{kind}_x"], g[p + f"{kind}_no"]
        xo, no_ = fo(y, h, s)
        xw, nw = fw(y, h, s)
        hc = np.linalg.cond(h.astype(np.complex128)) ** 2
        tol = 1e-5 * np.maximum(hc, 10.)[:, None]
        assert np.all(np.abs(xr - xw) <= tol * np.maximum(np.abs(xw), 1.0)), kind
        assert np.all(np.abs(xo - xw) <= tol * np.maximum(np.abs(xw), 1.0)), kind
        assert np.all(np.abs(nr - nw) <= tol * np.maximum(np.abs(nw), 1e-2)), kind
        assert np.all(np.abs(no_ - nw) <= tol * np.maximum(np.abs(nw), 1e-2)), kind


def test_inv_cholesky_and_pinv(g):
    for noise in ("col", "wht"):
        p = f"mimo4x2_{noise}_"
        s, h = g[p + "s"], g[p + "h"]
        li = np.linalg.inv(np.linalg.cholesky(s.astype(np.complex128)))
        assert np.allclose(g[p + "inv_chol"], li, rtol=1e-4, atol=1e-5)
        assert np.allclose(g[p + "pinv"], np.linalg.pinv(h.astype(np.complex128)), rtol=1e-4, atol=1e-5)


def test_ebnodb2no_and_hard_decisions_bit_exact(g):
    for (e, m, r), no in zip(g["ebno_grid"], g["ebno_no"]):
        got = outil.ebnodb2no(e, int(m), r)
        assert np.float32(got) == no or abs(np.float32(got) - no) <= np.spacing(no), (e, m, r, got, no)
    assert np.array_equal(outil.hard_decisions(g["hard_in"]), g["hard_out"])


@pytest.mark.parametrize("m", [1, 2, 4, 6])
def test_symbol_logits2llrs_matches_reference_execution(g, m):
    """SymbolLogits2LLRs as a block (mapping.py:794-967): logits on the points -> LLRs, app / maxlog, no prior, a prior per
    row, one prior vector for all rows, hard decisions."""
    z, pr, pv = g[f"l2l{m}_z"], g[f"l2l{m}_prior"], g[f"l2l{m}_prior_vec"]
    for meth in ("app", "maxlog"):
        for key, prior in (("", None), ("_prior", pr), ("_prior_vec", pv)):
            ref = g[f"l2l{m}_{meth}{key}"]
            got = om.symbol_logits2llrs(z, m, meth, prior)
            assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (meth, key)
    assert np.array_equal(om.symbol_logits2llrs(z, m, "app", pr, hard_out=True).astype(np.uint8), g[f"l2l{m}_hard"])
