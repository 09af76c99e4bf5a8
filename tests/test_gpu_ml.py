"""GPU parity of the maximum-likelihood detectors (csrc/mimo.hip ml_items_kernel / ofdm_ml_kernel behind
phy.mimo.MaximumLikelihoodDetector and phy.ofdm.MaximumLikelihoodDetector(.WithPrior)) against the reference's own detector
EXECUTED under the NumPy stand-in for TensorFlow (tests/golden/ml_ref_golden.npz, tools/gen_ml_ref_golden.py) and against the
float64 oracle (oracle/ofdm.py::ml_detector, pinned to the same fixture by tests/test_oracle_ref_exec_ml.py).  Float32 exponents
of magnitude up to ~7e2: soft values within 2e-3 absolute + 2e-4 relative (the tolerance the fixture needs against float64)."""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ofdm as o, mapping as omap

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ml_ref_golden.npz"))
CASES = ast.literal_eval(str(G["cases"]))


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


def _cplx(rng, shape, scale=1.0):
    return ((rng.normal(size=shape) + 1j * rng.normal(size=shape)) * scale / np.sqrt(2)).astype(np.complex64)


def _check(got, ref, soft_ref, output, hard):
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if not hard:
        assert np.allclose(got, ref, rtol=2e-4, atol=2e-3), float(np.max(np.abs(got - ref)))
        return
    if output == "bit":
        sure = np.abs(soft_ref) > 1e-2
    else:
        top2 = np.sort(soft_ref, -1)[..., -2:]
        sure = (top2[..., 1] - top2[..., 0]) > 1e-2
    assert sure.mean() > 0.95 and np.array_equal(got[sure], ref[sure])


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_ml_detector_matches_reference_execution(phy, ci):
    M, K, nb, output, method, hard, with_prior = CASES[ci]
    y, h, s, ref = G[f"c{ci}_y"], G[f"c{ci}_h"], G[f"c{ci}_s"], G[f"c{ci}_out"]
    prior = G[f"c{ci}_prior"] if with_prior else None
    det = phy.mimo.MaximumLikelihoodDetector(output, method, K, "qam", nb, hard_out=hard)
    got = _np(det(y, h, s, prior) if with_prior else det(y, h, s))
    if output == "symbol" and hard:
        assert got.dtype == np.int32
    soft = o.ml_detector(y, h, s, G[f"c{ci}_points"], method, prior, output, False)
    _check(got, ref, soft, output, hard)


@pytest.mark.parametrize("m,k,nb,method", [(4, 2, 2, "app"), (4, 2, 4, "maxlog"), (8, 4, 2, "app"), (2, 2, 4, "app"), (1, 1, 6, "maxlog"),
                                           (16, 4, 2, "maxlog"), (4, 1, 4, "app")])
def test_ml_detector_vs_oracle(phy, m, k, nb, method):
    rng = np.random.default_rng(m * 7 + k + nb)
    n = 300
    pts = omap.qam(nb)
    h = _cplx(rng, (n, m, k))
    x = pts[rng.integers(0, 1 << nb, (n, k))]
    a = _cplx(rng, (n, m, m), 0.3)
    s = (a @ np.conj(np.swapaxes(a, -1, -2)) + 0.1 * np.eye(m)).astype(np.complex64)
    y = (np.einsum("nmk,nk->nm", h, x) + _cplx(rng, (n, m), 0.3)).astype(np.complex64)
    prior_llr = (rng.normal(size=(n, k, nb)) * 2).astype(np.float32)
    for output, prior in (("bit", None), ("bit", prior_llr), ("symbol", None), ("symbol", (rng.normal(size=(n, k, 1 << nb))).astype(np.float32))):
        det = phy.mimo.MaximumLikelihoodDetector(output, method, k, "qam", nb)
        got = _np(det(y, h, s, prior) if prior is not None else det(y, h, s))
        ref = o.ml_detector(y, h, s, pts, method, prior, output, False)
        _check(got, ref, ref, output, False)
    # leading dimensions and broadcasting of y / s like the other detectors
    det = phy.mimo.MaximumLikelihoodDetector("bit", method, k, "qam", nb, hard_out=True)
    got = _np(det(y.reshape(3, 100, m), h.reshape(3, 100, m, k), s.reshape(3, 100, m, m)))
    soft = o.ml_detector(y, h, s, pts, method, None, "bit", False).reshape(3, 100, k, nb)
    _check(got, (soft > 0).astype(np.float32), soft, "bit", True)


def test_ml_detector_limits(phy):
    with pytest.raises(NotImplementedError):
        phy.mimo.MaximumLikelihoodDetector("bit", "app", 4, "qam", 6)              # 64^4 candidate vectors


def _grids(phy, **kw):
    base = dict(num_tx=1, num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[3, 4], dc_null=True,
                pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    base.update(kw)
    obase = {k_: v for k_, v in base.items() if k_ != "precision"}                # (the oracle's grid is index bookkeeping only)
    return phy.ofdm.ResourceGrid(14, 72, 15e3, **base), o.ResourceGrid(14, 72, 15e3, **obase)


@pytest.mark.parametrize("output,method,hard", [("bit", "app", False), ("bit", "maxlog", True), ("symbol", "app", False), ("symbol", "maxlog", True)])
def test_ofdm_ml_detector_vs_oracle(phy, output, method, hard):
    rg, org = _grids(phy)
    sm, osm = phy.mimo.StreamManagement(np.array([[1]]), 2), o.StreamManagement(np.array([[1]]), 2)
    rng = np.random.default_rng(8)
    B, nb = 3, 2
    pts = omap.qam(nb)
    y = _cplx(rng, (B, 1, 4, 14, 72))
    h_hat = _cplx(rng, (B, 1, 4, 1, 2, 14, rg.num_effective_subcarriers))
    ev = rng.uniform(0.0, 0.05, size=(1, 1, 1, 1, 2, 14, rg.num_effective_subcarriers)).astype(np.float32)
    det = phy.ofdm.MaximumLikelihoodDetector(output, method, rg, sm, constellation_type="qam", num_bits_per_symbol=nb, hard_out=hard)
    got = _np(det(y, h_hat, ev, 0.4))
    ref = o.ofdm_ml_detector(org, osm, y, h_hat, ev, 0.4, pts, method, None, output, hard)
    soft = o.ofdm_ml_detector(org, osm, y, h_hat, ev, 0.4, pts, method, None, output, False)
    _check(got, ref.astype(got.dtype) if hard else ref, soft, output, hard)
    # with prior
    nd = rg.num_data_symbols
    prior = (rng.normal(size=(B, 1, 2, nd * nb)) * 2).astype(np.float32) if output == "bit" else rng.normal(size=(B, 1, 2, nd, 1 << nb)).astype(np.float32)
    detp = phy.ofdm.MaximumLikelihoodDetectorWithPrior(output, method, rg, sm, constellation_type="qam", num_bits_per_symbol=nb, hard_out=hard)
    got = _np(detp(y, h_hat, prior, ev, 0.4))
    ref = o.ofdm_ml_detector(org, osm, y, h_hat, ev, 0.4, pts, method, prior, output, hard)
    soft = o.ofdm_ml_detector(org, osm, y, h_hat, ev, 0.4, pts, method, prior, output, False)
    _check(got, ref.astype(got.dtype) if hard else ref, soft, output, hard)


def test_ofdm_ml_two_receivers_with_interference(phy):
    """two receivers, one stream each, the other transmitter's stream is interference (undesired stream in the covariance)"""
    rg, org = _grids(phy, num_tx=2, num_streams_per_tx=1)
    assoc = np.array([[1, 0], [0, 1]])
    sm, osm = phy.mimo.StreamManagement(assoc, 1), o.StreamManagement(assoc, 1)
    rng = np.random.default_rng(9)
    B, nb = 2, 4
    pts = omap.qam(nb)
    y = _cplx(rng, (B, 2, 4, 14, 72))
    h_hat = _cplx(rng, (B, 2, 4, 2, 1, 14, rg.num_effective_subcarriers))
    det = phy.ofdm.MaximumLikelihoodDetector("bit", "app", rg, sm, constellation_type="qam", num_bits_per_symbol=nb)
    got = _np(det(y, h_hat, 0.0, 0.3))
    ref = o.ofdm_ml_detector(org, osm, y, h_hat, np.zeros(1, np.float32), 0.3, pts, "app")
    _check(got, ref, ref, "bit", False)
    # with prior: every receiver's stream takes the prior of the stream it detects
    prior = rng.normal(size=(B, 2, 1, rg.num_data_symbols * nb)).astype(np.float32)
    detp = phy.ofdm.MaximumLikelihoodDetectorWithPrior("bit", "app", rg, sm, constellation_type="qam", num_bits_per_symbol=nb)
    refp = o.ofdm_ml_detector(org, osm, y, h_hat, np.zeros(1, np.float32), 0.3, pts, "app", prior)
    _check(_np(detp(y, h_hat, prior, 0.0, 0.3)), refp, refp, "bit", False)


# ---------------------------------------------------------------------------------------------------------------------
# KBestDetector(use_real_rep=True): samd_kbest_real_f32 / samd_ofdm_kbest_real_f32 against the reference's detector executed
# (the same fixture file) and the float64 oracle (oracle/ofdm.py::kbest_detector_real)
KB = ast.literal_eval(str(G["kbest_real_cases"]))


@pytest.mark.parametrize("ci", range(len(KB)))
def test_kbest_real_rep_matches_reference_execution(phy, ci):
    M, K, nb, kk, output, hard = KB[ci]
    y, h, s, ref = G[f"k{ci}_y"], G[f"k{ci}_h"], G[f"k{ci}_s"], G[f"k{ci}_out"]
    det = phy.mimo.KBestDetector(output, K, kk, "qam", nb, hard_out=hard, use_real_rep=True)
    got = _np(det(y, h, s))
    assert got.shape == ref.shape
    if hard:
        assert np.mean(got == ref) > 0.97
    else:
        assert np.mean(np.isclose(got, ref, rtol=1e-3, atol=5e-3)) > 0.97      # a different k-th path at a float32 near-tie moves single LLRs


@pytest.mark.parametrize("m,k,nb,paths", [(4, 2, 4, 16), (2, 2, 2, 16), (4, 4, 2, 32), (8, 2, 6, 24), (2, 1, 4, 4), (1, 1, 2, 4), (4, 1, 6, 8)])
def test_kbest_real_rep_vs_oracle(phy, m, k, nb, paths):
    rng = np.random.default_rng(m * 5 + k + nb)
    n = 300
    pts = omap.qam(nb)
    h = _cplx(rng, (n, m, k))
    x = pts[rng.integers(0, 1 << nb, (n, k))]
    a = _cplx(rng, (n, m, m), 0.3)
    s = (a @ np.conj(np.swapaxes(a, -1, -2)) + 0.1 * np.eye(m)).astype(np.complex64)
    y = (np.einsum("nmk,nk->nm", h, x) + _cplx(rng, (n, m), 0.3)).astype(np.complex64)
    got = _np(phy.mimo.KBestDetector("bit", k, paths, "qam", nb, use_real_rep=True)(y, h, s))
    ref = o.kbest_detector_real(y, h, s, nb, paths)
    ok = np.all(np.isclose(got, ref, rtol=1e-3, atol=2e-3), axis=(1, 2))
    assert ok.mean() > 0.97, (1 - ok.mean(), np.max(np.abs(got - ref)))
    hard = _np(phy.mimo.KBestDetector("bit", k, paths, "qam", nb, hard_out=True, use_real_rep=True)(y, h, s))
    assert np.mean(np.all(hard == o.kbest_detector_real(y, h, s, nb, paths, hard_out=True), axis=(1, 2))) > 0.99
    sym = _np(phy.mimo.KBestDetector("symbol", k, paths, "qam", nb, hard_out=True, use_real_rep=True)(y, h, s))
    assert sym.dtype == np.int32 and np.mean(sym == o.kbest_detector_real(y, h, s, nb, paths, hard_out=True, output="symbol")) > 0.99
    # both representations search the same lattice: with every path kept the hard decisions coincide (mimo/detection.py:546-549)
    if (1 << nb) ** k <= 64:
        full = (1 << nb) ** k
        a_ = _np(phy.mimo.KBestDetector("bit", k, full, "qam", nb, hard_out=True, use_real_rep=True)(y, h, s))
        b_ = _np(phy.mimo.KBestDetector("bit", k, full, "qam", nb, hard_out=True)(y, h, s))
        assert np.mean(np.all(a_ == b_, axis=(1, 2))) > 0.99
    with pytest.raises(AssertionError):
        phy.mimo.KBestDetector("bit", k, paths, "pam", 2, use_real_rep=True)


def test_ofdm_kbest_real_rep_vs_oracle(phy):
    rg, org = _grids(phy)
    sm, osm = phy.mimo.StreamManagement(np.array([[1]]), 2), o.StreamManagement(np.array([[1]]), 2)
    rng = np.random.default_rng(11)
    B, nb = 3, 4
    pts = omap.qam(nb)
    y = _cplx(rng, (B, 1, 4, 14, 72))
    h_hat = _cplx(rng, (B, 1, 4, 1, 2, 14, rg.num_effective_subcarriers))
    got = _np(phy.ofdm.KBestDetector("bit", 2, 16, rg, sm, constellation_type="qam", num_bits_per_symbol=nb, use_real_rep=True)(y, h_hat, 0.0, 0.4))
    ref = o.ofdm_kbest_detector(org, osm, y, h_hat, np.zeros(1, np.float32), 0.4, pts, 16, use_real_rep=True)
    assert got.shape == ref.shape
    assert np.mean(np.isclose(got, ref, rtol=1e-3, atol=2e-3)) > 0.99


# ---------------------------------------------------------------------------------------------------------------------
# precision="double" (the reference's own ML tests run in double: test/unit/mimo/test_mimo_ml_det.py:371-407): samd_ml_detect_f64
# (csrc/f64.hip) against the float64 oracle at 1e-9, and against the reference-executed fixture
def _c128(rng, shape, scale=1.0):
    return (rng.normal(size=shape) + 1j * rng.normal(size=shape)) * scale / np.sqrt(2)


def _check64(got, ref, soft_ref, output, hard):
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if not hard:
        assert got.dtype == np.float64
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-9), float(np.max(np.abs(got - ref)))
        return
    if output == "bit":
        sure = np.abs(soft_ref) > 1e-9
    else:
        top2 = np.sort(soft_ref, -1)[..., -2:]
        sure = (top2[..., 1] - top2[..., 0]) > 1e-9
    assert sure.mean() > 0.99 and np.array_equal(got[sure], ref[sure])


@pytest.mark.parametrize("m,k,nb,method", [(4, 2, 2, "app"), (4, 2, 4, "maxlog"), (8, 4, 2, "app"), (2, 2, 4, "app"), (1, 1, 6, "maxlog"),
                                           (16, 4, 2, "maxlog"), (4, 1, 4, "app"), (3, 3, 1, "app")])
def test_ml_detector_double_vs_oracle(phy, m, k, nb, method):
    rng = np.random.default_rng(m * 11 + k + nb)
    n = 150
    pts = omap.qam(nb, dtype=np.complex128) if nb > 1 else np.array([1.0, -1.0], np.complex128)      # BPSK = 1-bit PAM (mapping.py:15-42)
    ctype = "qam" if nb > 1 else "pam"
    h = _c128(rng, (n, m, k))
    x = pts[rng.integers(0, 1 << nb, (n, k))]
    a = _c128(rng, (n, m, m), 0.3)
    s = a @ np.conj(np.swapaxes(a, -1, -2)) + 0.1 * np.eye(m)
    y = np.einsum("nmk,nk->nm", h, x) + _c128(rng, (n, m), 0.3)
    prior_llr = rng.normal(size=(n, k, nb)) * 2
    for output, prior in (("bit", None), ("bit", prior_llr), ("symbol", None), ("symbol", rng.normal(size=(n, k, 1 << nb)))):
        for hard in (False, True):
            det = phy.mimo.MaximumLikelihoodDetector(output, method, k, ctype, nb, hard_out=hard, precision="double")
            got = _np(det(y, h, s, prior) if prior is not None else det(y, h, s))
            ref = o.ml_detector(y, h, s, pts, method, prior, output, hard)
            soft = o.ml_detector(y, h, s, pts, method, prior, output, False)
            _check64(got, ref.astype(got.dtype) if hard else ref, soft, output, hard)


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_ml_detector_double_matches_reference_execution(phy, ci):
    """the fixture holds float32 executions of the reference: the double kernel agrees to float32 accuracy"""
    M, K, nb, output, method, hard, with_prior = CASES[ci]
    y, h, s, ref = G[f"c{ci}_y"], G[f"c{ci}_h"], G[f"c{ci}_s"], G[f"c{ci}_out"]
    prior = G[f"c{ci}_prior"] if with_prior else None
    det = phy.mimo.MaximumLikelihoodDetector(output, method, K, "qam", nb, hard_out=hard, precision="double")
    got = _np(det(y, h, s, prior) if with_prior else det(y, h, s))
    soft = o.ml_detector(y, h, s, G[f"c{ci}_points"], method, prior, output, False)
    _check(got, ref, soft, output, hard)


@pytest.mark.parametrize("output,method,hard", [("bit", "app", False), ("bit", "maxlog", True), ("symbol", "app", False), ("symbol", "maxlog", True)])
def test_ofdm_ml_detector_double_vs_oracle(phy, output, method, hard):
    rg, org = _grids(phy, precision="double")
    sm, osm = phy.mimo.StreamManagement(np.array([[1]]), 2), o.StreamManagement(np.array([[1]]), 2)
    rng = np.random.default_rng(18)
    B, nb = 2, 2
    pts = omap.qam(nb, dtype=np.complex128)
    y = _c128(rng, (B, 1, 4, 14, 72))
    h_hat = _c128(rng, (B, 1, 4, 1, 2, 14, rg.num_effective_subcarriers))
    ev = rng.uniform(0.0, 0.05, size=(1, 1, 1, 1, 2, 14, rg.num_effective_subcarriers))
    kw = dict(constellation_type="qam", num_bits_per_symbol=nb, hard_out=hard, precision="double")
    got = _np(phy.ofdm.MaximumLikelihoodDetector(output, method, rg, sm, **kw)(y, h_hat, ev, 0.25))
    ref = o.ofdm_ml_detector(org, osm, y, h_hat, ev, 0.25, pts, method, None, output, hard)
    soft = o.ofdm_ml_detector(org, osm, y, h_hat, ev, 0.25, pts, method, None, output, False)
    _check64(got, ref.astype(got.dtype) if hard else ref, soft, output, hard)
    nd = rg.num_data_symbols
    prior = rng.normal(size=(B, 1, 2, nd * nb)) * 2 if output == "bit" else rng.normal(size=(B, 1, 2, nd, 1 << nb))
    got = _np(phy.ofdm.MaximumLikelihoodDetectorWithPrior(output, method, rg, sm, **kw)(y, h_hat, prior, ev, 0.25))
    ref = o.ofdm_ml_detector(org, osm, y, h_hat, ev, 0.25, pts, method, prior, output, hard)
    soft = o.ofdm_ml_detector(org, osm, y, h_hat, ev, 0.25, pts, method, prior, output, False)
    _check64(got, ref.astype(got.dtype) if hard else ref, soft, output, hard)


def test_ofdm_ml_double_two_receivers_with_interference(phy):
    rg, org = _grids(phy, num_tx=2, num_streams_per_tx=1, precision="double")
    assoc = np.array([[1, 0], [0, 1]])
    sm, osm = phy.mimo.StreamManagement(assoc, 1), o.StreamManagement(assoc, 1)
    rng = np.random.default_rng(19)
    B, nb = 2, 4
    pts = omap.qam(nb, dtype=np.complex128)
    y = _c128(rng, (B, 2, 4, 14, 72))
    h_hat = _c128(rng, (B, 2, 4, 2, 1, 14, rg.num_effective_subcarriers))
    det = phy.ofdm.MaximumLikelihoodDetectorWithPrior("bit", "app", rg, sm, constellation_type="qam", num_bits_per_symbol=nb, precision="double")
    prior = rng.normal(size=(B, 2, 1, rg.num_data_symbols * nb))
    got = _np(det(y, h_hat, prior, 0.0, 0.25))
    ref = o.ofdm_ml_detector(org, osm, y, h_hat, np.zeros(1), 0.25, pts, "app", prior)
    _check64(got, ref, ref, "bit", False)
