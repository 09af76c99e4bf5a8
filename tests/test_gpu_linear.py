"""GPU tests of the generic linear encoder and the Gaussian prior source: bit-exact against the
GF(2) matrix product, codewords satisfy the parity checks, encode -> BP decode round trips on the
built-in example codes (reference test/unit/fec/test_linear_encoding.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("pcm_id", [0, 1, 2, 3, 4])
def test_linear_encoder_pcm(phy, pcm_id):
    u = phy.fec.utils
    pcm, k, n, _ = u.load_parity_check_examples(pcm_id)
    enc = phy.fec.linear.LinearEncoder(pcm, is_pcm=True)
    assert (enc.k, enc.n) == (k, n) and u.verify_gm_pcm(enc.gm, pcm)
    rng = np.random.default_rng(pcm_id)
    bits = rng.integers(0, 2, (3, 17, k)).astype(np.float32)
    c = _np(enc(bits))
    assert c.shape == (3, 17, n)
    assert np.array_equal(c, np.mod(bits @ enc.gm, 2))                     # c = u G over GF(2)
    assert not np.any(np.mod(c @ pcm.T, 2))                                  # H c^T = 0
    assert np.array_equal(_np(enc(bits[0, 0])), c[0, 0])                     # rank-1 input
    # noiseless and mildly noisy round trip through the BP decoder
    dec = phy.fec.ldpc.LDPCBPDecoder(pcm, num_iter=20, cn_update="minsum")
    c2 = c.reshape(-1, n)
    assert np.array_equal(_np(dec(8.0 * (2 * c2 - 1))), c2)
    with pytest.raises(ValueError):
        enc(np.zeros((2, k + 1), np.float32))


def test_linear_encoder_generator_matrix_and_errors(phy):
    rng = np.random.default_rng(9)
    for k, n in ((1, 1), (5, 31), (32, 33), (33, 64), (100, 257), (700, 1000)):
        gm = rng.integers(0, 2, (k, n))
        enc = phy.fec.linear.LinearEncoder(gm)
        bits = rng.integers(0, 2, (40, k)).astype(np.float32)
        assert np.array_equal(_np(enc(bits)), np.mod(bits @ gm, 2)), (k, n)
    with pytest.raises(ValueError):
        phy.fec.linear.LinearEncoder(np.array([[1, 2], [0, 1]]))
    with pytest.raises(ValueError):
        phy.fec.linear.LinearEncoder(np.ones((3, 2)))
    z = phy.fec.linear.AllZeroEncoder(4, 9)
    assert float(z(np.ones((2, 3, 4), np.float32)).abs().max()) == 0 and tuple(z(np.ones((2, 3, 4), np.float32)).shape) == (2, 3, 9)


def test_gaussian_prior_source(phy):
    u = phy.fec.utils
    src = u.GaussianPriorSource()
    llr = _np(src([200, 1000], no=0.5))
    sigma2 = 4 / 0.5
    assert abs(llr.mean() + sigma2 / 2) < 0.05 and abs(llr.var() - sigma2) < 0.15
    for mi in (0.2, 0.5, 0.9):
        llr = _np(src([400, 1000], mi=mi))
        assert abs(u.llr2mi(-llr) - mi) < 0.02 or abs(u.llr2mi(llr) - mi) < 0.02
    with pytest.raises(ValueError):
        src([2, 2])


def test_linear_encoder_on_dense_polar_matrices(phy):
    """LinearEncoder(gm) with gm from generate_dense_polar is the Polar encoder (reference test_linear_encoding.py:115-143), and the
    dense parity-check matrix checks its codewords"""
    from sionna_amd.phy.fec.polar import generate_dense_polar, generate_5g_ranking
    for k, n in ((32, 64), (100, 256), (300, 512)):
        f, _ = generate_5g_ranking(k, n)
        pcm, gm = generate_dense_polar(f, n, verbose=False)
        u = phy.mapping.BinarySource()([50, k])
        c = _np(phy.fec.linear.LinearEncoder(gm)(u))
        assert np.array_equal(c, _np(phy.fec.polar.PolarEncoder(f, n)(u)))
        assert not np.any((c.astype(int) @ pcm.astype(int).T) % 2)
