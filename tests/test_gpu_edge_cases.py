"""Edge cases of the blocks on the hot path, as the reference's Keras layers take them: empty batches (TensorFlow ops
accept tensors with a zero dimension), batches with several leading dimensions, one-element batches and the largest
5G code sizes.  Results are compared with the oracle where there is something to compare."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ldpc5g as ol, ldpc_bp as obp, mapping as om, polar as op, polar_c as pc


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


def _dev(a, dtype=torch.float32):
    from sionna_amd import _ffi
    return torch.as_tensor(np.asarray(a), dtype=dtype, device=_ffi.device())


# ------------------------------------------------------------------ empty batches
@pytest.mark.parametrize("lead", [(0,), (0, 3), (2, 0)])
def test_empty_batch_through_the_awgn_chain(phy, lead):
    k, n, m = 64, 128, 2
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, num_iter=4, cn_update="minsum")
    mapper = phy.mapping.Mapper("qam", m)
    demapper = phy.mapping.Demapper("app", "qam", m)
    ch = phy.channel.AWGN()
    u = _dev(np.zeros(lead + (k,), np.float32))
    c = enc(u)
    assert tuple(c.shape) == lead + (n,)
    x = mapper(c)
    assert tuple(x.shape) == lead + (n // m,) and x.dtype == torch.complex64
    y = ch(x, 0.5)
    assert tuple(y.shape) == tuple(x.shape)
    llr = demapper(y, 0.5)
    assert tuple(llr.shape) == lead + (n,)
    u_hat = dec(llr)
    assert tuple(u_hat.shape) == lead + (k,)


def test_empty_batch_polar_crc_scrambler(phy):
    enc = phy.fec.polar.Polar5GEncoder(32, 64)
    dec = phy.fec.polar.Polar5GDecoder(enc, "SCL", list_size=8, return_crc_status=True)
    u = _dev(np.zeros((0, 32), np.float32))
    c = enc(u)
    assert tuple(c.shape) == (0, 64)
    u_hat, status = dec(_dev(np.zeros((0, 64), np.float32)))
    assert tuple(u_hat.shape) == (0, 32) and status.shape[0] == 0
    crc = phy.fec.crc.CRCEncoder("CRC11")
    x = crc(_dev(np.zeros((0, 5, 40), np.float32)))
    assert tuple(x.shape) == (0, 5, 51)
    b, ok = phy.fec.crc.CRCDecoder(crc)(x)
    assert tuple(b.shape) == (0, 5, 40) and tuple(ok.shape)[:2] == (0, 5)
    scr = phy.fec.scrambling.Scrambler(seed=3)
    assert tuple(scr(_dev(np.zeros((0, 100), np.float32))).shape) == (0, 100)


def test_error_counters_on_empty_input(phy):
    a = _dev(np.zeros((0, 10), np.float32))
    assert float(phy.utils.count_errors(a, a)) == 0
    assert float(phy.utils.count_block_errors(a, a)) == 0


# ------------------------------------------------------------------ leading dimensions, single elements
@pytest.mark.parametrize("lead", [(1,), (2, 3, 5), (7, 1, 1, 2)])
def test_leading_dimensions_ldpc_chain_vs_oracle(phy, lead):
    k, n, m = 120, 300, 4
    rng = np.random.default_rng(len(lead) + 11)
    code = ol.LDPC5GCode(k, n, num_bits_per_symbol=m)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, num_iter=6, cn_update="minsum", hard_out=False)
    u = rng.integers(0, 2, lead + (k,)).astype(np.float32)
    c = _np(enc(_dev(u)))
    ref_c = code.encode(u.reshape(-1, k)).reshape(lead + (n,))
    assert np.array_equal(c, ref_c)
    y = (2 * c - 1) + 0.8 * rng.normal(size=c.shape).astype(np.float32)
    llr = (2 * y / 0.64).astype(np.float32)
    out = _np(dec(_dev(llr)))
    assert out.shape == lead + (k,)
    ref = obp.LDPC5GDecoder(code, num_iter=6, cn_update="minsum", hard_out=False).decode5g(llr.reshape(-1, n))
    assert np.array_equal(out.reshape(-1, k), ref.astype(np.float32))


@pytest.mark.parametrize("lead", [(1,), (3, 2), (2, 1, 4)])
def test_leading_dimensions_polar_chain_vs_oracle(phy, lead):
    k, n = 40, 128
    rng = np.random.default_rng(len(lead) + 5)
    enc = phy.fec.polar.Polar5GEncoder(k, n)
    dec = phy.fec.polar.Polar5GDecoder(enc, "SCL", list_size=8, return_crc_status=True)
    u = rng.integers(0, 2, lead + (k,)).astype(np.float32)
    c = _np(enc(_dev(u)))
    ocode = op.Polar5GCode(k, n)
    assert np.array_equal(c, ocode.encode(u.reshape(-1, k)).reshape(lead + (n,)))
    y = (2 * c - 1) + 0.7 * rng.normal(size=c.shape)
    llr = (2 * y / 0.49).astype(np.float32)
    u_hat, status = dec(_dev(llr))
    assert tuple(u_hat.shape) == lead + (k,) and tuple(status.shape) == lead
    ref, ref_status = pc.polar5g_decode(ocode, llr.reshape(-1, n), list_size=8, precision="f32", return_crc_status=True)
    assert np.array_equal(_np(u_hat).reshape(-1, k), ref)
    assert np.array_equal(_np(status).reshape(-1).astype(bool), ref_status.astype(bool))


def test_demapper_leading_dimensions_vs_oracle(phy):
    rng = np.random.default_rng(2)
    m = 6
    demapper = phy.mapping.Demapper("app", "qam", m)
    y = (rng.normal(size=(2, 3, 1, 50)) + 1j * rng.normal(size=(2, 3, 1, 50))).astype(np.complex64)
    no = rng.uniform(0.05, 1.0, size=(2, 3, 1, 50)).astype(np.float32)
    got = _np(demapper(_dev(y, torch.complex64), _dev(no)))
    ref = om.demapper(y, no, om.qam(m), method="app")
    assert got.shape == (2, 3, 1, 50 * m)
    assert np.allclose(got, ref.reshape(got.shape), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ largest code sizes
def test_largest_5g_ldpc_code_bit_exact(phy):
    """BG1, Z = 384: k = 8448, n = 25344 (rate 1/3) - the largest lifting size of 38.212."""
    k, n = 8448, 25344
    rng = np.random.default_rng(77)
    code = ol.LDPC5GCode(k, n)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, num_iter=3, cn_update="minsum", hard_out=False)
    u = rng.integers(0, 2, (2, k)).astype(np.float32)
    c = _np(enc(_dev(u)))
    assert np.array_equal(c, code.encode(u))
    y = (2 * c - 1) + 0.9 * rng.normal(size=c.shape)
    llr = (2 * y / 0.81).astype(np.float32)
    out = _np(dec(_dev(llr)))
    ref = obp.LDPC5GDecoder(code, num_iter=3, cn_update="minsum", hard_out=False).decode5g(llr)
    assert np.array_equal(out, ref.astype(np.float32))


@pytest.mark.parametrize("k,n", [(12, 1024), (1013, 1088), (1, 32)])
def test_polar_extreme_rates_bit_exact(phy, k, n):
    """Almost no / almost only information bits: schedules that are nearly all rate-0 nodes or all forks."""
    if k == 1:
        frozen, info = phy.fec.polar.generate_5g_ranking(k, n)
        dec = phy.fec.polar.PolarSCLDecoder(frozen, n, list_size=8)
        rng = np.random.default_rng(4)
        logits = rng.normal(size=(64, n)).astype(np.float32) * 3
        ref, _ = pc.SCLDecoder(frozen, n, 8, None, True).decode(logits)
        assert np.array_equal(_np(dec(_dev(logits))), ref)
        return
    rng = np.random.default_rng(k)
    enc = phy.fec.polar.Polar5GEncoder(k, n)
    dec = phy.fec.polar.Polar5GDecoder(enc, "SCL", list_size=8, return_crc_status=True)
    ocode = op.Polar5GCode(k, n)
    u = rng.integers(0, 2, (96, k)).astype(np.float32)
    c = _np(enc(_dev(u)))
    assert np.array_equal(c, ocode.encode(u))
    sigma = 0.5 if k > 500 else 1.2
    y = (2 * c - 1) + sigma * rng.normal(size=c.shape)
    llr = (2 * y / sigma ** 2).astype(np.float32)
    u_hat, status = dec(_dev(llr))
    ref, ref_status = pc.polar5g_decode(ocode, llr, list_size=8, precision="f32", return_crc_status=True)
    assert np.array_equal(_np(u_hat), ref)
    assert np.array_equal(_np(status).astype(bool), ref_status.astype(bool))
