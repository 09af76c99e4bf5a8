"""GPU parity of the 5G NR transport-block chain against the reference's golden vectors
(tests/golden/tb_golden.npz <- test/unit/nr/tb_refs) and oracle/nr_tb.py; bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nr_tb
from test_oracle_nr_tb import META, tb_case


@pytest.fixture(scope="module")
def nr():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p.nr


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("i", range(len(META)))
def test_tb_encoder_decoder_reference_vectors(nr, i):
    """Reference test_tb_encoder.py:20-63: encoder output == c_ref, decoder(2c-1) == u_ref."""
    u, c, c_ns, kw = tb_case(i)
    enc = nr.TBEncoder(channel_type="PUSCH", codeword_index=0, use_scrambler=True, verbose=False, **kw)
    got = _np(enc(u))
    assert np.array_equal(got, c)
    assert np.array_equal(_np(nr.TBEncoder(use_scrambler=False, **kw)(u)), c_ns)
    dec = nr.TBDecoder(enc, cn_update="minsum")           # min-sum does not need scaled LLRs
    u_hat, ok = dec(2 * c - 1)
    assert np.array_equal(_np(u_hat), u) and bool(ok.all())
    o = nr_tb.TBEncoder(**kw)
    assert (enc.tb_size, enc.num_cbs, enc.k_padding, enc.n) == (o.tb_size, o.num_cbs, o.k_padding, o.n)
    assert np.array_equal(enc.cw_lengths, o.cw_lengths) and np.array_equal(enc.output_perm_inv, o.output_perm_inv)


def test_tb_multi_stream(nr):
    """Reference test_tb_encoder.py:65-118: lists of n_rnti / n_id = independent streams on axis -2."""
    n_rnti = [224, 42, 1, 1337, 45666, 2333, 2133]
    n_id = [42, 123, 0, 3, 32, 456, 875]
    kw = dict(target_tb_size=50000, num_coded_bits=100000, target_coderate=0.5, num_bits_per_symbol=4, num_layers=2)
    enc = nr.TBEncoder(n_rnti=n_rnti, n_id=n_id, **kw)
    u = np.random.default_rng(0).integers(0, 2, (3, len(n_rnti), enc.k)).astype(np.float32)
    c = _np(enc(u))
    u_hat, ok = nr.TBDecoder(enc)(2 * c - 1)
    assert np.array_equal(_np(u_hat), u) and bool(ok.all()) and tuple(ok.shape) == (3, len(n_rnti))
    for idx, (nr_, ni) in enumerate(zip(n_rnti, n_id)):
        e1 = nr.TBEncoder(n_rnti=nr_, n_id=ni, **kw)
        assert np.array_equal(_np(e1(u[:, idx, :])), c[:, idx, :])
    assert np.array_equal(c[:, :2], nr_tb.TBEncoder(n_rnti=n_rnti[:2], n_id=n_id[:2], **kw).encode(u[:, :2]))


def test_tb_padding_noise_and_crc_status(nr):
    kw = dict(target_tb_size=1000, num_coded_bits=2400, target_coderate=1000 / 2400, num_bits_per_symbol=4)
    enc = nr.TBEncoder(**kw)
    o = nr_tb.TBEncoder(**kw)
    assert enc.k_padding == o.k_padding > 0 and enc.k == 1000
    rng = np.random.default_rng(1)
    u = rng.integers(0, 2, (64, 1000)).astype(np.float32)
    c = _np(enc(u))
    assert np.array_equal(c, o.encode(u))
    llr = ((2 * c - 1) + 0.7 * rng.normal(size=c.shape)).astype(np.float32) * 4
    dec = nr.TBDecoder(enc, num_bp_iter=10, cn_update="minsum")
    u_hat, ok = dec(llr)
    u_ref, ok_ref = nr_tb.TBDecoder(o, num_bp_iter=10, cn_update="minsum").decode(llr)
    assert np.array_equal(_np(u_hat), u_ref) and np.array_equal(_np(ok), ok_ref)
    assert 0 < np.mean(_np(ok)) <= 1                            # some blocks decode at this SNR
    wrong = np.any(_np(u_hat) != u, axis=1)
    assert np.all(_np(ok)[wrong] == 0)                          # every wrong TB is flagged by the CRC
    with pytest.raises(AssertionError):
        enc(np.zeros((2, 999), np.float32))
    with pytest.raises(AssertionError):
        nr.TBEncoder(target_tb_size=1000, num_coded_bits=2400, target_coderate=0.95, num_bits_per_symbol=4)
