"""Pin oracle/nr_tb.py against the reference's transport-block vectors
(test/unit/nr/tb_refs/*.npz -> tests/golden/tb_golden.npz; reference test
test/unit/nr/test_tb_encoder.py:20-63)."""
import os

import numpy as np
import pytest

from oracle import nr_tb

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "tb_golden.npz"))
META = GOLD["meta"]


def tb_case(i):
    tb, ncb, n_id, n_rnti, rate, m, layers = (int(v) for v in META[i])
    u = np.unpackbits(GOLD[f"u_ref_{i}"], axis=1)[:, :tb].astype(np.float32)
    c = np.unpackbits(GOLD[f"c_ref_{i}"], axis=1)[:, :ncb].astype(np.float32)
    c_ns = np.unpackbits(GOLD[f"c_ref_no_scr_{i}"], axis=1)[:, :ncb].astype(np.float32)
    kw = dict(target_tb_size=tb, num_coded_bits=ncb, target_coderate=rate / 1000, num_bits_per_symbol=m,
              num_layers=layers, n_rnti=n_rnti, n_id=n_id)
    return u, c, c_ns, kw


@pytest.mark.parametrize("i", range(len(META)))
def test_tb_encoder_golden(i):
    u, c, c_ns, kw = tb_case(i)
    enc = nr_tb.TBEncoder(**kw)
    assert enc.k == u.shape[1] and enc.n == c.shape[1]
    assert np.array_equal(enc.encode(u), c)
    assert np.array_equal(nr_tb.TBEncoder(use_scrambler=False, **kw).encode(u), c_ns)


@pytest.mark.parametrize("i", [0, 3])
def test_tb_decoder_roundtrip(i):
    u, c, _, kw = tb_case(i)
    enc = nr_tb.TBEncoder(**kw)
    u_hat, ok = nr_tb.TBDecoder(enc, num_bp_iter=5, cn_update="minsum").decode(2 * c - 1)
    assert np.array_equal(u_hat, u) and ok.all()
    bad = (2 * c - 1).copy()
    bad[:, :200] *= -1                               # too many errors for 5 iterations -> CRC flags it
    _, ok = nr_tb.TBDecoder(enc, num_bp_iter=1, cn_update="minsum").decode(bad)
    assert not ok.any()


def test_tb_size_quantisation():
    # tb_encoder.py:205-214: target size quantised up, zero padded internally
    enc = nr_tb.TBEncoder(target_tb_size=1000, num_coded_bits=2400, target_coderate=1000 / 2400, num_bits_per_symbol=4)
    assert enc.tb_size >= 1000 and enc.k_padding == enc.tb_size - 1000 and enc.num_cbs == 1 and enc.tb_crc_length == 16
    u = np.random.default_rng(0).integers(0, 2, (3, 1000)).astype(np.float32)
    c = enc.encode(u)
    assert c.shape == (3, 2400)
    u_hat, ok = nr_tb.TBDecoder(enc, num_bp_iter=5, cn_update="minsum").decode(2 * c - 1)
    assert np.array_equal(u_hat, u) and ok.all()
