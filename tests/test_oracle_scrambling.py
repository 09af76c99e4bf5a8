"""Pin oracle/scrambling.py: the reference's 38.211 known-answer vector
(test/unit/fec/test_scrambling.py:632-646) and the row/column interleaver properties
(test/unit/fec/test_interleaving.py:326-420)."""
import numpy as np
import pytest

from oracle import scrambling as osc

S_REF = np.array([0., 1., 1., 1., 1., 0., 0., 0., 0., 1., 0., 1., 0., 1., 1., 1., 0., 0., 0., 1., 1., 1., 0., 0., 0., 1.,
                  1., 0., 0., 1., 1., 1., 0., 1., 0., 0., 1., 1., 1., 0., 1., 0., 0., 0., 0., 0., 1., 1., 1., 0., 1., 1.,
                  0., 1., 1., 0., 0., 0., 1., 0., 0., 1., 0., 0., 1., 0., 0., 0., 0., 0., 0., 1., 1., 1., 0., 1., 0., 0.,
                  1., 1., 0., 1., 1., 1., 0., 0., 0., 0., 0., 1., 0., 1., 1., 1., 1., 1., 1., 1., 0., 0.], np.float32)


def test_5gnr_reference_sequence():
    s = osc.apply_scrambling(np.zeros((1, 100), np.float32), osc.generate_prng_seq(100, osc.tb5g_c_init(20001, 41)))
    assert np.array_equal(s[0], S_REF)
    assert not np.array_equal(osc.generate_prng_seq(100, osc.tb5g_c_init(20002, 41)), S_REF)
    assert not np.array_equal(osc.generate_prng_seq(100, osc.tb5g_c_init(20001, 42)), S_REF)
    # PDSCH cw 0 == PUSCH, cw 1 differs (test_scrambling.py:656-680)
    assert osc.tb5g_c_init(20001, 41, "PDSCH", 0) == osc.tb5g_c_init(20001, 41, "PUSCH", 1)
    assert osc.tb5g_c_init(20001, 41, "PDSCH", 1) != osc.tb5g_c_init(20001, 41)
    # longer sequences extend shorter ones
    assert np.array_equal(osc.generate_prng_seq(300, 12345)[:100], osc.generate_prng_seq(100, 12345))


def test_scrambling_involution_and_llr_domain():
    rng = np.random.default_rng(0)
    b = rng.integers(0, 2, (5, 64)).astype(np.float32)
    seq = osc.random_scrambling_sequence(b.shape, 7)
    y = osc.apply_scrambling(b, seq)
    assert np.array_equal(osc.apply_scrambling(y, seq), b) and not np.array_equal(y, b)
    # flipping bits == flipping signs of the bipolar values (test_scrambling.py:145-153)
    z = osc.apply_scrambling(2 * b - 1, seq, binary=False)
    assert np.array_equal(0.5 * (1 + z), y)
    # keep_batch_constant: one row for the whole batch
    seq_c = osc.random_scrambling_sequence(b.shape, 7, keep_batch_constant=True)
    assert seq_c.shape == (1, 64)


@pytest.mark.parametrize("n,depth", [(12, 3), (13, 3), (100, 7), (1, 5), (64, 64), (64, 1)])
def test_rc_perm(n, depth):
    perm, inv = osc.rc_perm(n, depth)
    assert sorted(perm) == list(range(n)) and np.array_equal(perm[inv], np.arange(n))
    if n % depth == 0:       # pure row/column transpose
        assert np.array_equal(perm, np.arange(n).reshape(n // depth, depth).T.reshape(-1))
