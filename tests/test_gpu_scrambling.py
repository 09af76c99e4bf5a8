"""GPU parity of the scramblers and the row/column interleaver against oracle/scrambling.py
(bit-exact), with the properties the reference's unit tests assert
(test/unit/fec/test_scrambling.py, test_interleaving.py:326-420)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import scrambling as osc
from test_oracle_scrambling import S_REF


@pytest.fixture(scope="module")
def fec():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p.fec


def _np(t):
    return t.detach().cpu().numpy()


def test_tb5g_reference_vector_and_params(fec):
    sc = fec.scrambling
    s = _np(sc.TB5GScrambler(n_id=41, n_rnti=20001)(np.zeros((1, 100), np.float32)))[0]
    assert np.array_equal(s, S_REF)
    assert not np.array_equal(_np(sc.TB5GScrambler(n_id=41, n_rnti=20002)(np.zeros((1, 100), np.float32)))[0], S_REF)
    ref = _np(sc.TB5GScrambler(n_id=41, n_rnti=20001, channel_type="PUSCH", codeword_index=1)(np.zeros((1, 100), np.float32)))
    assert np.array_equal(ref[0], S_REF)
    assert np.array_equal(_np(sc.TB5GScrambler(n_id=41, n_rnti=20001, channel_type="PDSCH")(np.zeros((1, 100), np.float32)))[0], S_REF)
    assert not np.array_equal(_np(sc.TB5GScrambler(n_id=41, n_rnti=20001, channel_type="PDSCH", codeword_index=1)(np.zeros((1, 100), np.float32)))[0], S_REF)
    for n_r, n_id in ((-1, 0), (1.2, 10), (65536, 1023), (0, -1), (10, 1.2), (65535, 1024)):
        with pytest.raises(ValueError):
            sc.TB5GScrambler(n_id=n_id, n_rnti=n_r)
    with pytest.raises(TypeError):
        sc.TB5GScrambler(channel_type="PUCCH")


def test_tb5g_multi_stream_and_descrambler(fec):
    sc = fec.scrambling
    rng = np.random.default_rng(1)
    n_rntis, n_ids = [1, 500, 65535], [3, 77, 1023]
    x = rng.integers(0, 2, (4, 2, 3, 333)).astype(np.float32)
    s = sc.TB5GScrambler(n_rnti=n_rntis, n_id=n_ids)
    y = _np(s(x))
    for i, (r, d) in enumerate(zip(n_rntis, n_ids)):
        seq = osc.generate_prng_seq(333, osc.tb5g_c_init(r, d))
        assert np.array_equal(y[..., i, :], osc.apply_scrambling(x[..., i, :], seq))
        assert np.array_equal(y[..., i, :], _np(sc.TB5GScrambler(n_rnti=r, n_id=d)(x[..., i, :])))
    assert np.array_equal(_np(sc.Descrambler(s)(y)), x)
    # soft values: descrambling LLRs of the scrambled bits restores the sign pattern
    llr = (2 * y - 1) * rng.uniform(0.1, 5, y.shape).astype(np.float32)
    z = _np(sc.Descrambler(s, binary=False)(llr))
    assert np.array_equal(z > 0, x > 0)
    # sequence is rebuilt for a new length
    x2 = np.zeros((2, 3, 50), np.float32)
    assert np.array_equal(_np(s(x2))[0, 0], osc.generate_prng_seq(50, osc.tb5g_c_init(1, 3)))


def test_random_scrambler(fec):
    sc = fec.scrambling
    b = np.zeros((10, 100), np.float32)
    s1 = sc.Scrambler(seed=12345)
    x = _np(s1(b))
    assert np.array_equal(x, osc.random_scrambling_sequence(b.shape, 12345))
    assert set(np.unique(x)) == {0.0, 1.0} and 0.4 < x.mean() < 0.6
    assert np.array_equal(_np(sc.Descrambler(s1)(x)), b)                      # seed retrieved from the scrambler
    x2 = _np(s1(b, seed=1234))
    assert not np.array_equal(x, x2) and np.array_equal(_np(sc.Descrambler(s1)(x2, seed=1234)), b)
    assert not np.array_equal(_np(sc.Descrambler(s1)(x2)), b)
    # keep_batch_constant: same sequence for every batch item
    xc = _np(sc.Scrambler(seed=5, keep_batch_constant=True)(b))
    assert np.all(xc == xc[:1]) and np.array_equal(xc[:1], osc.random_scrambling_sequence(b.shape, 5, True))
    # binary vs soft domain (test_descrambler_nonbin)
    scr, des = sc.Scrambler(seed=1235456, binary=True), None
    des = sc.Descrambler(scr, binary=False)
    y = _np(scr(b, seed=8764))
    z = 0.5 * (1 + _np(des(2 * y - 1, seed=8764)))
    assert np.array_equal(z, b)
    # explicit sequence
    seq = np.random.default_rng(0).integers(0, 2, (1, 100)).astype(np.float32)
    xe = _np(sc.Scrambler(sequence=seq)(b))
    assert np.array_equal(xe, np.broadcast_to(seq, b.shape))
    with pytest.raises(ValueError):
        sc.Scrambler(sequence=seq + 0.5)
    with pytest.raises(TypeError):
        sc.Scrambler(seed=1.5)
    # keep_state=False: new sequence on every call
    s3 = sc.Scrambler(keep_state=False)
    assert not np.array_equal(_np(s3(b)), _np(s3(b)))


@pytest.mark.parametrize("shape,axis,depth", [((7, 12), -1, 3), ((3, 13, 5), 1, 4), ((2, 3, 100), -1, 7), ((64, 9), 0, 16)])
def test_row_column_interleaver(fec, shape, axis, depth):
    il = fec.interleaving
    rng = np.random.default_rng(3)
    x = rng.normal(size=shape).astype(np.float32)
    inter = il.RowColumnInterleaver(depth, axis=axis)
    y = _np(inter(x))
    perm, inv = osc.rc_perm(shape[axis], depth)
    assert np.array_equal(y, np.take(x, perm, axis=axis))
    assert np.array_equal(inter.perm_seq, perm) and np.array_equal(inter.perm_seq_inv, inv)
    assert np.array_equal(_np(il.Deinterleaver(inter)(y)), x)
    assert np.array_equal(_np(inter(y, inverse=True)), x)
    assert np.array_equal(_np(il.RowColumnInterleaver(depth, axis=axis, inverse=True)(y)), x)
    with pytest.raises(ValueError):
        il.RowColumnInterleaver(depth, axis=5)(x)
    with pytest.raises(TypeError):
        il.RowColumnInterleaver(2.5)


def test_random_interleaver(fec):
    """RandomInterleaver (reference interleaving.py:198-497; tests test/unit/fec/test_interleaving.py): a permutation along
    one axis, identical for all batch examples, undone by its Deinterleaver; explicit seeds pair up; keep_state=False
    draws a fresh permutation per call."""
    il = fec.interleaving
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 3, 97)).astype(np.float32)
    for axis in (-1, 1, 2):
        i = il.RandomInterleaver(seed=11, axis=axis)
        d = il.Deinterleaver(i)
        y = _np(i(x))
        assert y.shape == x.shape and not np.array_equal(y, x)
        assert np.array_equal(np.sort(y, axis=axis), np.sort(x, axis=axis))          # a permutation along the axis ...
        assert np.array_equal(_np(d(y)), x)                                           # ... undone by the deinterleaver
        ref = np.moveaxis(np.moveaxis(x, axis, -1)[..., _np(i._perm(11, x.shape[axis])[0]).astype(int)], -1, axis)
        assert np.array_equal(y, ref)                                                 # the same permutation for every example
    a, b = il.RandomInterleaver(seed=1), il.RandomInterleaver(seed=2)
    assert not np.array_equal(_np(a(x)), _np(b(x))) and np.array_equal(_np(a(x, seed=2)), _np(b(x)))
    assert np.array_equal(_np(il.Deinterleaver(a)(_np(a(x, seed=77)), seed=77)), x)
    assert np.array_equal(_np(a(x)), _np(a(x)))                                       # keep_state: the same permutation again
    f = il.RandomInterleaver(seed=5, keep_state=False)
    assert not np.array_equal(_np(f(x)), _np(f(x)))
    assert np.array_equal(_np(il.RandomInterleaver(seed=5, inverse=True)(_np(il.RandomInterleaver(seed=5)(x)))), x)
    s = il.RandomInterleaver(seed=4).find_s_min(4, 40)
    assert 1 <= s <= 40
    with pytest.raises(ValueError):
        il.RandomInterleaver(axis=0)
    with pytest.raises(NotImplementedError):
        il.RandomInterleaver(keep_batch_constant=False)
