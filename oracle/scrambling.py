"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the scramblers and the row/column interleaver:
  generate_prng_seq       /root/reference/src/sionna/phy/nr/utils.py:14-78
  Scrambler               /root/reference/src/sionna/phy/fec/scrambling.py:10-261
  TB5GScrambler           /root/reference/src/sionna/phy/fec/scrambling.py:263-468
  RowColumnInterleaver    /root/reference/src/sionna/phy/fec/interleaving.py:12-195

Pinned by the reference's known-answer vector for n_rnti=20001, n_id=41
(test/unit/fec/test_scrambling.py:632-646).  The random Scrambler draws its sequence from TF's
stateless RNG in the reference; this build specifies its own stream instead:
sequence = oracle.utils.random_bits(seed, call=1337, n)  ("parity unpinned" for that stream).
"""
import numpy as np

from . import utils as outil


def generate_prng_seq(length, c_init):
    """nr/utils.py:14-78 (38.211 5.2.1), literal loops."""
    assert length > 0 and 0 <= c_init < 2 ** 32
    n_seq, n_c = 31, 1600
    x1 = np.zeros(length + n_c + n_seq, np.int64)
    x2 = np.zeros(length + n_c + n_seq, np.int64)
    bits = [int(b) for b in format(c_init, f"0{n_seq}b")[-n_seq:]]
    x1[0] = 1
    x2[:n_seq] = bits[::-1]
    for i in range(length + n_c):
        x1[i + 31] = (x1[i + 3] + x1[i]) % 2
        x2[i + 31] = (x2[i + 3] + x2[i + 2] + x2[i + 1] + x2[i]) % 2
    return ((x1[n_c:n_c + length] + x2[n_c:n_c + length]) % 2).astype(np.float32)


def tb5g_c_init(n_rnti, n_id, channel_type="PUSCH", codeword_index=0):
    """scrambling.py:391-399"""
    if channel_type == "PUSCH":
        return n_rnti * 2 ** 15 + n_id
    return n_rnti * 2 ** 15 + codeword_index * 2 ** 14 + n_id


def apply_scrambling(x, seq, binary=True):
    """scrambling.py:252-259"""
    x = np.asarray(x, np.float32)
    if binary:
        return np.abs(x - seq).astype(np.float32)
    return (x * (np.float32(-2) * seq + np.float32(1))).astype(np.float32)


def random_scrambling_sequence(shape, seed, keep_batch_constant=False):
    """This build's stream for Scrambler._generate_scrambling (:160-183)."""
    shp = tuple(shape[1:]) if keep_batch_constant else tuple(shape)
    seq = outil.random_bits(seed, 1337, int(np.prod(shp))).reshape(shp)
    return seq[None] if keep_batch_constant else seq


def rc_perm(n_seq, row_depth):
    """interleaving.py:111-144"""
    n = int(np.ceil(n_seq / row_depth) * row_depth)
    ind = np.arange(n).reshape(n // row_depth, -1).T.reshape(-1)
    perm = ind[ind < n_seq]
    return perm, np.argsort(perm)
