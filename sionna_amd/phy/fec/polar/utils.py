"""Polar code construction helpers - mirror of reference src/sionna/phy/fec/polar/utils.py:13-112."""
import os

import numpy as np

_SEQ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "codes", "polar_5g_sequence.npy")


def generate_5g_ranking(k, n, sort=True):
    """Frozen and information positions of the 5G Polar code (38.212 Tab. 5.3.1.2-1): the n-k
    least reliable of the channels 0..n-1 are frozen.  Returns [frozen_pos, info_pos]."""
    if not isinstance(k, int):
        raise TypeError("k must be integer.")
    if not isinstance(n, int):
        raise TypeError("n must be integer.")
    if not isinstance(sort, bool):
        raise TypeError("sort must be bool.")
    if k < 0:
        raise ValueError("k cannot be negative.")
    if k > 1024:
        raise ValueError("k cannot be larger than 1024.")
    if n > 1024:
        raise ValueError("n cannot be larger than 1024.")
    if n < 32:
        raise ValueError("n must be >=32.")
    if n < k:
        raise ValueError("Invalid coderate (>1).")
    if np.log2(n) != int(np.log2(n)):
        raise ValueError("n must be a power of 2.")
    seq = np.load(_SEQ).astype(int)            # channel indices in ascending reliability
    seq = seq[seq < n]
    frozen, info = seq[:n - k], seq[n - k:]
    if sort:
        frozen, info = np.sort(frozen), np.sort(info)
    return [frozen.astype(int), info.astype(int)]


def generate_rm_code(r, m):
    """Reed-Muller code RM(r, m) as a Polar-like code (reference polar/utils.py:148-214): the rows of the Polar
    transform whose index has Hamming weight < m - r are frozen.  Returns (frozen_pos, info_pos, n, k, d_min)."""
    if not isinstance(r, int):
        raise TypeError("r must be int.")
    if not isinstance(m, int):
        raise TypeError("m must be int.")
    if r > m:
        raise ValueError("order r cannot be larger than m.")
    if r < 0:
        raise ValueError("r must be positive.")
    if m < 0:
        raise ValueError("m must be positive.")
    from math import comb
    n = 1 << m
    idx = np.arange(n)
    weight = np.zeros(n, int)
    for b in range(m):
        weight += (idx >> b) & 1
    frozen = weight < m - r
    k = int(np.sum(~frozen))
    if k != sum(comb(m, i) for i in range(r + 1)):
        raise ValueError("Error: resulting k is inconsistent.")
    return idx[frozen], idx[~frozen], n, k, 1 << (m - r)


def generate_polar_transform_mat(n_lift):
    """The Polar transformation matrix: the ``n_lift``-fold Kronecker power of [[1, 0], [1, 1]], [2^n_lift, 2^n_lift] of 0 / 1
    (reference polar/utils.py:114-146).  Entry (i, j) is 1 iff the set bits of j are a subset of those of i."""
    if int(n_lift) != n_lift:
        raise ValueError("n_lift must be integer.")
    if n_lift < 0:
        raise ValueError("n_lift must be positive.")
    if n_lift >= 20:
        raise ValueError("Warning: the resulting code length is large (=2^n_lift).")
    idx = np.arange(1 << max(int(n_lift), 1))
    return ((idx[:, None] & idx[None, :]) == idx[None, :]).astype(float)


def generate_dense_polar(frozen_pos, n, verbose=True):
    """Dense parity-check and generator matrices of the Polar code with the given frozen positions (Lemma 1 of [Goala_LP]; reference
    polar/utils.py:217-290): gm = the information rows of the transformation matrix [k, n], pcm = its frozen columns
    transposed [n-k, n]; the all-zero syndrome pcm gm^T is verified.  (Usable with LinearEncoder and LDPCBPDecoder; the
    graph is dense, PolarBPDecoder is the iterative decoder to use.)"""
    import numbers
    if not isinstance(n, numbers.Number):
        raise TypeError("n must be a number.")
    n = int(n)
    frozen_pos = np.asarray(frozen_pos)
    if not np.issubdtype(frozen_pos.dtype, np.integer):
        raise TypeError("frozen_pos must consist of ints.")
    if len(frozen_pos) > n:
        raise ValueError("Number of elements in frozen_pos cannot be greater than n.")
    if n < 1 or n & (n - 1):
        raise ValueError("n must be a power of 2.")
    info_pos = np.setdiff1d(np.arange(n), frozen_pos)
    if n - len(frozen_pos) != len(info_pos):
        raise ArithmeticError("Internal error: invalid info_pos generated.")
    g = generate_polar_transform_mat(n.bit_length() - 1)
    gm, pcm = g[info_pos, :], np.transpose(g[:, frozen_pos])
    if verbose:
        print("Shape of the generator matrix: ", gm.shape)
        print("Shape of the parity-check matrix: ", pcm.shape)
    if np.any((pcm.astype(np.int64) @ gm.astype(np.int64).T) & 1):
        raise ArithmeticError("Non-zero syndrome for H*G'.")
    return pcm, gm
