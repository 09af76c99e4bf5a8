"""Host-side helpers around the BP decoder and the generic linear encoder - mirror of reference
src/sionna/phy/fec/utils.py: parity-check examples (:478-530), ``alist`` import (:650-795), GF(2)
systematic forms ``make_systematic`` / ``gm2pcm`` / ``pcm2gm`` / ``verify_gm_pcm`` (:797-1113),
``generate_reg_ldpc`` (:1115-1234), bit/integer conversions (:532-648), the J-function pair and
``llr2mi`` (:116-267) and ``GaussianPriorSource`` (:16-114).  Everything here is init-time or
analysis code on the host (NumPy); only ``GaussianPriorSource`` draws on the device RNG."""
import os
import warnings

import numpy as np
import torch

from ... import _ffi
from ..block import Block, wrap
from ..config import config

_CODES = os.path.join(os.path.dirname(__file__), "ldpc", "codes", "example_pcms.npz")


# ------------------------------------------------------------------ example codes, alist
def load_parity_check_examples(pcm_id, verbose=False):
    """Built-in example parity-check matrices: 0 (7,4) Hamming, 1 (63,45) BCH, 2 (127,106) BCH,
    3 random (3,6)-regular LDPC n=100, 4 802.11n LDPC n=648.  Returns (pcm, k, n, coderate)."""
    ex = np.load(_CODES)
    if f"shape_{pcm_id}" not in ex.files:
        raise IndexError("pcm_id out of range")
    pcm = np.zeros(tuple(ex[f"shape_{pcm_id}"]), dtype=np.float64)
    rc = ex[f"rc_{pcm_id}"]
    pcm[rc[0], rc[1]] = 1
    n = int(pcm.shape[1])
    k = int(n - pcm.shape[0])
    coderate = k / n
    if verbose:
        print(f"\nn: {n}, k: {k}, coderate: {coderate:.3f}")
    return pcm, k, n, coderate


def load_alist(path):
    """Nested list of the integers of an ``alist`` file [MacKay]; empty lines are skipped."""
    alist = []
    with open(path, "r") as reader:  # pylint: disable=unspecified-encoding
        for line in reader:
            row = [int(w) for w in line.split()]
            if row:
                alist.append(row)
    return alist


def alist2mat(alist, verbose=True):
    """``alist`` (nested list) -> (pcm [n-k, n], k, n, coderate).  The CN perspective, if present,
    must agree with the VN perspective."""
    assert len(alist) > 4, "Invalid alist format."
    n, m = alist[0][0], alist[0][1]
    v_max, c_max = alist[1][0], alist[1][1]
    k = n - m
    coderate = k / n
    vn_profile, cn_profile = alist[2], alist[3]
    assert np.sum(vn_profile) == np.sum(cn_profile), "Invalid alist format."
    assert np.max(vn_profile) == v_max, "Invalid alist format."
    assert np.max(cn_profile) == c_max, "Invalid alist format."
    if len(alist) == len(vn_profile) + 4:
        print("Note: .alist does not contain (redundant) CN perspective.")
        print("Recovering parity-check matrix from VN only.")
        print("Please verify the correctness of the results manually.")
        vn_only = True
    else:
        assert len(alist) == len(vn_profile) + len(cn_profile) + 4, "Invalid alist format."
        vn_only = False
    pcm = np.zeros((m, n))
    num_edges = 0
    for v in range(n):
        for c in alist[4 + v][:vn_profile[v]]:
            pcm[c - 1, v] = 1                       # alist indices are 1-based
            num_edges += 1
    if not vn_only:
        for c in range(m):
            for v in alist[4 + n + c][:cn_profile[c]]:
                assert pcm[c, v - 1] == 1           # both perspectives must describe the same edges
    if verbose:
        print("Number of variable nodes (columns): ", n)
        print("Number of check nodes (rows): ", m)
        print("Number of information bits per cw: ", k)
        print("Number edges: ", num_edges)
        print("Max. VN degree: ", v_max)
        print("Max. CN degree: ", c_max)
        print("VN degree: ", vn_profile)
        print("CN degree: ", cn_profile)
    return pcm, k, n, coderate


# ------------------------------------------------------------------ GF(2) linear algebra
def make_systematic(mat, is_pcm=False):
    """Gaussian elimination over GF(2) to [I | M] (generator) or [M | I] (``is_pcm``), with the
    reference's pivoting order so that the same matrices result: for column i first a row swap
    with the first lower row holding a one, otherwise a column swap with the first later column
    holding a one in row i.  Returns (matrix, list of column swaps)."""
    m, n = mat.shape
    assert m <= n, "Invalid matrix dimensions."
    if is_pcm and np.any(np.sum(mat, axis=0) == 0):
        warnings.warn("All-zero column in parity-check matrix detected. It seems as if the code contains "
                      "unprotected nodes.")
    a = np.array(mat).astype(bool)
    swaps = []
    for i in range(m):
        if not a[i, i]:
            below = np.flatnonzero(a[i + 1:, i])
            if below.size:
                r = i + 1 + int(below[0])
                a[[i, r]] = a[[r, i]]
            else:
                right = np.flatnonzero(a[i, i + 1:])
                if not right.size:
                    raise ValueError("Could not succeed; mat is not full rank?")
                c = i + 1 + int(right[0])
                a[:, [i, c]] = a[:, [c, i]]
                swaps.append([i, c])
        rows = i + 1 + np.flatnonzero(a[i + 1:, i])
        a[rows] ^= a[i]
    for i in range(m - 1, -1, -1):
        rows = np.flatnonzero(a[:i, i])
        a[rows] ^= a[i]
    assert np.array_equal(a[:, :m], np.eye(m, dtype=bool)), "Internal error, could not find systematic matrix."
    if is_pcm:                                       # move the identity to the right-hand side
        for i in range(n - 1, (n - 1) - m, -1):
            j = i - (n - m)
            a[:, [i, j]] = a[:, [j, i]]
            swaps.append([i, j])
    return a.astype(int), swaps


def verify_gm_pcm(gm, pcm):
    """True iff H G^T = 0 over GF(2)."""
    k, n = gm.shape
    n_pcm = pcm.shape[1]
    k_pcm = n_pcm - pcm.shape[0]
    assert k == k_pcm, "Inconsistent shape of gm and pcm."
    assert n == n_pcm, "Inconsistent shape of gm and pcm."
    assert ((gm == 0) | (gm == 1)).all(), "gm is not binary."
    assert ((pcm == 0) | (pcm == 1)).all(), "pcm is not binary."
    return np.sum(np.mod(np.matmul(pcm, np.transpose(gm)), 2)) == 0


def gm2pcm(gm, verify_results=True):
    """Parity-check matrix of a generator matrix [k, n] (full rank)."""
    k, n = gm.shape
    assert k < n, "Invalid matrix dimensions."
    gm_sys, swaps = make_systematic(gm, is_pcm=False)
    pcm = np.concatenate((np.transpose(gm_sys[:, -(n - k):]), np.eye(n - k)), axis=1)
    for i, j in swaps[::-1]:
        pcm[:, [i, j]] = pcm[:, [j, i]]
    if verify_results:
        assert verify_gm_pcm(gm=gm, pcm=pcm), "Resulting parity-check matrix does not match to generator matrix."
    return pcm


def pcm2gm(pcm, verify_results=True):
    """Generator matrix [k, n] of a full-rank parity-check matrix [n-k, n]."""
    n = pcm.shape[1]
    k = n - pcm.shape[0]
    assert k < n, "Invalid matrix dimensions."
    pcm_sys, swaps = make_systematic(pcm, is_pcm=True)
    gm = np.concatenate((np.eye(k), np.transpose(pcm_sys[:, :k])), axis=1)
    for i, j in swaps[::-1]:
        gm[:, [i, j]] = gm[:, [j, i]]
    if verify_results:
        assert verify_gm_pcm(gm=gm, pcm=pcm), "Resulting parity-check matrix does not match to generator matrix."
    return gm


def generate_reg_ldpc(v, c, n, allow_flex_len=True, verbose=True):
    """Random (v, c)-regular LDPC code by socket matching (no cycle optimisation); draws from
    ``config.np_rng``.  Returns (pcm, k, n, coderate)."""
    assert isinstance(allow_flex_len, bool), "allow_flex_len must be bool."
    if allow_flex_len:
        for n_mod in range(n, n + 2 * c):
            if np.mod((v / c) * n_mod, 1.) == 0:
                n = n_mod
                if verbose:
                    print("Setting n to: ", n)
                break
    coderate = 1 - (v / c)
    n_v, n_c = n, int((v / c) * n)
    k = n_v - n_c
    v_socks = np.tile(np.arange(n_v), v)
    c_socks = np.tile(np.arange(n_c), c)
    if verbose:
        print("Number of edges (VN perspective): ", len(v_socks))
        print("Number of edges (CN perspective): ", len(c_socks))
    assert len(v_socks) == len(c_socks), \
        "Number of edges from VN and CN perspective does not match. Consider to (slightly) change n."
    rng = config.np_rng
    rng.shuffle(v_socks)
    rng.shuffle(c_socks)
    pcm = np.zeros([n_c, n_v])
    idx, stalls = 0, 0
    while idx < len(v_socks):
        if pcm[c_socks[idx], v_socks[idx]] == 0:
            pcm[c_socks[idx], v_socks[idx]] = 1
            idx += 1
            stalls = 0
        else:                                           # double edge: reshuffle the open sockets
            stalls += 1
            if stalls >= 200:
                print("Stopping - no solution found!")
                break
            rng.shuffle(v_socks[idx:])
            rng.shuffle(c_socks[idx:])
    assert (np.sum(pcm, axis=0) == v).all(), "VN degree not always v."
    assert (np.sum(pcm, axis=1) == c).all(), "CN degree not always c."
    if verbose:
        print(f"Generated regular ({v},{c}) LDPC code of length n={n}")
        print(f"Code rate is r={coderate:.3f}.")
    return pcm, k, n, coderate


# ------------------------------------------------------------------ bit / integer helpers
def bin2int(arr):
    """[1, 0, 1] -> 5 (MSB first); ``None`` for an empty input."""
    if len(arr) == 0:
        return None
    return int("".join(str(int(x)) for x in arr), 2)


def int2bin(num, length):
    """5, 4 -> [0, 1, 0, 1] (MSB first, truncated to the ``length`` least significant bits)."""
    assert num >= 0, "Input integer should be non-negative"
    assert length >= 0, "length should be non-negative"
    s = format(int(num), f"0{length}b")
    return [int(x) for x in s[-length:]] if length else []


def bin2int_tf(arr):
    """Array version of :func:`bin2int` along the last axis."""
    a = np.asarray(arr.cpu() if isinstance(arr, torch.Tensor) else arr).astype(np.int64)
    shifts = np.arange(a.shape[-1] - 1, -1, -1)
    return np.sum(a << shifts, axis=-1)


def int2bin_tf(ints, length):
    """Array version of :func:`int2bin`: [...] -> [..., length]."""
    assert length >= 0
    a = np.asarray(ints.cpu() if isinstance(ints, torch.Tensor) else ints).astype(np.int64)
    shifts = np.arange(length - 1, -1, -1)
    return (a[..., None] >> shifts) % 2


def int_mod_2(x):
    """x mod 2 for integer-valued arrays (float inputs are rounded first)."""
    a = np.asarray(x.cpu() if isinstance(x, torch.Tensor) else x)
    if np.issubdtype(a.dtype, np.integer):
        return a & 1
    r = np.abs(np.round(a))
    return (r - 2 * np.floor(r / 2)).astype(a.dtype)


# ------------------------------------------------------------------ mutual information helpers
def j_fun(mu):
    """J-function approximation of [Brannstrom]: mean of the Gaussian LLRs -> mutual information."""
    mu = np.minimum(np.maximum(np.asarray(mu, np.float64), 1e-10), 1000)
    h1, h2, h3 = 0.3073, 0.8935, 1.1064
    return (1 - 2 ** (-h1 * (2 * mu) ** h2)) ** h3


def j_fun_inv(mi):
    """Inverse J-function, output clipped to 20."""
    mi = np.minimum(np.maximum(np.asarray(mi, np.float64), 1e-10), 1.)
    h1, h2, h3 = 0.3073, 0.8935, 1.1064
    with np.errstate(divide="ignore"):
        mu = 0.5 * ((-1 / h1) * np.log2(1 - mi ** (1 / h3))) ** (1 / h2)
    return np.minimum(mu, 20)


def llr2mi(llr, s=None, reduce_dims=True):
    """Mutual information estimate 1 - E[log2(1 + exp(s * llr))] of LLRs of the all-zero codeword
    (or with the signs ``s`` = +-1 of the transmitted bits).  Analysis helper: evaluated on the host."""
    a = np.asarray(llr.detach().cpu() if isinstance(llr, torch.Tensor) else llr)
    if not np.issubdtype(a.dtype, np.floating):
        raise TypeError("Dtype of llr must be a real-valued float.")
    if s is not None:
        a = a * np.asarray(s.detach().cpu() if isinstance(s, torch.Tensor) else s).astype(a.dtype)
    x = np.log2(1. + np.exp(np.clip(a, -100., 100.)))
    return 1. - (np.mean(x) if reduce_dims else np.mean(x, axis=-1))


class GaussianPriorSource(Block):
    """``GaussianPriorSource()(output_shape, no=None, mi=None)``: LLRs of the all-zero codeword over a
    BPSK/AWGN channel with noise variance ``no`` (or with mutual information ``mi`` through the
    J-function): N(-mu, sigma^2) with sigma^2 = 4/no, mu = sigma^2/2."""

    def call(self, output_shape, no=None, mi=None):
        if no is None:
            if mi is None:
                raise ValueError("Either no or mi must be provided.")
            mi = float(min(max(float(mi), 1e-7), 1.))
            mu_llr = float(j_fun_inv(mi))
            sigma_llr = float(np.sqrt(2 * mu_llr))
        else:
            no = max(float(no), 1e-7)
            sigma_llr = float(np.sqrt(4 / no))
            mu_llr = sigma_llr ** 2 / 2
        from ..utils.misc import complex_normal
        w = complex_normal(list(output_shape), 2.0 * sigma_llr ** 2, precision=self.precision).as_subclass(torch.Tensor)   # real part ~ N(0, sigma^2)
        return wrap((w.real - mu_llr).contiguous())
