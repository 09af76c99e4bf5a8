"""Oracle (NumPy, CPU): QAM constellation, mapper and LLR demapper.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates ``mapping.py`` of the
reference (paths relative to /root/reference/src/sionna/phy):

* pam_gray / qam               mapping.py:15-118
* Mapper.call                  mapping.py:497-519
* Demapper.call                mapping.py:664-691
* SymbolLogits2LLRs            mapping.py:880-967  (app = logsumexp, maxlog = max)

The demapper formula is pinned by the reference's own NumPy test
(test/unit/mapping/test_mapping.py:175-199): scipy ``logsumexp`` over the point sets
C_{i,1} / C_{i,0}, atol 1e-5.
"""
import numpy as np


def pam_gray(b):
    """mapping.py:15-42"""
    if len(b) > 1:
        return (1 - 2 * b[0]) * (2 ** len(b[1:]) - pam_gray(b[1:]))
    return 1 - 2 * b[0]


def qam(num_bits_per_symbol, normalize=True, dtype=np.complex64):
    """mapping.py:44-118"""
    if num_bits_per_symbol % 2 != 0 or num_bits_per_symbol <= 0:
        raise ValueError("num_bits_per_symbol must be a multiple of 2")
    rdtype = np.float32 if dtype == np.complex64 else np.float64
    c = np.zeros([2 ** num_bits_per_symbol], dtype=dtype)
    for i in range(2 ** num_bits_per_symbol):
        b = np.array(list(np.binary_repr(i, num_bits_per_symbol)), dtype=np.int32)
        c[i] = pam_gray(b[0::2]) + 1j * pam_gray(b[1::2])
    if normalize:
        n = num_bits_per_symbol // 2
        qam_var = 1 / (2 ** (n - 2)) * np.sum(
            np.linspace(1, 2 ** n - 1, 2 ** (n - 1), dtype=rdtype) ** 2)
        c /= np.sqrt(qam_var)
    return c


def mapper(bits, points):
    """mapping.py:497-519: [...,n] 0/1 -> [...,n/m] symbols (MSB first)."""
    m = int(np.log2(len(points)))
    bits = np.asarray(bits).astype(np.int32)
    b = bits.reshape(bits.shape[:-1] + (bits.shape[-1] // m, m))
    idx = np.sum(b << np.arange(m - 1, -1, -1, dtype=np.int32), axis=-1)
    return points[idx]


def _bit_sets(m):
    """mapping.py:888-901: indices of the points whose i-th label bit is 0 / 1."""
    num_points = 2 ** m
    a = np.zeros([num_points, m], np.int32)
    for i in range(num_points):
        a[i, :] = np.array(list(np.binary_repr(i, m)), dtype=np.int32)
    c0 = np.zeros([num_points // 2, m], np.int64)
    c1 = np.zeros([num_points // 2, m], np.int64)
    for i in range(m):
        c0[:, i] = np.where(a[:, i] == 0)[0]
        c1[:, i] = np.where(a[:, i] == 1)[0]
    return c0, c1


def _logsumexp(x, axis):
    mx = np.max(x, axis=axis, keepdims=True)
    mx = np.where(np.isfinite(mx), mx, 0)
    return (np.log(np.sum(np.exp(x - mx), axis=axis, keepdims=True)) + mx).squeeze(axis)


def demapper(y, no, points, method="app", hard_out=False, prior=None):
    """mapping.py:664-691 + 927-967.  y: [...,S] complex; no scalar or [...,S]; prior: LLRs [m] or [...,S,m].

    Returns logits log p(b=1)/p(b=0), shape [..., S*m], symbol-major then bit index.
    """
    rdtype = np.float32 if y.dtype == np.complex64 else np.float64
    m = int(np.log2(len(points)))
    c0, c1 = _bit_sets(m)
    sq = np.abs(y[..., None] - points.astype(y.dtype)) ** 2          # [...,S,2^m]
    no = np.asarray(no, rdtype)
    if no.ndim > 0:
        no = np.broadcast_to(no, y.shape)
    no = np.maximum(no[..., None], np.finfo(rdtype).tiny)
    expo = (-sq / no).astype(rdtype)
    if prior is not None:                                            # :944-958
        pr = np.broadcast_to(np.asarray(prior, rdtype), y.shape + (m,))
        lab = ((np.arange(2 ** m)[:, None] >> (m - 1 - np.arange(m))) & 1) * 2 - 1        # [2^m, m] +-1
        logsig = -np.logaddexp(0, -(lab * pr[..., None, :]).astype(np.float64))
        expo = (np.sum(logsig, axis=-1) + expo).astype(rdtype)
    e0 = expo[..., c0]                                               # [...,S,2^m/2,m]
    e1 = expo[..., c1]
    if method == "app":
        llr = _logsumexp(e1, axis=-2) - _logsumexp(e0, axis=-2)
    elif method == "maxlog":
        llr = np.max(e1, axis=-2) - np.max(e0, axis=-2)
    else:
        raise AssertionError("Unknown demapping method")
    llr = llr.reshape(y.shape[:-1] + (y.shape[-1] * m,)).astype(rdtype)
    if hard_out:
        return (llr > 0).astype(rdtype)
    return llr


def symbol_demapper(y, no, points, prior=None, hard_out=False):
    """SymbolDemapper.call (mapping.py:776-792) in float64: log_softmax over the points of -|y - c|^2 / no (+ prior), or
    the index of the most likely point."""
    y = np.asarray(y, np.complex128)
    d = np.abs(y[..., None] - np.asarray(points, np.complex128))
    no = np.asarray(no, np.float64)
    if no.ndim > 0:
        no = np.broadcast_to(no, y.shape)[..., None]
    e = -d ** 2 / no
    if prior is not None:
        e = e + np.asarray(prior, np.float64)
    if hard_out:
        return np.argmax(e, axis=-1).astype(np.int32)
    mx = np.max(e, axis=-1, keepdims=True)
    return e - (mx + np.log(np.sum(np.exp(e - mx), axis=-1, keepdims=True)))


def symbol_logits2llrs(logits, num_bits_per_symbol, method="app", prior=None, hard_out=False):
    """SymbolLogits2LLRs.call (mapping.py:927-967) in float64: logits [..., 2^m] -> LLRs [..., m]."""
    m = num_bits_per_symbol
    z = np.asarray(logits, np.float64)
    c0, c1 = _bit_sets(m)
    labels = (np.arange(1 << m)[:, None] >> np.arange(m - 1, -1, -1)[None, :]) & 1           # [P, m], MSB first
    if prior is not None:
        pr = np.asarray(prior, np.float64)[..., None, :]                                       # [..., 1, m]
        a = 2.0 * labels - 1.0
        x = a * pr
        ls = np.where(x < 0, x - np.log1p(np.exp(np.minimum(x, 0))), -np.log1p(np.exp(-np.maximum(x, 0))))
        z = z + np.sum(ls, axis=-1)
    red = (lambda v: _logsumexp(v, -2)) if method == "app" else (lambda v: np.max(v, axis=-2))
    llr = red(z[..., c1]) - red(z[..., c0])
    return (llr > 0).astype(np.float32) if hard_out else llr


# ------------------------------------------------------------------ bit LLRs <-> point logits, moments, index tables
def _labels(num_bits):
    """[2^num_bits, num_bits]: binary representation of the index, MSB first (mapping.py:1031-1034, 1165-1170)."""
    return (np.arange(1 << num_bits)[:, None] >> np.arange(num_bits - 1, -1, -1)[None, :]) & 1


def llrs2symbol_logits(llrs, num_bits_per_symbol, hard_out=False):
    """LLRs2SymbolLogits.call (mapping.py:1043-1058) in float64: llrs [..., m] -> logits [..., 2^m] = sum_j
    log_sigmoid(a_cj llr_j), a = +-1 labels; hard_out: argmax (first maximum) as int32."""
    a = 2.0 * _labels(num_bits_per_symbol) - 1.0
    x = a * np.asarray(llrs, np.float64)[..., None, :]
    ls = np.where(x < 0, x - np.log1p(np.exp(np.minimum(x, 0))), -np.log1p(np.exp(-np.maximum(x, 0))))
    logits = np.sum(ls, axis=-1)
    return np.argmax(logits, axis=-1).astype(np.int32) if hard_out else logits


def symbol_logits2moments(logits, points):
    """SymbolLogits2Moments.call (mapping.py:1125-1138) in float64 -> (mean complex [...], var [...])."""
    z = np.asarray(logits, np.float64)
    p = np.exp(z - z.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    pts = np.asarray(points, np.complex128)
    mean = np.sum(p * pts, -1)
    var = np.sum(p * np.abs(pts - mean[..., None]) ** 2, -1)
    return mean, var


def symbol_inds2bits(ind, num_bits_per_symbol):
    """SymbolInds2Bits.call (mapping.py:1177-1178)."""
    return _labels(num_bits_per_symbol)[np.asarray(ind)].astype(np.float32)


def qam2pam(ind_qam, num_bits_per_symbol):
    """QAM2PAM.__call__ (mapping.py:1212-1231): even label bits -> pam1 index, odd label bits -> pam2 index."""
    lab = _labels(num_bits_per_symbol)
    base = 1 << np.arange(num_bits_per_symbol // 2 - 1, -1, -1)
    t1, t2 = np.sum(lab[:, 0::2] * base, -1), np.sum(lab[:, 1::2] * base, -1)
    q = np.asarray(ind_qam)
    return t1[q].astype(np.int32), t2[q].astype(np.int32)


def pam2qam_table(num_bits_per_symbol):
    """qam_ind[i, j] of PAM2QAM.__init__ (mapping.py:1278-1291): the QAM index whose label interleaves those of i and j."""
    nbh = num_bits_per_symbol // 2
    lab = _labels(nbh)
    P = 1 << nbh
    b = np.zeros([P, P, num_bits_per_symbol], np.int64)
    b[:, :, 0::2] = lab[:, None, :]
    b[:, :, 1::2] = lab[None, :, :]
    return np.sum(b * (1 << np.arange(num_bits_per_symbol - 1, -1, -1)), -1)


def pam2qam(pam1, pam2, num_bits_per_symbol, hard_in_out=True):
    """PAM2QAM.__call__ (mapping.py:1293-1314).  Indices: table lookup.  Logits: the P x P matrix pam1_i + pam2_j flattened
    and GATHERED with the flattened table - the reference's expression, literally (for 64-QAM and above the bit
    permutation is not an involution, so this is not the scatter one might expect; parity follows the reference)."""
    tab = pam2qam_table(num_bits_per_symbol)
    if hard_in_out:
        return tab[np.asarray(pam1), np.asarray(pam2)].astype(np.int32)
    a, b = np.asarray(pam1), np.asarray(pam2)
    mat = a[..., :, None] + b[..., None, :]
    flat = mat.reshape(mat.shape[:-2] + (-1,))
    return flat[..., tab.reshape(-1)]
