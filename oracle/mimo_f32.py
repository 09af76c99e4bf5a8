"""Oracle (NumPy float32, CPU): the per-resource-element linear MIMO equalisers in SINGLE precision with a
DEFINED operation order.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates (paths relative to /root/reference/src/sionna/phy):

* lmmse_equalizer          mimo/equalization.py:101-233  (x_hat = diag(GH)^-1 G y, no_eff = Re(1/diag(GH) - 1))
* whiten_channel           mimo/utils.py:292-356         (L = chol(S), y <- L^-1 y, H <- L^-1 H)
* lmmse_matrix             mimo/equalization.py:11-99    (G = (H^H H + I)^-1 H^H through a Cholesky solve)
* inv_cholesky / matrix_pinv   utils/linalg.py:8-58
* zf_equalizer             mimo/equalization.py:235-298
* mf_equalizer             mimo/equalization.py:300-463
* OFDMEqualizer.call       ofdm/equalization.py:109-275  (covariance S = H_u H_u^H + diag(no) + diag(sum err_var))

Why a second oracle next to ``oracle/ofdm.py``'s complex128 one: the reference evaluates these formulas in
complex64 through TensorFlow's batched LAPACK-style kernels, whose internal operation order is not part of
its contract, so single-precision results are only defined up to ``cond(S) * 2^-24``.  Like for belief
propagation (``oracle/ldpc_bp.py``) this file therefore DEFINES the order - plain textbook algorithms, every
sum sequential in ascending index, no fused multiply-add, IEEE division and square root:

* complex product  (a+ib)(c+id) = (ac - bd) + i(ad + bc); a conj(b) = (ac + bd) + i(bc - ad)
* Cholesky-Banachiewicz by columns j: d_j = S_jj - sum_{k<j} |L_jk|^2 (k ascending),
  L_jj = sqrt(d_j), L_ij = (S_ij - sum_{k<j} L_ik conj(L_jk)) * (1 / L_jj)
* forward / backward substitution rows in order, subtraction of the products in ascending (forward) /
  ascending from i+1 (backward) column order, then one multiplication with the reciprocal of the diagonal
* Gramians / matrix-vector products: accumulate over the contracted index in ascending order starting from
  the additive term (identity, S, 0)

so that a float32 implementation that follows the same order is reproducible BIT FOR BIT, and every other
float32 implementation (the reference's) agrees within the conditioning bound that
``tests/test_oracle_mimo_f32.py`` asserts against the complex128 restatement.  ``csrc/mimo.hip`` follows
this order (compiled with -ffp-contract=off)."""
import numpy as np

F = np.float32


class C:
    """Array of complex numbers as two float32 arrays (no complex dtype: NumPy's complex64 product may use
    FMA / different association on some builds)."""
    __slots__ = ("re", "im")

    def __init__(self, re, im):
        self.re, self.im = np.asarray(re, F), np.asarray(im, F)

    def __add__(a, b): return C(a.re + b.re, a.im + b.im)
    def __sub__(a, b): return C(a.re - b.re, a.im - b.im)
    def __mul__(a, b): return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re)
    def mulc(a, b): return C(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im)      # a * conj(b)
    def scale(a, s): return C(a.re * s, a.im * s)
    def conj(a): return C(a.re, -a.im)

    def div(a, b):
        d = b.re * b.re + b.im * b.im
        return C((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d)

    def to_np(a): return (a.re + 1j * a.im.astype(np.complex64)).astype(np.complex64)


def _split(x, idx):
    x = np.asarray(x)
    v = x[(Ellipsis,) + idx]
    return C(np.real(v).astype(F), np.imag(v).astype(F))


def _zero(n): return C(np.zeros(n, F), np.zeros(n, F))
def _const(n, r): return C(np.full(n, r, F), np.zeros(n, F))


def cholesky(a, n):
    """In place lower Cholesky of the n x n Hermitian matrices a[i][j] (lower part used)."""
    for j in range(n):
        d = a[j][j].re
        for k in range(j):
            d = d - (a[j][k].re * a[j][k].re + a[j][k].im * a[j][k].im)
        l = np.sqrt(d)
        a[j][j] = C(l, np.zeros_like(l))
        inv = F(1) / l
        for i in range(j + 1, n):
            v = a[i][j]
            for k in range(j):
                v = v - a[i][k].mulc(a[j][k])
            a[i][j] = v.scale(inv)


def _whiten(y, h, s, M, K):
    cholesky(s, M)
    for i in range(M):
        v = y[i]
        for k in range(i):
            v = v - s[i][k] * y[k]
        y[i] = v.scale(F(1) / s[i][i].re)
        for c in range(K):
            w = h[i][c]
            for k in range(i):
                w = w - s[i][k] * h[k][c]
            h[i][c] = w.scale(F(1) / s[i][i].re)


def _chol_solve_rows(a, h, M, K, n):
    """g[i][m]: solution of (L L^H) g[:, m] = H^H e_m for the factor L stored in a."""
    g = [[None] * M for _ in range(K)]
    for m in range(M):
        z = [None] * K
        for i in range(K):
            v = h[m][i].conj()
            for k in range(i):
                v = v - a[i][k] * z[k]
            z[i] = v.scale(F(1) / a[i][i].re)
        for i in range(K - 1, -1, -1):
            v = z[i]
            for k in range(i + 1, K):
                v = v - a[k][i].conj() * g[k][m]
            g[i][m] = v.scale(F(1) / a[i][i].re)
    return g


def _unpack(y, h, s):
    y, h, s = np.asarray(y), np.asarray(h), np.asarray(s)
    M, K = h.shape[-2:]
    lead = h.shape[:-2]
    y = np.broadcast_to(y, lead + (M,)).reshape(-1, M)
    s = np.broadcast_to(s, lead + (M, M)).reshape(-1, M, M)
    h = h.reshape(-1, M, K)
    yy = [_split(y, (m,)) for m in range(M)]
    hh = [[_split(h, (m, k)) for k in range(K)] for m in range(M)]
    ss = [[_split(s, (i, j)) for j in range(M)] for i in range(M)]
    return yy, hh, ss, M, K, lead, h.shape[0]


def _pack(xh, ne, lead, K):
    x = np.stack([v.to_np() for v in xh], axis=-1).reshape(lead + (K,))
    n = np.stack([np.asarray(v, F) for v in ne], axis=-1).reshape(lead + (K,))
    return x, n


def lmmse_equalizer(y, h, s, whiten_interference=True):
    """y [...,M], h [...,M,K], s [...,M,M] (complex64) -> x_hat [...,K] complex64, no_eff [...,K] float32."""
    with np.errstate(all="ignore"):
        yy, hh, ss, M, K, lead, n = _unpack(y, h, s)
        xh, ne = _lmmse_core(yy, hh, ss, M, K, n, whiten_interference)
        return _pack(xh, ne, lead, K)


def _lmmse_core(y, h, s, M, K, n, whiten):
    if whiten:
        _whiten(y, h, s, M, K)
        a = [[None] * K for _ in range(K)]
        for i in range(K):
            for j in range(i + 1):
                v = _const(n, 1.0 if i == j else 0.0)
                for m in range(M):
                    v = v + h[m][j].mulc(h[m][i])
                a[i][j] = v
        cholesky(a, K)
        g = _chol_solve_rows(a, h, M, K, n)
    else:
        q = [[None] * M for _ in range(M)]
        for i in range(M):
            for j in range(i + 1):
                v = s[i][j]
                for c in range(K):
                    v = v + h[i][c].mulc(h[j][c])
                q[i][j] = v
        cholesky(q, M)
        g = [[None] * M for _ in range(K)]
        for c in range(K):
            z = [None] * M
            for i in range(M):
                v = h[i][c]
                for k in range(i):
                    v = v - q[i][k] * z[k]
                z[i] = v.scale(F(1) / q[i][i].re)
            gt = [None] * M
            for i in range(M - 1, -1, -1):
                v = z[i]
                for k in range(i + 1, M):
                    v = v - q[k][i].conj() * gt[k]
                gt[i] = v.scale(F(1) / q[i][i].re)
            for i in range(M):
                g[c][i] = gt[i].conj()
    xh, ne = [], []
    one = _const(n, 1.0)
    for k in range(K):
        gy, d = _zero(n), _zero(n)
        for m in range(M):
            gy = gy + g[k][m] * y[m]
            d = d + g[k][m] * h[m][k]
        xh.append(gy.div(d))
        ne.append(one.div(d).re - F(1))
    return xh, ne


def _zf_mf_core(y, h, s, M, K, n, mf):
    a = [[None] * K for _ in range(K)]
    for i in range(K):
        for j in range(K):
            v = _zero(n)
            for m in range(M):
                v = v + h[m][j].mulc(h[m][i])
            a[i][j] = v
    if mf:
        g = [[h[m][k].conj().div(a[k][k]) for m in range(M)] for k in range(K)]
    else:
        l = [[a[i][j] for j in range(K)] for i in range(K)]
        cholesky(l, K)
        g = _chol_solve_rows(l, h, M, K, n)
    xh, ne = [], []
    for k in range(K):
        gy = _zero(n)
        for m in range(M):
            gy = gy + g[k][m] * y[m]
        xh.append(gy)
        q = _zero(n)
        for a2 in range(M):
            for b2 in range(M):
                sv = s[a2][b2] if b2 <= a2 else s[b2][a2].conj()
                q = q + (g[k][a2] * sv).mulc(g[k][b2])
        if mf:
            r = np.zeros(n, F)
            for j in range(K):
                gh = _zero(n)
                for m in range(M):
                    gh = gh + g[k][m] * h[m][j]
                e_re = F(1.0 if k == j else 0.0) - gh.re
                e_im = -gh.im
                r = r + (e_re * e_re + e_im * e_im)
            re, im = r + q.re, q.im
            ne.append(np.sqrt(re * re + im * im))
        else:
            ne.append(q.re)
    return xh, ne


def zf_equalizer(y, h, s):
    with np.errstate(all="ignore"):
        yy, hh, ss, M, K, lead, n = _unpack(y, h, s)
        return _pack(*_zf_mf_core(yy, hh, ss, M, K, n, False), lead, K)


def mf_equalizer(y, h, s):
    with np.errstate(all="ignore"):
        yy, hh, ss, M, K, lead, n = _unpack(y, h, s)
        return _pack(*_zf_mf_core(yy, hh, ss, M, K, n, True), lead, K)


def ofdm_covariance(hu, no_dt, ev):
    """S of OFDMEqualizer.call in float32 with a defined order: diagonal = no + err_var of stream 0 + stream 1
    + ... (ALL streams, ascending global stream id), then + h_u conj(h_u)^T for the undesired streams in
    ascending order; lower triangle only (the upper one is its conjugate).
    hu [...,M,U] complex64, no_dt [...,M] float32, ev [...,M,S] float32 -> [...,M,M] complex64."""
    M, U = hu.shape[-2:]
    lead = hu.shape[:-2]
    s = np.zeros(lead + (M, M), np.complex64)
    s_re, s_im = np.zeros(lead + (M, M), F), np.zeros(lead + (M, M), F)
    for m in range(M):
        dg = np.asarray(no_dt[..., m], F).copy()
        for q in range(ev.shape[-1]):
            dg = dg + ev[..., m, q].astype(F)
        s_re[..., m, m] = dg
    for u in range(U):
        hr, hi = np.real(hu[..., u]).astype(F), np.imag(hu[..., u]).astype(F)
        for a in range(M):
            for c in range(a + 1):
                s_re[..., a, c] = s_re[..., a, c] + (hr[..., a] * hr[..., c] + hi[..., a] * hi[..., c])
                s_im[..., a, c] = s_im[..., a, c] + (hi[..., a] * hr[..., c] - hr[..., a] * hi[..., c])
    for a in range(M):
        for c in range(a):
            s_re[..., c, a] = s_re[..., a, c]
            s_im[..., c, a] = -s_im[..., a, c]
    s.real, s.imag = s_re, s_im
    return s


def ofdm_equalize(rg, sm, y, h_hat, err_var, no, kind="lmmse", whiten_interference=True):
    """OFDMEqualizer.call (ofdm/equalization.py:109-275) in float32: same layout handling as
    ``oracle.ofdm._ofdm_preprocess`` / ``_extract_data`` with the float32 covariance and solve above.
    Returns x_hat, no_eff [B, tx, streams, num_data]."""
    from . import ofdm as o
    y = np.asarray(y, np.complex64)
    h_hat = np.asarray(h_hat, np.complex64)
    y_eff = o.remove_nulled(rg, y)
    y_dt = np.transpose(y_eff, [0, 1, 3, 4, 2])                                   # [B,rx,T,F,M]
    ev = np.broadcast_to(np.asarray(err_var, F), h_hat.shape)
    ev = np.transpose(ev, [0, 1, 5, 6, 2, 3, 4])
    ev = ev.reshape(ev.shape[:5] + (-1,))                                          # [B,rx,T,F,M,S]
    h_dt = np.transpose(h_hat, [1, 3, 4, 0, 2, 5, 6])
    h_dt = h_dt.reshape((-1,) + h_dt.shape[3:])
    hd = h_dt[sm.detection_desired_ind].reshape((sm.num_rx, sm.num_streams_per_rx) + h_dt.shape[1:])
    # np.where(stream_association == 0) lists every receiver's undesired streams in ascending global stream id -
    # the order the covariance accumulates them in
    hu = h_dt[sm.detection_undesired_ind].reshape((sm.num_rx, -1) + h_dt.shape[1:])
    perm = [2, 0, 4, 5, 3, 1]
    hd, hu = np.transpose(hd, perm), np.transpose(hu, perm)                        # [B,rx,T,F,M,K/U]
    no = np.asarray(no, F)
    no_dt = no.reshape(no.shape + (1,) * (3 - no.ndim))
    no_dt = np.broadcast_to(no_dt, y.shape[:3])[..., None, None]
    no_dt = np.transpose(np.broadcast_to(no_dt, y_eff.shape), [0, 1, 3, 4, 2])     # [B,rx,T,F,M]
    s = ofdm_covariance(hu, no_dt, ev)
    if kind == "lmmse":
        x_hat, no_eff = lmmse_equalizer(y_dt, hd, s, whiten_interference)
    else:
        x_hat, no_eff = {"zf": zf_equalizer, "mf": mf_equalizer}[kind](y_dt, hd, s)
    B = y.shape[0]
    return o._extract_data(rg, sm, x_hat, B).astype(np.complex64), o._extract_data(rg, sm, no_eff, B).astype(F)
