"""Spatial correlation models of the flat-fading channel - mirrors of ``SpatialCorrelation``, ``KroneckerModel`` and
``PerColumnModel`` (reference src/sionna/phy/channel/spatial_correlation.py:13-195) and of ``exp_corr_mat`` /
``one_ring_corr_mat`` (channel/utils.py:1490-1652).

The Cholesky factors are taken on the host when the matrices are set (they are [M, M] / [K, K]); applying a model to
channel matrices h [..., M, K] is ONE launch of ``samd_spatial_corr_c64`` with the [M K, M K] matrix that acts on the
rx-major vector of h:  L_rx h L_tx^H  <->  (L_rx (x) conj(L_tx)) vec(h);  per-column L_rx[k] h[:, k]  <->  a block matrix.
Batched correlation matrices (one per example) have no HIP path."""
from abc import ABC, abstractmethod

import numpy as np
import torch

from ... import _ffi
from ..block import Object, wrap
from ..config import config, dtypes


def _toeplitz(col):
    """Hermitian Toeplitz matrices from their first columns [..., n] (first row = conjugate): LinearOperatorToeplitz."""
    n = col.shape[-1]
    i = np.arange(n)
    d = i[:, None] - i[None, :]
    return np.where(d >= 0, col[..., np.abs(d)], np.conj(col[..., np.abs(d)]))


def exp_corr_mat(a, n, precision=None):
    """Exponential correlation matrices R[i, j] = a^(i-j) for i >= j, conj(a)^(j-i) above the diagonal
    (channel/utils.py:1490-1560); ``a`` [...] complex with |a| < 1 -> [..., n, n]."""
    cd = dtypes[precision or config.precision]["np"]["cdtype"]
    a = np.asarray(a, cd)[..., None]
    assert np.all(np.abs(a) < 1), "The absolute value of the elements of `a` must be smaller than one"
    with np.errstate(invalid="ignore"):
        col = np.power(a, np.arange(n).astype(cd))
    col = np.where(np.isnan(col.real), np.ones_like(col), col)          # 0 ** 0
    return _toeplitz(col).astype(cd)


def one_ring_corr_mat(phi_deg, num_ant, d_h=0.5, sigma_phi_deg=15, precision=None):
    """One-ring model covariance of a uniform linear array (channel/utils.py:1563-1652): angle of arrival ``phi_deg``,
    antenna spacing ``d_h`` wavelengths, angular spread ``sigma_phi_deg`` (<= 15 for the approximation to hold)."""
    import warnings
    p = dtypes[precision or config.precision]["np"]
    if sigma_phi_deg > 15:
        warnings.warn("sigma_phi_deg should be smaller than 15.")
    phi = np.deg2rad(np.asarray(phi_deg, p["rdtype"]))[..., None]
    sigma = np.deg2rad(np.asarray(sigma_phi_deg, p["rdtype"]))[..., None]
    d = (p["rdtype"](2 * np.pi * d_h) * np.arange(num_ant, dtype=p["rdtype"]))
    col = np.exp(1j * d * np.sin(phi)) * np.exp(-0.5 * (sigma * d * np.cos(phi)) ** 2)
    return _toeplitz(col.astype(p["cdtype"])).astype(p["cdtype"])


def _np(a):
    return None if a is None else np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a)


class SpatialCorrelation(Object, ABC):
    """``h_corr = model(h)`` for h [..., M, K] (spatial_correlation.py:13-38)."""

    @abstractmethod
    def __call__(self, h, *args, **kwargs):
        return NotImplemented

    def _apply(self, h, mat):
        """h [..., M, K] on the device (complex64, complex128 with precision="double") -> mat [M K, M K] applied to the
        rx-major vector of every matrix."""
        dbl = self.precision == "double"
        h = _ffi.to_device(h, self.cdtype).contiguous()
        m, k = int(h.shape[-2]), int(h.shape[-1])
        assert mat.shape == (m * k, m * k), "correlation matrices do not fit the channel matrices"
        out = torch.empty_like(h)
        b = h.numel() // (m * k)
        mat_d = _ffi.to_device(np.ascontiguousarray(mat), self.cdtype)               # (held until the launch is queued)
        if b:
            fn = _ffi.lib().samd_spatial_corr_c128 if dbl else _ffi.lib().samd_spatial_corr_c64
            _ffi.check(fn(_ffi.ptr(h), _ffi.ptr(mat_d), b, m, k, 1, _ffi.ptr(out), _ffi.stream()), "spatial correlation")
        return wrap(out)


class KroneckerModel(SpatialCorrelation):
    """``KroneckerModel(r_tx=None, r_rx=None)(h)`` = R_rx^(1/2) h R_tx^(1/2) with the CHOLESKY factors as square roots:
    L_rx h L_tx^H (spatial_correlation.py:41-122).  r_tx [K, K], r_rx [M, M]."""

    def __init__(self, r_tx=None, r_rx=None, precision=None):
        super().__init__(precision=precision)
        self.r_tx, self.r_rx = r_tx, r_rx

    def _set(self, name, value):
        v = _np(value)
        if v is not None:
            if v.ndim != 2:
                raise NotImplementedError("KroneckerModel: batched correlation matrices have no HIP path")
            v = v.astype(dtypes[self.precision]["np"]["cdtype"])
        setattr(self, "_" + name, v)
        setattr(self, "_l_" + name, None if v is None else np.linalg.cholesky(v.astype(np.complex128)))

    r_tx = property(lambda self: self._r_tx, lambda self, v: self._set("r_tx", v))
    r_rx = property(lambda self: self._r_rx, lambda self, v: self._set("r_rx", v))

    def __call__(self, h):
        m, k = int(h.shape[-2]), int(h.shape[-1])
        if self._r_tx is None and self._r_rx is None:
            return h
        l_rx = np.eye(m) if self._l_r_rx is None else self._l_r_rx
        l_tx = np.eye(k) if self._l_r_tx is None else self._l_r_tx
        return self._apply(h, np.kron(l_rx, np.conj(l_tx)).astype(self._np_cdtype))


class PerColumnModel(SpatialCorrelation):
    """``PerColumnModel(r_rx)(h)``: column k of h is correlated with its own matrix, h[:, k] <- L_rx[k] h[:, k]
    (spatial_correlation.py:125-195).  r_rx [K, M, M]."""

    def __init__(self, r_rx, precision=None):
        super().__init__(precision=precision)
        self.r_rx = r_rx

    @property
    def r_rx(self):
        return self._r_rx

    @r_rx.setter
    def r_rx(self, value):
        v = _np(value)
        if v is not None:
            if v.ndim != 3:
                raise NotImplementedError("PerColumnModel: r_rx must be [K, M, M] (one matrix per column; no batch dimension)")
            v = v.astype(dtypes[self.precision]["np"]["cdtype"])
        self._r_rx = v
        self._l_rx = None if v is None else np.linalg.cholesky(v.astype(np.complex128))

    def __call__(self, h):
        if self._r_rx is None:
            return h
        m, k = int(h.shape[-2]), int(h.shape[-1])
        assert self._l_rx.shape == (k, m, m), "r_rx must be [K, M, M]"
        mat = np.zeros((m, k, m, k), np.complex128)
        for kk in range(k):
            mat[:, kk, :, kk] = self._l_rx[kk]
        return self._apply(h, mat.reshape(m * k, m * k).astype(self._np_cdtype))
