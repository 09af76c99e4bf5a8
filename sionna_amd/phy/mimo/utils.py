"""``whiten_channel`` - mirror of reference src/sionna/phy/mimo/utils.py:292-356 on csrc/mimo_linalg.hip, and the
complex <-> real-valued representation helpers (:11-289: index plumbing on device tensors)."""
import torch

from ... import _ffi
from ..block import wrap
from ..utils.linalg import _as_complex, _back


def whiten_channel(y, h, s, return_s=True):
    """y [...,M], h [...,M,K], s [...,M,M] = L L^H -> (L^-1 y, L^-1 h[, I_M]); real inputs give real outputs, the precision
    follows ``s``."""
    s_c, real, dbl = _as_complex(s)
    cdt = s_c.dtype
    h_c = _ffi.to_device(h, cdt).contiguous()
    m, k = int(h_c.shape[-2]), int(h_c.shape[-1])
    lead = tuple(h_c.shape[:-2])
    y_c = torch.broadcast_to(_ffi.to_device(y, cdt), lead + (m,)).contiguous()
    s_c = torch.broadcast_to(s_c, lead + (m, m)).contiguous()
    yw, hw = torch.empty_like(y_c), torch.empty_like(h_c)
    fn = _ffi.lib().samd_whiten_channel_c128 if dbl else _ffi.lib().samd_whiten_channel_c64
    _ffi.check(fn(_ffi.ptr(y_c), _ffi.ptr(h_c), _ffi.ptr(s_c), h_c.numel() // (m * k), m, k, _ffi.ptr(yw), _ffi.ptr(hw),
                  _ffi.stream()), "whiten_channel")
    if not return_s:
        return _back(yw, real), _back(hw, real)
    sw = torch.eye(m, dtype=cdt, device=s_c.device).expand(lead + (m, m)).contiguous()
    return _back(yw, real), _back(hw, real), _back(sw, real)


def _dev(z):
    """device tensor of the input's own dtype"""
    import numpy as np
    dt = z.dtype if isinstance(z, torch.Tensor) else torch.from_numpy(np.zeros(0, np.asarray(z).dtype)).dtype
    return _ffi.to_device(z, dt)


def complex2real_vector(z):
    """[...,M] complex -> [...,2M] real: real parts, then imaginary parts (utils.py:11-33)."""
    z = _dev(z)
    return wrap(torch.cat([z.real, z.imag], dim=-1))


def real2complex_vector(z):
    """[...,2M] real -> [...,M] complex (utils.py:36-57)."""
    z = _dev(z)
    m = z.shape[-1] // 2
    return wrap(torch.complex(z[..., :m].contiguous(), z[..., m:].contiguous()))


def complex2real_matrix(z):
    """[...,M,K] complex -> [...,2M,2K] real: [[Re, -Im], [Im, Re]] (utils.py:60-90)."""
    z = _dev(z)
    return wrap(torch.cat([torch.cat([z.real, -z.imag], dim=-1), torch.cat([z.imag, z.real], dim=-1)], dim=-2))


def real2complex_matrix(z):
    """[...,2M,2K] real -> [...,M,K] complex (utils.py:93-125)."""
    z = _dev(z)
    m, k = z.shape[-2] // 2, z.shape[-1] // 2
    return wrap(torch.complex(z[..., :m, :k].contiguous(), z[..., m:, :k].contiguous()))


def complex2real_covariance(r):
    """[...,M,M] complex covariance -> [...,2M,2M] real covariance of the stacked vector: 1/2 [[Re, -Im], [Im, Re]]
    (utils.py:128-157)."""
    r = _dev(r)
    q = torch.cat([torch.cat([r.real, -r.imag], dim=-1), torch.cat([r.imag, r.real], dim=-1)], dim=-2)
    return wrap(q * 0.5)


def real2complex_covariance(q):
    """[...,2M,2M] real -> [...,M,M] complex covariance (utils.py:160-191)."""
    q = _dev(q)
    m = q.shape[-1] // 2
    return wrap(torch.complex((2 * q[..., :m, :m]).contiguous(), (2 * q[..., m:, :m]).contiguous()))


def complex2real_channel(y, h, s):
    """the real-valued equivalent (y, h, s) of a complex MIMO channel (utils.py:194-241)."""
    return complex2real_vector(y), complex2real_matrix(h), complex2real_covariance(s)


def real2complex_channel(y, h, s):
    """inverse of ``complex2real_channel`` (utils.py:244-289)."""
    return real2complex_vector(y), real2complex_matrix(h), real2complex_covariance(s)
