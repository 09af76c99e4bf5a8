"""``lmmse_equalizer`` / ``zf_equalizer`` / ``mf_equalizer`` - mirrors of reference
src/sionna/phy/mimo/equalization.py:101-233, 235-298, 300-470 on the HIP kernel
``samd_lmmse_equalizer_c64`` (mode 1/0 LMMSE with / without whitening, 2 ZF, 3 MF)."""
import torch

from ... import _ffi
from ..block import wrap


def _equalize(y, h, s, mode, precision, name):
    from ..config import config
    if (precision or config.precision) == "double":
        return _equalize_f64(y, h, s, mode, name)
    y = _ffi.to_device(y, torch.complex64)
    h = _ffi.to_device(h, torch.complex64)
    s = _ffi.to_device(s, torch.complex64)
    m, k = h.shape[-2], h.shape[-1]
    lead = tuple(h.shape[:-2])
    y = torch.broadcast_to(y, lead + (m,)).contiguous()
    s = torch.broadcast_to(s, lead + (m, m)).contiguous()
    n = y.numel() // m
    x_hat = torch.empty(lead + (k,), dtype=torch.complex64, device=y.device)
    no_eff = torch.empty(lead + (k,), dtype=torch.float32, device=y.device)
    h = h.contiguous()                      # (a named tensor: its storage must outlive the launch call)
    _ffi.check(_ffi.lib().samd_lmmse_equalizer_c64(_ffi.ptr(y), _ffi.ptr(h), _ffi.ptr(s), n, m, k, int(mode),
                                                  _ffi.ptr(x_hat), _ffi.ptr(no_eff), _ffi.stream()), name)
    return wrap(x_hat), wrap(no_eff)


def _equalize_f64(y, h, s, mode, name):
    """precision="double" (block.py:25-52): the complex128 kernel samd_lmmse_equalizer_c128 (csrc/f64.hip)."""
    y = _ffi.to_device(y, torch.complex128)
    h = _ffi.to_device(h, torch.complex128)
    s = _ffi.to_device(s, torch.complex128)
    m, k = h.shape[-2], h.shape[-1]
    lead = tuple(h.shape[:-2])
    y = torch.broadcast_to(y, lead + (m,)).contiguous()
    s = torch.broadcast_to(s, lead + (m, m)).contiguous()
    n = y.numel() // m
    x_hat = torch.empty(lead + (k,), dtype=torch.complex128, device=y.device)
    no_eff = torch.empty(lead + (k,), dtype=torch.float64, device=y.device)
    h = h.contiguous()                      # (a named tensor: its storage must outlive the launch call)
    _ffi.check(_ffi.lib().samd_lmmse_equalizer_c128(_ffi.ptr(y), _ffi.ptr(h), _ffi.ptr(s), n, m, k, int(mode),
                                                   _ffi.ptr(x_hat), _ffi.ptr(no_eff), _ffi.stream()), name + "(double)")
    return wrap(x_hat), wrap(no_eff)


def lmmse_matrix(h, s=None, precision=None):
    """h [...,M,K] (and the noise covariance s [...,M,M]; None: white, unit variance) -> the LMMSE equalisation matrices
    G = H^H (H H^H + S)^-1, [...,K,M] (equalization.py:11-99; csrc/mimo_linalg.hip, 1 <= M, K <= 16)."""
    from ..config import config
    dbl = (precision or config.precision) == "double"
    cdt = torch.complex128 if dbl else torch.complex64
    h = _ffi.to_device(h, cdt).contiguous()
    m, k = int(h.shape[-2]), int(h.shape[-1])
    lead = tuple(h.shape[:-2])
    if s is not None:
        s = torch.broadcast_to(_ffi.to_device(s, cdt), lead + (m, m)).contiguous()
    g = torch.empty(lead + (k, m), dtype=cdt, device=h.device)
    fn = _ffi.lib().samd_lmmse_matrix_c128 if dbl else _ffi.lib().samd_lmmse_matrix_c64
    _ffi.check(fn(_ffi.ptr(h), _ffi.ptr(s), h.numel() // (m * k), m, k, _ffi.ptr(g), _ffi.stream()), "lmmse_matrix")
    return wrap(g)


def lmmse_equalizer(y, h, s, whiten_interference=True, precision=None):
    """y [...,M], h [...,M,K], s [...,M,M] -> (x_hat [...,K] complex, no_eff [...,K] float)."""
    return _equalize(y, h, s, int(bool(whiten_interference)), precision, "lmmse_equalizer")


def zf_equalizer(y, h, s, precision=None):
    """Zero-forcing: G = (H^H H)^-1 H^H, x_hat = G y, no_eff = diag(G S G^H)."""
    return _equalize(y, h, s, 2, precision, "zf_equalizer")


def mf_equalizer(y, h, s, precision=None):
    """Matched filter: G = diag(H^H H)^-1 H^H, no_eff = |diag((I - G H)(I - G H)^H + G S G^H)|."""
    return _equalize(y, h, s, 3, precision, "mf_equalizer")
