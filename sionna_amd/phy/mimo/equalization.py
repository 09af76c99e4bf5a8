"""``lmmse_equalizer(y, h, s, whiten_interference=True)`` - mirror of reference
src/sionna/phy/mimo/equalization.py:101-233 on the HIP kernel ``samd_lmmse_equalizer_c64``."""
import torch

from ... import _ffi
from ..block import wrap


def lmmse_equalizer(y, h, s, whiten_interference=True, precision=None):
    """y [...,M], h [...,M,K], s [...,M,M] -> (x_hat [...,K] complex, no_eff [...,K] float)."""
    if precision not in (None, "single"):
        raise NotImplementedError("lmmse_equalizer: the MI355X kernels implement precision='single' only")
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
    _ffi.check(_ffi.lib().samd_lmmse_equalizer_c64(_ffi.ptr(y), _ffi.ptr(h), _ffi.ptr(s), n, m, k,
                                                  int(bool(whiten_interference)), _ffi.ptr(x_hat),
                                                  _ffi.ptr(no_eff), _ffi.stream()), "lmmse_equalizer")
    return wrap(x_hat), wrap(no_eff)
