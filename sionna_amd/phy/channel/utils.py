"""Channel utilities of the hot path - mirror of reference src/sionna/phy/channel/utils.py
(``subcarrier_frequencies`` :15-66, ``cir_to_ofdm_channel`` :180-253)."""
import numpy as np
import torch

from ... import _ffi
from ..block import wrap
from ..config import config, dtypes


def subcarrier_frequencies(num_subcarriers, subcarrier_spacing, precision=None):
    """Baseband frequencies of the subcarriers, DC at index num_subcarriers//2 (utils.py:15-66)."""
    p = config.precision if precision is None else precision
    f = dtypes[p]["np"]["rdtype"]
    start = -(num_subcarriers // 2)
    limit = num_subcarriers // 2 + (num_subcarriers % 2)
    return (np.arange(start, limit, dtype=f) * f(subcarrier_spacing)).astype(f)


def cir_to_ofdm_channel(frequencies, a, tau, normalize=False):
    """h_f[b,rx,ra,tx,ta,t,f] = sum_p a[b,rx,ra,tx,ta,p,t] exp(-j 2 pi f tau[b,rx,tx,p])
    (+ optional normalisation to unit mean energy per (b,rx,tx)) - utils.py:180-253."""
    # the precision follows the coefficients like in the reference (real_dtype = tau.dtype, utils.py:228): complex128 taps
    # run the float64 kernel (csrc/f64_ofdm.hip)
    dbl = str(getattr(a, "dtype", "")).endswith("complex128")
    cdt, rdt = (torch.complex128, torch.float64) if dbl else (torch.complex64, torch.float32)
    a = _ffi.to_device(a, cdt)
    tau = _ffi.to_device(tau, rdt)
    fr = _ffi.to_device(np.asarray(frequencies.cpu() if isinstance(frequencies, torch.Tensor) else frequencies,
                                   np.float64 if dbl else np.float32), rdt)
    if tau.dim() != 4:
        raise NotImplementedError("cir_to_ofdm_channel: per-antenna delays (rank-6 tau) are outside the hot path")
    b, rx, ra, tx, ta, p, t = a.shape
    assert tuple(tau.shape) == (b, rx, tx, p), "tau must have shape [batch, num_rx, num_tx, num_paths]"
    h = torch.empty((b, rx, ra, tx, ta, t, fr.numel()), dtype=cdt, device=a.device)
    fn = _ffi.lib().samd_cir_to_ofdm_c128 if dbl else _ffi.lib().samd_cir_to_ofdm_c64
    _ffi.check(fn(_ffi.ptr(a), _ffi.ptr(tau), _ffi.ptr(fr), b, rx, ra, tx, ta, p, t, fr.numel(), int(bool(normalize)),
                  _ffi.ptr(h), _ffi.stream()), "cir_to_ofdm_channel")
    return wrap(h)
