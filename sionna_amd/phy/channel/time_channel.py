"""Time-domain channel: ``cir_to_time_channel``, ``time_lag_discrete_time_channel``,
``GenerateTimeChannel``, ``ApplyTimeChannel``, ``TimeChannel`` - mirrors of reference
src/sionna/phy/channel/utils.py:121-178 and :256-349, generate_time_channel.py:9-100,
apply_time_channel.py:14-137, time_channel.py:13-163."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, Object, wrap
from .awgn import AWGN


def time_lag_discrete_time_channel(bandwidth, maximum_delay_spread=3e-6):
    """Recommended (l_min, l_max) = (-6, ceil(W * delay_spread) + 6) (utils.py:121-178)."""
    return -6, int(np.ceil(maximum_delay_spread * bandwidth)) + 6


def cir_to_time_channel(bandwidth, a, tau, l_min, l_max, normalize=False, _defer_norm=False):
    """h[b,rx,ra,tx,ta,t,l] = sum_p a[b,rx,ra,tx,ta,p,t] sinc(l - W tau[b,rx,tx,p]),
    l = l_min..l_max (utils.py:256-349).  ``_defer_norm`` (internal, TimeChannel): return (h un-normalised,
    scale [b,rx,tx]) so that the normalisation is applied to the received signal instead of in a second
    pass over h."""
    dbl = str(getattr(a, "dtype", "")).endswith("complex128")          # the precision follows the coefficients (utils.py:309)
    a = _ffi.to_device(a, torch.complex128 if dbl else torch.complex64)
    tau = _ffi.to_device(tau, torch.float64 if dbl else torch.float32)
    if tau.dim() != 4:
        raise NotImplementedError("cir_to_time_channel: per-antenna delays (rank-6 tau) are outside the hot path")
    b, rx, ra, tx, ta, p, t = a.shape
    assert tuple(tau.shape) == (b, rx, tx, p), "tau must have shape [batch, num_rx, num_tx, num_paths]"
    l_min, l_max = int(l_min), int(l_max)
    if dbl:                                                            # float64 kernels (csrc/f64_time.hip): normalised in place
        h = torch.empty((b, rx, ra, tx, ta, t, l_max - l_min + 1), dtype=torch.complex128, device=a.device)
        _ffi.check(_ffi.lib().samd_cir_to_time_c128(float(bandwidth), _ffi.ptr(a), _ffi.ptr(tau), l_min, l_max, b, rx, ra, tx, ta, p, t,
                                                    int(bool(normalize)), _ffi.ptr(h), _ffi.stream()), "cir_to_time_channel")
        return (wrap(h), None) if _defer_norm else wrap(h)
    h = torch.empty((b, rx, ra, tx, ta, t, l_max - l_min + 1), dtype=torch.complex64, device=a.device)
    scale = torch.empty((b, rx, tx), dtype=torch.float32, device=a.device) if (_defer_norm and normalize) else None
    _ffi.check(_ffi.lib().samd_cir_to_time_c64(float(bandwidth), _ffi.ptr(a), _ffi.ptr(tau), l_min, l_max, b, rx, ra,
                                               tx, ta, p, t, int(bool(normalize)), _ffi.ptr(h), _ffi.ptr(scale),
                                               _ffi.stream()),
               "cir_to_time_channel")
    return (wrap(h), scale) if _defer_norm else wrap(h)


class GenerateTimeChannel(Object):
    """``GenerateTimeChannel(channel_model, bandwidth, num_time_samples, l_min, l_max,
    normalize_channel=False)(batch_size)`` -> h_time [batch, num_rx, num_rx_ant, num_tx,
    num_tx_ant, num_time_samples + l_max - l_min, l_max - l_min + 1]."""

    def __init__(self, channel_model, bandwidth, num_time_samples, l_min, l_max, normalize_channel=False,
                 precision=None, **kwargs):
        super().__init__(precision=precision)
        self._cir_sampler = channel_model
        self._l_min, self._l_max = int(l_min), int(l_max)
        self._l_tot = self._l_max - self._l_min + 1
        self._bandwidth = bandwidth
        self._num_time_steps = int(num_time_samples)
        self._normalize_channel = normalize_channel

    def __call__(self, batch_size=None, _defer_norm=False):
        h, tau = self._cir_sampler(batch_size, self._num_time_steps + self._l_tot - 1, self._bandwidth)
        return cir_to_time_channel(self._bandwidth, h, tau, self._l_min, self._l_max, self._normalize_channel,
                                   _defer_norm=_defer_norm)


class ApplyTimeChannel(Block):
    """y[b,rx,ra,t] = sum_{tx,ta} sum_l h_time[...,t,l] x[b,tx,ta,t-l] (+ AWGN), t = 0..N+L-2."""

    def __init__(self, num_time_samples, l_tot, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._num_time_samples, self._l_tot = int(num_time_samples), int(l_tot)
        self._awgn = AWGN(precision=self.precision)

    def call(self, x, h_time, no=None, _link_scale=None):
        dbl = self.precision == "double"
        x = _ffi.to_device(x, self.cdtype)
        h = _ffi.to_device(h_time, self.cdtype)
        b, rx, ra, tx, ta, tout, l = h.shape
        tn = self._num_time_samples
        assert l == self._l_tot and tout == tn + l - 1, "h_time must have num_time_samples + l_tot - 1 time steps"
        assert tuple(x.shape) == (b, tx, ta, tn), "x must have shape [batch, num_tx, num_tx_ant, num_time_samples]"
        y = torch.empty((b, rx, ra, tout), dtype=self.cdtype, device=x.device)
        if dbl:
            assert _link_scale is None
            _ffi.check(_ffi.lib().samd_apply_time_channel_c128(_ffi.ptr(x), _ffi.ptr(h), b, rx, ra, tx, ta, tn, l, _ffi.ptr(y),
                                                               _ffi.stream()), "ApplyTimeChannel")
        else:
            _ffi.check(_ffi.lib().samd_apply_time_channel_c64(_ffi.ptr(x), _ffi.ptr(h), _ffi.ptr(_link_scale), b, rx, ra, tx,
                                                              ta, tn, l, _ffi.ptr(y), _ffi.stream()), "ApplyTimeChannel")
        if no is not None:
            y = self._awgn(y, no)
        return wrap(y)


class TimeChannel(Block):
    """``TimeChannel(channel_model, bandwidth, num_time_samples, maximum_delay_spread=3e-6,
    l_min=None, l_max=None, normalize_channel=False, return_channel=False)(x, no=None)``."""

    def __init__(self, channel_model, bandwidth, num_time_samples, maximum_delay_spread=3e-6, l_min=None,
                 l_max=None, normalize_channel=False, return_channel=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        l_min_default, l_max_default = time_lag_discrete_time_channel(bandwidth, maximum_delay_spread)
        self._l_min = l_min_default if l_min is None else int(l_min)
        self._l_max = l_max_default if l_max is None else int(l_max)
        self._l_tot = self._l_max - self._l_min + 1
        self._return_channel = return_channel
        self._generate_channel = GenerateTimeChannel(channel_model, bandwidth, num_time_samples, self._l_min,
                                                     self._l_max, normalize_channel, self.precision)
        self._apply_channel = ApplyTimeChannel(num_time_samples, self._l_tot, precision=self.precision)

    def call(self, x, no=None):
        if not self._return_channel:
            # the channel itself is not handed out: its normalisation factor goes onto the received signal
            # instead of a second pass over h_time (same result up to float32 rounding)
            h_time, scale = self._generate_channel(x.shape[0], _defer_norm=True)
            return self._apply_channel(x, h_time, no, _link_scale=scale)
        h_time = self._generate_channel(x.shape[0])
        y = self._apply_channel(x, h_time, no)
        return y, h_time
