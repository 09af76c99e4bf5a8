"""Frequency-domain OFDM channel: ``GenerateOFDMChannel``, ``ApplyOFDMChannel``, ``OFDMChannel``
and ``RayleighBlockFading`` - mirrors of reference src/sionna/phy/channel/
generate_ofdm_channel.py:9-85, apply_ofdm_channel.py:14-80, ofdm_channel.py:13-115,
rayleigh_block_fading.py:10-110."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, Object, Pending, defer, wrap
from ..config import config
from .awgn import AWGN
from .utils import subcarrier_frequencies, cir_to_ofdm_channel


class GenerateOFDMChannel(Object):
    """``GenerateOFDMChannel(channel_model, resource_grid, normalize_channel=False)(batch_size)`` ->
    h_freq [batch, num_rx, num_rx_ant, num_tx, num_tx_ant, num_ofdm_symbols, fft_size]."""

    def __init__(self, channel_model, resource_grid, normalize_channel=False, precision=None, **kwargs):
        super().__init__(precision=precision)
        self._cir_sampler = channel_model
        self._num_ofdm_symbols = resource_grid.num_ofdm_symbols
        self._normalize_channel = normalize_channel
        self._sampling_frequency = 1. / resource_grid.ofdm_symbol_duration
        self._frequencies = subcarrier_frequencies(resource_grid.fft_size, resource_grid.subcarrier_spacing,
                                                   self.precision)

    def __call__(self, batch_size=None):
        h, tau = self._cir_sampler(batch_size, self._num_ofdm_symbols, self._sampling_frequency)
        return cir_to_ofdm_channel(self._frequencies, h, tau, self._normalize_channel)


class ApplyOFDMChannel(Block):
    """y[b,rx,ra,t,f] = sum_{tx,ta} h_freq * x (+ AWGN of variance ``no``)."""

    def __init__(self, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._awgn = AWGN(precision=self.precision)

    def call(self, x, h_freq, no=None):
        x = _ffi.to_device(x, self.cdtype)
        h = _ffi.to_device(h_freq, self.cdtype)
        b, rx, ra, tx, ta, t, f = h.shape
        assert tuple(x.shape) == (b, tx, ta, t, f), "x must have shape [batch, num_tx, num_tx_ant, num_ofdm_symbols, fft_size]"
        y = self._apply(x, h)
        if no is not None:
            y = self._awgn(y, no)
        return y

    def _apply(self, x, h):
        b, rx, ra, tx, ta, t, f = h.shape
        y = torch.empty((b, rx, ra, t, f), dtype=self.cdtype, device=x.device)
        fn = _ffi.lib().samd_apply_ofdm_channel_c128 if self.precision == "double" else _ffi.lib().samd_apply_ofdm_channel_c64
        _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(h), b, rx * ra, tx * ta, t * f, _ffi.ptr(y), _ffi.stream()), "ApplyOFDMChannel")
        return y


class OFDMChannel(Block):
    """``OFDMChannel(channel_model, resource_grid, normalize_channel=False, return_channel=False)``
    ``(x, no=None)`` -> y or (y, h_freq)."""

    def __init__(self, channel_model, resource_grid, normalize_channel=False, return_channel=False,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._return_channel = return_channel
        self._generate_channel = GenerateOFDMChannel(channel_model, resource_grid, normalize_channel, self.precision)
        self._apply_channel = ApplyOFDMChannel(self.precision)

    def call(self, x, no=None):
        fused = self._call_fused(x, no)
        if fused is not None:
            return fused
        h_freq = self._generate_channel(x.shape[0])
        y = self._apply_channel(x, h_freq, no)
        return (y, h_freq) if self._return_channel else y

    def _call_fused(self, x, no):
        """cir_to_ofdm_channel + ApplyOFDMChannel + AWGN in one launch (samd_ofdm_channel_fused_c64: the same bits, the
        frequency response - 558 MB at config C4 - never written): one transmitter, single precision, scalar ``no``.  A
        returned ``h_freq`` is DEFERRED: allocated, and filled by cir_to_ofdm_channel only if somebody reads it (a receiver
        that estimates the channel never does).  None: the separate blocks run."""
        gen = self._generate_channel
        if self.precision != "single" or not isinstance(x, torch.Tensor) or x.dim() != 5 or x.shape[1] != 1:
            return None
        if no is not None and int(np.size(no.detach().cpu() if isinstance(no, torch.Tensor) else no)) != 1:
            return None                                     # per-element noise variances: the separate blocks
        x = _ffi.to_device(x, self.cdtype)
        a, tau = gen._cir_sampler(x.shape[0], gen._num_ofdm_symbols, gen._sampling_frequency)
        a_t, tau_t = _ffi.to_device(a, self.cdtype), _ffi.to_device(tau, self.rdtype)
        ok = a_t.dim() == 7 and tau_t.dim() == 4 and a_t.shape[3] == 1 and a_t.shape[4] == x.shape[2] and a_t.shape[6] == x.shape[3]
        fr = _ffi.to_device(gen._frequencies, self.rdtype)
        norm = gen._normalize_channel

        def separate():                                     # (the channel realisation is drawn already: finish without the fusion)
            h = cir_to_ofdm_channel(fr, a_t, tau_t, norm)
            yy = self._apply_channel(x, h, no)
            return (yy, h) if self._return_channel else yy
        if not ok or fr.numel() != x.shape[4]:
            return separate()
        b, rx, ra, tx, ta, p, t = a_t.shape
        y = torch.empty((b, rx, ra, t, fr.numel()), dtype=self.cdtype, device=x.device)
        no_t = None if no is None else _ffi.to_device(no, self.rdtype).reshape(1)
        rng = config.rng
        call_id = rng.next_call() if no is not None else 0
        # The noise is NOT added inside the launch although the entry can do it (bit-identical): Philox + Box-Muller per element
        # on top of the 48 staged result registers of a thread ran at 5 waves per SIMD with spills - 747 us against 343 us for
        # the noise-free launch at config C4 (profiles/r06g_kernel_stats_c4.txt); awgn_kernel in place on y costs ~150 us.
        rc = _ffi.lib().samd_ofdm_channel_fused_c64(_ffi.ptr(a_t), _ffi.ptr(tau_t), _ffi.ptr(fr), _ffi.ptr(x), None, rng.seed, call_id,
                                                    b, rx, ra, tx, ta, p, t, fr.numel(), int(bool(norm)), _ffi.ptr(y), _ffi.stream())
        if rc == _ffi.ERR_UNSUPPORTED:
            h = cir_to_ofdm_channel(fr, a_t, tau_t, norm)
            yy = self._apply_channel._apply(x, h)
            if no is not None:                              # the noise call id is taken: the same stream as the fused launch
                yy = self._apply_channel._awgn._add(yy, no_t, call_id)
            return (wrap(yy), h) if self._return_channel else wrap(yy)
        _ffi.check(rc, "OFDMChannel")
        if no is not None:
            y = self._apply_channel._awgn._add(y, no_t, call_id, out=y)
        if not self._return_channel:
            return wrap(y)
        h = torch.empty((b, rx, ra, tx, ta, t, fr.numel()), dtype=self.cdtype, device=x.device)

        def fill(plain):
            _ffi.check(_ffi.lib().samd_cir_to_ofdm_c64(_ffi.ptr(a_t), _ffi.ptr(tau_t), _ffi.ptr(fr), b, rx, ra, tx, ta, p, t, fr.numel(),
                                                       int(bool(norm)), _ffi.ptr(plain), _ffi.stream()), "cir_to_ofdm_channel")
        return wrap(y), defer(h, Pending("cir_to_ofdm_channel", fill, guard=(a_t, tau_t)))


class RayleighBlockFading(Object):
    """i.i.d. CN(0,1) single-tap block fading: a [batch, num_rx, num_rx_ant, num_tx, num_tx_ant, 1,
    num_time_steps] (constant over time), tau = 0 (rayleigh_block_fading.py:62-110)."""

    def __init__(self, num_rx, num_rx_ant, num_tx, num_tx_ant, precision=None, **kwargs):
        super().__init__(precision=precision)
        self.num_rx, self.num_rx_ant, self.num_tx, self.num_tx_ant = num_rx, num_rx_ant, num_tx, num_tx_ant

    def __call__(self, batch_size, num_time_steps, sampling_frequency=None):
        from ..utils.misc import complex_normal
        shape = [int(batch_size), self.num_rx, self.num_rx_ant, self.num_tx, self.num_tx_ant, 1, 1]
        h = complex_normal(shape, 1.0, precision=self.precision).as_subclass(torch.Tensor)
        h = h.expand(*shape[:-1], int(num_time_steps)).contiguous()
        tau = torch.zeros((int(batch_size), self.num_rx, self.num_tx, 1), dtype=self.rdtype, device=h.device)
        return wrap(h), wrap(tau)
