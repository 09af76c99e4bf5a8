"""Frequency-domain OFDM channel: ``GenerateOFDMChannel``, ``ApplyOFDMChannel``, ``OFDMChannel``
and ``RayleighBlockFading`` - mirrors of reference src/sionna/phy/channel/
generate_ofdm_channel.py:9-85, apply_ofdm_channel.py:14-80, ofdm_channel.py:13-115,
rayleigh_block_fading.py:10-110."""
import torch

from ... import _ffi
from ..block import Block, Object, wrap
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
        y = torch.empty((b, rx, ra, t, f), dtype=self.cdtype, device=x.device)
        fn = _ffi.lib().samd_apply_ofdm_channel_c128 if self.precision == "double" else _ffi.lib().samd_apply_ofdm_channel_c64
        _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(h), b, rx * ra, tx * ta, t * f, _ffi.ptr(y), _ffi.stream()), "ApplyOFDMChannel")
        if no is not None:
            y = self._awgn(y, no)
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
        h_freq = self._generate_channel(x.shape[0])
        y = self._apply_channel(x, h_freq, no)
        return (y, h_freq) if self._return_channel else y


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
