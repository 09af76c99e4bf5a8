"""Flat-fading MIMO channel - mirrors of ``GenerateFlatFadingChannel``, ``ApplyFlatFadingChannel`` and
``FlatFadingChannel`` (reference src/sionna/phy/channel/flat_fading_channel.py:14-246): i.i.d. CN(0,1)
channel matrices on the Philox stream and y = H x (+ AWGN) through ``samd_apply_ofdm_channel_c64``
(one "resource element" per batch item); spatial correlation models (spatial_correlation.py) through
``samd_spatial_corr_c64``."""
import torch

from ... import _ffi
from ..block import Block, Object, wrap
from .awgn import AWGN


class GenerateFlatFadingChannel(Object):
    """``GenerateFlatFadingChannel(num_tx_ant, num_rx_ant, spatial_corr=None)(batch_size)`` ->
    h [batch_size, num_rx_ant, num_tx_ant]."""

    def __init__(self, num_tx_ant, num_rx_ant, spatial_corr=None, precision=None, **kwargs):
        super().__init__(precision=precision)
        self._num_tx_ant, self._num_rx_ant = int(num_tx_ant), int(num_rx_ant)
        self.spatial_corr = spatial_corr

    @property
    def spatial_corr(self):
        return self._spatial_corr

    @spatial_corr.setter
    def spatial_corr(self, value):
        from .spatial_correlation import SpatialCorrelation
        if value is not None and not isinstance(value, SpatialCorrelation):
            raise TypeError("spatial_corr must be a SpatialCorrelation (KroneckerModel, PerColumnModel) or None")
        self._spatial_corr = value

    def __call__(self, batch_size):
        """flat_fading_channel.py:63-76: i.i.d. CN(0, 1) matrices, then the correlation model."""
        from ..utils.misc import complex_normal
        h = complex_normal([int(batch_size), self._num_rx_ant, self._num_tx_ant], 1.0, precision=self.precision)
        return h if self._spatial_corr is None else self._spatial_corr(h)


class ApplyFlatFadingChannel(Block):
    """``ApplyFlatFadingChannel()(x [batch, num_tx_ant], h [batch, num_rx_ant, num_tx_ant], no=None)``
    -> y [batch, num_rx_ant]."""

    def __init__(self, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._awgn = AWGN(precision=self.precision)

    def call(self, x, h, no=None):
        x = _ffi.to_device(x, self.cdtype)
        h = _ffi.to_device(h, self.cdtype)
        b, rx, tx = h.shape
        x = torch.broadcast_to(x, (b, tx)).contiguous()
        y = torch.empty((b, rx), dtype=self.cdtype, device=x.device)
        if b:
            h = h.contiguous()                      # (a named tensor: its storage must outlive the launch call)
            fn = _ffi.lib().samd_apply_ofdm_channel_c128 if self.precision == "double" else _ffi.lib().samd_apply_ofdm_channel_c64
            _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(h), b, rx, tx, 1, _ffi.ptr(y), _ffi.stream()), "ApplyFlatFadingChannel")
        if no is not None:
            y = self._awgn(y, no)
        return wrap(y)


class FlatFadingChannel(Block):
    """``FlatFadingChannel(num_tx_ant, num_rx_ant, spatial_corr=None, return_channel=False)(x, no=None)``
    -> y or (y, h)."""

    def __init__(self, num_tx_ant, num_rx_ant, spatial_corr=None, return_channel=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._num_tx_ant, self._num_rx_ant, self._return_channel = int(num_tx_ant), int(num_rx_ant), return_channel
        self._gen_chn = GenerateFlatFadingChannel(num_tx_ant, num_rx_ant, spatial_corr, precision=self.precision)
        self._app_chn = ApplyFlatFadingChannel(precision=self.precision)

    spatial_corr = property(lambda self: self._gen_chn.spatial_corr, lambda self, v: setattr(self._gen_chn, "spatial_corr", v))
    generate = property(lambda self: self._gen_chn)
    apply = property(lambda self: self._app_chn)

    def call(self, x, no=None):
        h = self._gen_chn(x.shape[0])
        y = self._app_chn(x, h, no)
        return (y, h) if self._return_channel else y
