"""``OFDMModulator`` / ``OFDMDemodulator`` - mirrors of reference src/sionna/phy/ofdm/
modulator.py:13-124 and demodulator.py:14-203.  The transform is rocFFT's, reached through
``samd_ofdm_modulate_c64`` / ``samd_ofdm_demodulate_c64`` (``_c128`` with ``precision="double"``)."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, wrap


def _cp_tables(cyclic_prefix_length, fft_size, num_ofdm_symbols):
    cp = np.asarray(cyclic_prefix_length)
    if cp.ndim > 1:
        raise ValueError("`cyclic_prefix_length` must be of rank 0 or 1")
    if not np.issubdtype(cp.dtype, np.integer) and not np.all(cp == np.round(cp)):
        raise ValueError("`cyclic_prefix_length` must be an integer")
    cp = cp.astype(np.int64)
    if np.any(cp < 0):
        raise ValueError("`cyclic_prefix_length` must be nonnegative.")
    if cp.ndim == 1 and cp.shape[0] != num_ofdm_symbols:
        raise ValueError("shape(inputs)[-2] must match shape(cyclic_prefix_length)[0]")
    if np.any(cp > fft_size):
        raise ValueError("shape(inputs)[-1] must not be smaller than `cylic_prefix_length`")
    cpv = np.full(num_ofdm_symbols, int(cp), np.int32) if cp.ndim == 0 else cp.astype(np.int32)
    off = np.concatenate([[0], np.cumsum(cpv.astype(np.int64) + fft_size)]).astype(np.int64)
    return cpv, off[:-1].astype(np.int32), int(off[-1])


class OFDMModulator(Block):
    """``OFDMModulator(cyclic_prefix_length=0)(x)``: [..., num_ofdm_symbols, fft_size] ->
    [..., num_ofdm_symbols*fft_size + sum(cyclic_prefix_length)]."""

    def __init__(self, cyclic_prefix_length=0, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self.cyclic_prefix_length = cyclic_prefix_length
        self._tables = {}

    @property
    def cyclic_prefix_length(self):
        return self._cyclic_prefix_length

    @cyclic_prefix_length.setter
    def cyclic_prefix_length(self, value):
        v = np.asarray(value)
        if v.ndim > 1:
            raise ValueError("`cyclic_prefix_length` must be of rank 0 or 1")
        if np.any(v < 0):
            raise ValueError("`cyclic_prefix_length` must be nonnegative.")
        self._cyclic_prefix_length = v.astype(np.int32)

    def call(self, inputs):
        x = _ffi.to_device(inputs, self.cdtype)
        nsym, n = int(x.shape[-2]), int(x.shape[-1])
        key = (nsym, n)
        if key not in self._tables:
            cp, off, out_len = _cp_tables(self._cyclic_prefix_length, n, nsym)
            self._tables[key] = (_ffi.to_device(cp, torch.int32), _ffi.to_device(off, torch.int32), int(cp.max()),
                                 out_len)
        cp, off, max_cp, out_len = self._tables[key]
        lead = tuple(x.shape[:-2])
        rows = int(np.prod(lead)) if lead else 1
        work = torch.empty((rows, nsym, n), dtype=self.cdtype, device=x.device)
        out = torch.empty(lead + (out_len,), dtype=self.cdtype, device=x.device)
        fn = _ffi.lib().samd_ofdm_modulate_c128 if self.precision == "double" else _ffi.lib().samd_ofdm_modulate_c64
        _ffi.check(fn(_ffi.ptr(x), rows, nsym, n, _ffi.ptr(cp), _ffi.ptr(off), max_cp,
                      out_len, _ffi.ptr(work), _ffi.ptr(out), _ffi.stream()), "OFDMModulator")
        return wrap(out)


class OFDMDemodulator(Block):
    """``OFDMDemodulator(fft_size, l_min, cyclic_prefix_length=0)(y)``: [..., num_time_samples]
    -> [..., num_ofdm_symbols, fft_size]; samples after the last complete symbol are dropped."""

    def __init__(self, fft_size, l_min, cyclic_prefix_length=0, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert fft_size > 0, "`fft_size` must be positive."
        assert l_min <= 0, "l_min must be nonpositive."
        self._fft_size, self._l_min = int(fft_size), int(l_min)
        v = np.asarray(cyclic_prefix_length)
        if v.ndim > 1:
            raise ValueError("`cyclic_prefix_length` must be of rank 0 or 1")
        if np.any(v < 0):
            raise ValueError("`cyclic_prefix_length` must be nonnegative.")
        self._cyclic_prefix_length = v.astype(np.int32)
        self._tables = {}

    fft_size = property(lambda self: self._fft_size)
    l_min = property(lambda self: self._l_min)
    cyclic_prefix_length = property(lambda self: self._cyclic_prefix_length)

    def call(self, inputs):
        y = _ffi.to_device(inputs, self.cdtype)
        in_len, n = int(y.shape[-1]), self._fft_size
        if in_len not in self._tables:
            cp0 = self._cyclic_prefix_length
            nsym = in_len // (n + int(cp0)) if cp0.ndim == 0 else int(cp0.shape[0])
            if nsym == 0:
                raise ValueError("input shorter than one OFDM symbol")
            cp, off, need = _cp_tables(cp0, n, nsym)
            if need > in_len:
                raise ValueError("input shorter than the configured OFDM symbols")
            self._tables[in_len] = (_ffi.to_device(cp, torch.int32), _ffi.to_device(off, torch.int32), nsym)
        cp, off, nsym = self._tables[in_len]
        lead = tuple(y.shape[:-1])
        rows = int(np.prod(lead)) if lead else 1
        work = torch.empty((rows, nsym, n), dtype=self.cdtype, device=y.device)
        out = torch.empty(lead + (nsym, n), dtype=self.cdtype, device=y.device)
        fn = _ffi.lib().samd_ofdm_demodulate_c128 if self.precision == "double" else _ffi.lib().samd_ofdm_demodulate_c64
        _ffi.check(fn(_ffi.ptr(y), rows, in_len, nsym, n, _ffi.ptr(cp), _ffi.ptr(off), self._l_min, _ffi.ptr(work),
                      _ffi.ptr(out), _ffi.stream()), "OFDMDemodulator")
        return wrap(out)
