"""``RowColumnInterleaver`` and ``Deinterleaver`` - mirrors of reference
src/sionna/phy/fec/interleaving.py:12-195, 500-596 (a gather along one axis through
``samd_gather3``).  ``RandomInterleaver`` / ``Turbo3GPPInterleaver`` are outside the hot path."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, wrap


class RowColumnInterleaver(Block):
    """``RowColumnInterleaver(row_depth, axis=-1, inverse=False)(x, inverse=None)``."""

    def __init__(self, row_depth, axis=-1, inverse=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(axis, int):
            raise TypeError("axis must be int.")
        self._axis = axis
        if not isinstance(row_depth, int):
            raise TypeError("row_depth must be int.")
        self._row_depth = row_depth
        if not isinstance(inverse, bool):
            raise TypeError("inverse must be bool.")
        self._inverse = inverse
        self._keep_state = True
        self._perm_seq = self._perm_seq_inv = None
        self._dev = {}

    axis = property(lambda self: self._axis)
    row_depth = property(lambda self: self._row_depth)
    perm_seq = property(lambda self: self._perm_seq)
    perm_seq_inv = property(lambda self: self._perm_seq_inv)
    keep_state = property(lambda self: True)

    @staticmethod
    def _generate_perm_rc(n_seq, r_depth):
        """interleaving.py:111-144: write row-wise into rows of length ceil-padded n/r_depth ... read
        column-wise, dropping the filler positions."""
        n = int(np.ceil(n_seq / r_depth) * r_depth)
        ind = np.arange(n, dtype=np.int32).reshape(n // r_depth, -1).T.reshape(-1)
        perm = ind[ind < n_seq]
        return perm.astype(np.int32), np.argsort(perm).astype(np.int32)

    def build(self, input_shape, **kwargs):
        if self._axis >= len(input_shape) or self._axis < -len(input_shape):
            raise ValueError("Axis does match input shape")
        self._perm_seq, self._perm_seq_inv = self._generate_perm_rc(int(input_shape[self._axis]), self._row_depth)
        self._dev = {}

    def _device_perm(self, inverse):
        if inverse not in self._dev:
            p = self._perm_seq_inv if inverse else self._perm_seq
            self._dev[inverse] = (_ffi.to_device(p, torch.int32), _ffi.to_device(np.zeros(1, np.int32), torch.int32))
        return self._dev[inverse]

    def call(self, x, /, *, inverse=None, **kwargs):
        self._require_single()
        x = _ffi.to_device(x, torch.float32)
        if self._perm_seq is None or x.shape[self._axis] != self._perm_seq.shape[0]:
            self.build(tuple(x.shape))
        inverse = self._inverse if inverse is None else bool(inverse)
        perm, zero = self._device_perm(inverse)
        xm = torch.movedim(x, self._axis, -1).contiguous()          # layout plumbing only
        n = xm.shape[-1]
        rows = xm.numel() // n if n else 0
        out = torch.empty_like(xm)
        if rows:
            _ffi.check(_ffi.lib().samd_gather3(_ffi.ptr(xm), _ffi.ptr(zero), _ffi.ptr(perm), rows, 1, n, 1, n, 1,
                                               _ffi.ptr(out), _ffi.stream()), "RowColumnInterleaver")
        return wrap(torch.movedim(out, -1, self._axis).contiguous())


class Deinterleaver(Block):
    """``Deinterleaver(interleaver)(x, seed=None)`` = ``interleaver(x, inverse=True)``."""

    def __init__(self, interleaver, precision=None, **kwargs):
        if not isinstance(interleaver, RowColumnInterleaver):
            raise ValueError("interleaver is not a valid interleaver instance.")
        self._interleaver = interleaver
        super().__init__(precision=interleaver.precision if precision is None else precision, **kwargs)

    interleaver = property(lambda self: self._interleaver)

    def call(self, x, seed=None):
        return self._interleaver(x, seed=seed, inverse=True)
