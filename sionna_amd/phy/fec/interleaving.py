"""``RowColumnInterleaver`` and ``Deinterleaver`` - mirrors of reference
src/sionna/phy/fec/interleaving.py:12-195, 500-596 (a gather along one axis through
``samd_gather3``), and ``RandomInterleaver`` (:198-497) with one permutation for the whole batch.
``Turbo3GPPInterleaver`` is outside the hot path."""
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
        x = _ffi.to_device(x, self.rdtype)
        if self._perm_seq is None or x.shape[self._axis] != self._perm_seq.shape[0]:
            self.build(tuple(x.shape))
        inverse = self._inverse if inverse is None else bool(inverse)
        perm, zero = self._device_perm(inverse)
        xm = torch.movedim(x, self._axis, -1).contiguous()          # layout plumbing only
        n = xm.shape[-1]
        rows = xm.numel() // n if n else 0
        out = torch.empty_like(xm)
        if rows:
            _ffi.check(_ffi.lib().samd_gather3(_ffi.ptr(xm), _ffi.ptr(zero), _ffi.ptr(perm), rows, 1, n, 1, n, xm.element_size() // 4,
                                               _ffi.ptr(out), _ffi.stream()), "RowColumnInterleaver")
        return wrap(torch.movedim(out, -1, self._axis).contiguous())


class RandomInterleaver(Block):
    """``RandomInterleaver(seed=None, keep_batch_constant=True, inverse=False, keep_state=True, axis=-1)(x, seed=None,
    inverse=None)`` (interleaving.py:198-497): a pseudo-random permutation along ``axis`` = the argsort of i.i.d.
    draws (:330-366).  ``keep_state=True`` (default): one permutation for the block's lifetime, drawn from the build's
    Philox stream with the block's seed; an explicit ``seed`` in the call draws the permutation of that seed (the way the
    reference pairs an interleaver with its deinterleaver); ``keep_state=False`` without a seed: a fresh one per call.
    One permutation per batch EXAMPLE (``keep_batch_constant=False``) has no HIP path."""

    def __init__(self, seed=None, keep_batch_constant=True, inverse=False, keep_state=True, axis=-1, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(keep_batch_constant, bool):
            raise TypeError("keep_batch_constant must be bool.")
        if not keep_batch_constant:
            raise NotImplementedError("RandomInterleaver: keep_batch_constant=False has no HIP path")
        if not isinstance(axis, int):
            raise TypeError("axis must be int.")
        if not isinstance(inverse, bool):
            raise TypeError("inverse must be bool.")
        if not isinstance(keep_state, bool):
            raise TypeError("keep_state must be bool.")
        if seed is not None and not isinstance(seed, int):
            raise TypeError("seed must be int.")
        if axis == 0:
            raise ValueError("Cannot permute batch_dim.")
        from ..config import config
        self._seed = int(seed) if seed is not None else int(config.rng.seed) * 7919 + int(config.rng.next_call()) + 1
        self._keep_batch_constant, self._inverse, self._keep_state, self._axis = True, inverse, keep_state, axis
        self._fresh = 0
        self._cache = {}

    seed = property(lambda self: self._seed)
    axis = property(lambda self: self._axis)
    keep_state = property(lambda self: self._keep_state)

    def _perm(self, seed, n):
        """(perm, inverse perm) int32 device tables of the permutation of ``seed`` for length n"""
        key = (int(seed), int(n))
        if key not in self._cache:
            if len(self._cache) > 64:
                self._cache.clear()
            r = torch.zeros(max(n, 1), dtype=torch.complex64, device=_ffi.device())
            one = torch.ones(1, dtype=torch.float32, device=r.device)
            _ffi.check(_ffi.lib().samd_awgn_c64(_ffi.ptr(r), _ffi.ptr(one), 1, key[0] & 0x7FFFFFFFFFFFFFFF, 1, r.numel(), _ffi.ptr(r),
                                                _ffi.stream()), "RandomInterleaver draws")      # CN(0, 1) on the stream (seed, call 1)
            perm = torch.argsort(r.real[:n], stable=True).to(torch.int32)               # (index plumbing)
            inv = torch.argsort(perm, stable=True).to(torch.int32)
            self._cache[key] = (perm.contiguous(), inv.contiguous(), _ffi.to_device(np.zeros(1, np.int32), torch.int32))
        return self._cache[key]

    def find_s_min(self, seed, seq_length, s_min_stop=0):
        """S-parameter of the permutation of ``seed`` (interleaving.py:285-328): the smallest distance between the images
        of any two positions less than S apart (host; analysis helper)."""
        perm = self._perm(seed, seq_length)[0].cpu().numpy().astype(np.int64)
        s_min = seq_length
        for i in range(len(perm)):
            for j in range(-s_min, s_min):
                if j == 0 or not 0 <= i + j < seq_length:
                    continue
                d = abs(perm[i] - perm[i + j])
                if d <= abs(j):
                    s_min = min(s_min, abs(j))
                if d < s_min and abs(j) < s_min:
                    s_min = min(s_min, int(d))
            if s_min <= s_min_stop:
                break
        return int(s_min)

    def call(self, x, /, *, seed=None, inverse=None, **kwargs):
        x = _ffi.to_device(x, self.rdtype)
        if self._axis >= x.dim() or self._axis < -x.dim():
            raise ValueError("Axis does not match input shape")
        if seed is None:
            if self._keep_state:
                seed = self._seed
            else:
                self._fresh += 1
                seed = self._seed + 104729 * self._fresh
        inverse = self._inverse if inverse is None else bool(inverse)
        xm = torch.movedim(x, self._axis, -1).contiguous()
        n = xm.shape[-1]
        perm, inv, zero = self._perm(int(seed), n)
        rows = xm.numel() // n if n else 0
        out = torch.empty_like(xm)
        if rows:
            _ffi.check(_ffi.lib().samd_gather3(_ffi.ptr(xm), _ffi.ptr(zero), _ffi.ptr(inv if inverse else perm), rows, 1, n, 1, n, xm.element_size() // 4,
                                               _ffi.ptr(out), _ffi.stream()), "RandomInterleaver")
        return wrap(torch.movedim(out, -1, self._axis).contiguous())


class Deinterleaver(Block):
    """``Deinterleaver(interleaver)(x, seed=None)`` = ``interleaver(x, inverse=True)``."""

    def __init__(self, interleaver, precision=None, **kwargs):
        if not isinstance(interleaver, (RowColumnInterleaver, RandomInterleaver)):
            raise ValueError("interleaver is not a valid interleaver instance.")
        self._interleaver = interleaver
        super().__init__(precision=interleaver.precision if precision is None else precision, **kwargs)

    interleaver = property(lambda self: self._interleaver)

    def call(self, x, seed=None):
        return self._interleaver(x, seed=seed, inverse=True)
