"""``LinearEncoder`` and ``AllZeroEncoder`` - mirrors of reference src/sionna/phy/fec/linear/
encoding.py:15-282 on ``samd_gf2_encode_f32`` (packed GF(2) parity instead of a float matmul)."""
import numpy as np
import torch

from .... import _ffi
from ...block import Block, wrap
from ..utils import pcm2gm


class LinearEncoder(Block):
    """``LinearEncoder(enc_mat, is_pcm=False)(info_bits [..., k]) -> [..., n]``: c = u G mod 2 for a
    binary generator matrix [k, n]; with ``is_pcm`` the matrix is a full-rank parity-check matrix
    [n-k, n] that is converted with ``pcm2gm`` (same pivoting as the reference, hence the same
    codewords)."""

    def __init__(self, enc_mat, is_pcm=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        enc_mat = np.asarray(enc_mat.todense() if hasattr(enc_mat, "todense") else enc_mat)
        if not ((enc_mat == 0) | (enc_mat == 1)).all():
            raise ValueError("enc_mat is not binary.")
        if enc_mat.ndim != 2:
            raise ValueError("enc_mat must be 2-D array.")
        self._gm = pcm2gm(enc_mat) if is_pcm else enc_mat
        self._k, self._n = int(self._gm.shape[0]), int(self._gm.shape[1])
        self._coderate = self._k / self._n
        if self._k > self._n:
            raise ValueError("Invalid matrix dimensions.")
        # columns of G packed along k, LSB = lowest row index
        words = (self._k + 31) // 32
        g = np.zeros((self._n, words * 32), np.uint8)
        g[:, :self._k] = np.asarray(self._gm, np.uint8).T
        self._cols = np.packbits(g.reshape(self._n, words, 32), axis=-1, bitorder="little").view(np.uint32).reshape(self._n, words)
        self._dev = None

    k = property(lambda self: self._k)
    n = property(lambda self: self._n)
    gm = property(lambda self: self._gm)
    coderate = property(lambda self: self._coderate)

    def build(self, input_shapes):
        if input_shapes[-1] != self._k:
            raise ValueError(f"Last dimension must be of size k={self._k}.")

    def call(self, bits, /):
        u = _ffi.to_device(bits, torch.float32)              # bits are exact in either precision (block.py::_bits)
        if u.shape[-1] != self._k:
            raise ValueError(f"Last dimension must be of size k={self._k}.")
        if self._dev is None:
            self._dev = _ffi.to_device(self._cols.view(np.int32), torch.int32)
        lead = tuple(u.shape[:-1])
        u2 = u.reshape(-1, self._k).contiguous()
        out = torch.empty((u2.shape[0], self._n), dtype=torch.float32, device=u.device)
        _ffi.check(_ffi.lib().samd_gf2_encode_f32(_ffi.ptr(u2), _ffi.ptr(self._dev), u2.shape[0], self._k, self._n,
                                                  _ffi.ptr(out), _ffi.stream()), "LinearEncoder")
        return wrap(self._bits(out.reshape(lead + (self._n,))))


class AllZeroEncoder(Block):
    """``AllZeroEncoder(k, n)(bits [..., k]) -> zeros [..., n]`` (all-zero codeword simulations)."""

    def __init__(self, k, n, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(k, int) or not isinstance(n, int):
            raise TypeError("k and n must be int.")
        if k < 0 or n < 0:
            raise ValueError("k and n cannot be negative.")
        if k > n:
            raise ValueError("Invalid coderate (>1).")
        self._k, self._n, self._coderate = k, n, k / n

    k = property(lambda self: self._k)
    n = property(lambda self: self._n)
    coderate = property(lambda self: self._coderate)

    def call(self, inputs, /):
        u = _ffi.to_device(inputs, torch.float32)
        return wrap(torch.zeros(tuple(u.shape[:-1]) + (self._n,), dtype=torch.float32, device=u.device))
