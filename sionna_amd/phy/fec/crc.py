"""Cyclic redundancy checks of 38.212 Sec. 5.1 - mirror of ``sionna.phy.fec.crc`` (reference
src/sionna/phy/fec/crc.py:11-321).  The reference multiplies by a dense GF(2) generator matrix;
the HIP kernel ``samd_crc_f32`` runs the equivalent shift register (one lane per word)."""
import torch

from ... import _ffi
from ..block import Block

# generator polynomials: exponents with non-zero coefficient (crc.py:100-122)
_POLYS = {"CRC24A": (24, [24, 23, 18, 17, 14, 11, 10, 7, 6, 5, 4, 3, 1, 0]),
          "CRC24B": (24, [24, 23, 6, 5, 1, 0]),
          "CRC24C": (24, [24, 23, 21, 20, 17, 15, 13, 12, 8, 4, 2, 1, 0]),
          "CRC16": (16, [16, 12, 5, 0]), "CRC11": (11, [11, 10, 9, 5, 0]), "CRC6": (6, [6, 5, 0])}


def crc_poly_mask(crc_degree):
    """(crc_length, bit mask of the generator without its leading term)."""
    if crc_degree not in _POLYS:
        raise ValueError("Invalid CRC Polynomial")
    length, exps = _POLYS[crc_degree]
    return length, sum(1 << e for e in exps if e < length)


class CRCEncoder(Block):
    """``CRCEncoder(crc_degree)(bits[..., k]) -> [..., k + crc_length]`` (parity appended)."""

    def __init__(self, crc_degree, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(crc_degree, str):
            raise TypeError("crc_degree must be str")
        self._crc_degree = crc_degree
        self._crc_length, self._mask = crc_poly_mask(crc_degree)
        self._k = self._n = None

    crc_degree = property(lambda self: self._crc_degree)
    crc_length = property(lambda self: self._crc_length)
    k = property(lambda self: self._k)
    n = property(lambda self: self._n)

    @property
    def crc_pol(self):
        """Polynomial coefficients, MSB first (crc.py:124-128)."""
        import numpy as np
        length, exps = _POLYS[self._crc_degree]
        p = np.zeros(length + 1, int)
        p[[length - e for e in exps]] = 1
        return p

    def build(self, input_shape):
        self._k = input_shape[-1]
        self._n = self._k + self._crc_length

    def call(self, bits, /):
        bits = _ffi.to_device(bits, torch.float32)          # bits are exact in either precision (block.py::_bits)
        if bits.shape[-1] != self._k:
            self.build(bits.shape)
        k = bits.shape[-1]
        out = torch.empty(tuple(bits.shape[:-1]) + (k + self._crc_length,), dtype=torch.float32, device=bits.device)
        _ffi.check(_ffi.lib().samd_crc_f32(_ffi.ptr(bits), bits.numel() // k, k, self._mask, self._crc_length, 0,
                                           _ffi.ptr(out), _ffi.stream()), "CRCEncoder")
        return self._bits(out)


class CRCDecoder(Block):
    """``CRCDecoder(crc_encoder)(x_crc[..., k+crc]) -> (x_info[..., k], crc_valid[..., 1] bool)``."""

    def __init__(self, crc_encoder, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(crc_encoder, CRCEncoder):
            raise TypeError("crc_encoder must be an instance of CRCEncoder.")
        self._encoder = crc_encoder

    crc_degree = property(lambda self: self._encoder.crc_degree)
    encoder = property(lambda self: self._encoder)

    def call(self, x_crc, /):
        x = _ffi.to_device(x_crc, torch.float32)
        n = x.shape[-1]
        enc = self._encoder
        valid = torch.empty(tuple(x.shape[:-1]) + (1,), dtype=torch.float32, device=x.device)
        _ffi.check(_ffi.lib().samd_crc_f32(_ffi.ptr(x), x.numel() // n, n, enc._mask, enc.crc_length, 1,
                                           _ffi.ptr(valid), _ffi.stream()), "CRCDecoder")
        return self._bits(x[..., :n - enc.crc_length]), valid > 0.5
