"""Polar encoders - mirror of ``sionna.phy.fec.polar.PolarEncoder`` / ``Polar5GEncoder``
(reference src/sionna/phy/fec/polar/encoding.py:14-740).  Code construction and the 38.212
rate-matching index tables are host-side NumPy (init time); encoding runs in the HIP kernel
``samd_polar_encode_f32`` (bits in LDS, log2(n) XOR stages, rate-matching gather on the way out)."""
import numbers

import numpy as np
import torch

from .... import _ffi
from ...block import Block
from ..crc import CRCEncoder
from .utils import generate_5g_ranking

# 38.212 Tab. 5.4.1.1-1 sub-block interleaver pattern and Tab. 5.3.1.1-1 input interleaver pattern
_P_SUB = np.array([0, 1, 2, 4, 3, 5, 6, 7, 8, 16, 9, 17, 10, 18, 11, 19, 12, 20, 13, 21, 14, 22, 15, 23, 24, 25, 26,
                   28, 27, 29, 30, 31])
_P_IL_MAX = np.array([
    0, 2, 4, 7, 9, 14, 19, 20, 24, 25, 26, 28, 31, 34, 42, 45, 49, 50, 51, 53, 54, 56, 58, 59, 61, 62, 65, 66, 67, 69,
    70, 71, 72, 76, 77, 81, 82, 83, 87, 88, 89, 91, 93, 95, 98, 101, 104, 106, 108, 110, 111, 113, 115, 118, 119, 120,
    122, 123, 126, 127, 129, 132, 134, 138, 139, 140, 1, 3, 5, 8, 10, 15, 21, 27, 29, 32, 35, 43, 46, 52, 55, 57, 60,
    63, 68, 73, 78, 84, 90, 92, 94, 96, 99, 102, 105, 107, 109, 112, 114, 116, 121, 124, 128, 130, 133, 135, 141, 6,
    11, 16, 22, 30, 33, 36, 44, 47, 64, 74, 79, 85, 97, 100, 103, 117, 125, 131, 136, 142, 12, 17, 23, 37, 48, 75, 80,
    86, 137, 143, 13, 18, 38, 144, 39, 145, 40, 146, 41, 147, 148, 149, 150, 151, 152, 153, 154, 155, 156, 157, 158,
    159, 160, 161, 162, 163])


class PolarEncoder(Block):
    """``PolarEncoder(frozen_pos, n)(bits[..., k]) -> [..., n]``."""

    def __init__(self, frozen_pos, n, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(n, numbers.Number):
            raise TypeError("n must be a number.")
        n = int(n)
        frozen_pos = np.asarray(frozen_pos)
        if not np.issubdtype(frozen_pos.dtype, np.integer):
            raise TypeError("frozen_pos must consist of ints.")
        if len(frozen_pos) > n:
            raise ValueError("Number of elements in frozen_pos cannot be greater than n.")
        if np.log2(n) != int(np.log2(n)):
            raise ValueError("n must be a power of 2.")
        self._k = n - len(frozen_pos)
        self._n = n
        self._frozen_pos = frozen_pos
        self._info_pos = np.setdiff1d(np.arange(n), frozen_pos)
        if self._k != len(self._info_pos):
            raise ValueError("Internal error: invalid info_pos generated.")
        self._out_idx = np.arange(n)
        self._dev = None

    k = property(lambda self: self._k)
    n = property(lambda self: self._n)
    frozen_pos = property(lambda self: self._frozen_pos)
    info_pos = property(lambda self: self._info_pos)

    def _encode_2d(self, u):
        """u [B, k] device float32 -> [B, len(out_idx)]"""
        if self._dev is None:
            i32 = lambda a: _ffi.to_device(np.ascontiguousarray(a, np.int32), torch.int32)
            self._dev = (i32(self._info_pos), i32(self._out_idx))
        info, oidx = self._dev
        out = torch.empty((u.shape[0], oidx.numel()), dtype=torch.float32, device=u.device)
        if u.shape[0] > 0:
            _ffi.check(_ffi.lib().samd_polar_encode_f32(_ffi.ptr(u), _ffi.ptr(info), _ffi.ptr(oidx), u.shape[0],
                                                        self._k, self._n, oidx.numel(), _ffi.ptr(out), _ffi.stream()),
                       "PolarEncoder")
        return out

    def build(self, input_shape):
        if input_shape[-1] != self._k:
            raise ValueError("Last input dimension must be of length k.")

    def call(self, bits):
        bits = _ffi.to_device(bits, torch.float32)          # bits are exact in either precision (block.py::_bits)
        if bits.shape[-1] != self._k:
            raise ValueError("Last input dimension must be of length k.")
        out = self._encode_2d(bits.reshape(-1, self._k))
        return self._bits(out.reshape(tuple(bits.shape[:-1]) + (out.shape[-1],)))


def _subblock_pattern(k):
    """y[n] = u[P(floor(32 n / k)) * k/32 + n mod k/32]   (38.212 Sec. 5.4.1.1)"""
    k = int(k)
    if k % 32 != 0:
        raise ValueError("length for sub-block interleaving must be a multiple of 32.")
    n = np.arange(k)
    return _P_SUB[(32 * n) // k] * (k // 32) + n % (k // 32)


def _channel_pattern(e):
    """Triangular channel interleaver of 38.212 Sec. 5.4.1.3: write row-wise into the upper-left
    triangle of a T x T matrix, read column-wise."""
    t = 0
    while t * (t + 1) // 2 < e:
        t += 1
    rows, cols = [], []
    for i in range(t):
        rows += [i] * (t - i)
        cols += list(range(t - i))
    rows, cols = np.array(rows)[:e], np.array(cols)[:e]          # the first e cells carry data
    return np.lexsort((rows, cols))                               # read order: column, then row


def _input_pattern(k):
    """Input bit interleaver of 38.212 Sec. 5.3.1.1 (downlink), K_IL_max = 164."""
    if k > 164:
        raise ValueError("Input interleaver only defined for length of 164.")
    return _P_IL_MAX[_P_IL_MAX >= 164 - k] - (164 - k)


class Polar5GEncoder(PolarEncoder):
    """``Polar5GEncoder(k, n, channel_type="uplink")``: CRC attachment, (downlink) input
    interleaving, Polar encoding, sub-block interleaving, rate matching and (uplink) channel
    interleaving of 38.212 (reference encoding.py:211-740)."""

    def __init__(self, k, n, channel_type="uplink", verbose=False, precision=None, **kwargs):
        if not isinstance(k, numbers.Number):
            raise TypeError("k must be a number.")
        if not isinstance(n, numbers.Number):
            raise TypeError("n must be a number.")
        k, n = int(k), int(n)
        if n < k:
            raise ValueError("Invalid coderate (>1).")
        if not isinstance(verbose, bool):
            raise TypeError("verbose must be bool.")
        if channel_type not in ("uplink", "downlink"):
            raise ValueError("Unsupported channel_type.")
        self._channel_type, self._k_target, self._n_target, self._verbose = channel_type, k, n, verbose
        crc_degree, n_polar, frozen_pos, idx_rm, idx_input = self._init_rate_match(k, n)
        self._ind_rate_matching, self._ind_input_int = idx_rm, idx_input
        self._enc_crc = CRCEncoder(crc_degree, precision="single")       # inner stage: bits as float32 (exact)
        super().__init__(frozen_pos, n_polar, precision=precision, **kwargs)
        self._out_idx = np.asarray(idx_rm, np.int32)
        self._dev_iil = None

    enc_crc = property(lambda self: self._enc_crc)
    k_target = property(lambda self: self._k_target)
    n_target = property(lambda self: self._n_target)
    k_polar = property(lambda self: self._k)
    n_polar = property(lambda self: self._n)
    k = property(lambda self: self._k_target)
    n = property(lambda self: self._n_target)

    # the reference exposes the patterns as methods acting on index vectors
    def subblock_interleaving(self, u):
        u = np.asarray(u)
        return u[_subblock_pattern(u.shape[-1])]

    def channel_interleaver(self, c):
        c = np.asarray(c)
        return c[_channel_pattern(c.shape[-1])]

    def input_interleaver(self, c):
        c = np.asarray(c)
        return c[_input_pattern(len(c))]

    def _init_rate_match(self, k_target, n_target):
        if n_target < k_target:
            raise ValueError("n must be larger or equal k.")
        if n_target < 18:
            raise ValueError("n<18 is not supported by the 5G Polar coding scheme.")
        if k_target > 1013:
            raise ValueError("k too large - currently, no codeword segmentation supported.")
        if n_target > 1088:
            raise ValueError("n too large - currently, no codeword segmentation supported.")
        if self._channel_type == "uplink":
            if 12 <= k_target <= 19:
                crc_pol, k_crc = "CRC6", 6
                print("Warning: For 12<=k<=19 additional 3 parity-check bits are defined in 38.212. "
                      "They are currently not implemented by this encoder and, thus, ignored.")
            elif k_target >= 20:
                crc_pol, k_crc = "CRC11", 11
            else:
                raise ValueError("k_target<12 is not supported in 5G NR for the uplink; please use 'channel coding "
                                 "of small block lengths' scheme from Sec. 5.3.3 in 3GPP 38.212 instead.")
        else:
            if k_target > 140:
                raise ValueError("k too large for downlink configuration.")
            if n_target < 25:
                raise ValueError("n too small for downlink configuration with 24 bit CRC.")
            if n_target > 576:
                raise ValueError("n too large for downlink configuration.")
            crc_pol, k_crc = "CRC24C", 24
        k_polar = k_target + k_crc
        if k_polar > n_target:
            raise ValueError("Device is not expected to be configured with k_polar + k_crc + n_pc > n_target.")
        # mother code length (38.212 Sec. 5.3.1)
        lg = int(np.ceil(np.log2(n_target)))
        n1 = lg - 1 if (n_target <= 9 / 8 * 2 ** (lg - 1) and k_polar / n_target < 9 / 16) else lg
        n2 = int(np.ceil(np.log2(8 * k_polar)))
        n_polar = 2 ** max(min(n1, n2, 10), 5)
        # bit channels frozen by the rate matching (38.212 Sec. 5.4.1.1)
        sub = _subblock_pattern(n_polar)
        if n_target >= n_polar:
            pre, mode = np.zeros(0, int), "repetition"
        elif k_polar / n_target <= 7 / 16:
            mode = "puncturing"
            pre = _subblock_pattern(32 * int(np.ceil((n_polar - n_target) / 32)))[:n_polar - n_target]
            if n_target >= 3 * n_polar / 4:
                t = int(np.ceil(3 / 4 * n_polar - n_target / 2) - 1)
            else:
                t = int(np.ceil(9 / 16 * n_polar - n_target / 4) - 1)
            pre = np.concatenate([pre, np.arange(max(t, 0))])
        else:
            mode = "shortening"
            pre = sub[n_target:]
        pre = np.unique(pre).astype(int)
        ranking, _ = generate_5g_ranking(0, n_polar, sort=False)
        cand = ranking[~np.isin(ranking, pre)]                     # still in ascending reliability
        info_pos = np.sort(cand[len(cand) - k_polar:]).astype(int)
        frozen_pos = np.setdiff1d(np.arange(n_polar), info_pos, assume_unique=True)
        # bit selection (38.212 Sec. 5.4.1.2) on the sub-block interleaved codeword
        e = np.arange(n_target)
        sel = e % n_polar if mode == "repetition" else (e + n_polar - n_target if mode == "puncturing" else e)
        if self._channel_type == "uplink":
            sel = sel[_channel_pattern(n_target)]
        idx_rm = sub[sel]
        idx_input = _input_pattern(k_polar) if self._channel_type == "downlink" else None
        if self._verbose:
            print(f"Using {mode} for rate-matching.")
            print(f"Code parameters after rate-matching: k = {k_target}, n = {n_target}")
            print(f"Polar mother code: k_polar = {k_polar}, n_polar = {n_polar}")
            print("Using", crc_pol)
            print("Frozen positions: ", frozen_pos)
            print("Channel type: " + self._channel_type)
        return crc_pol, n_polar, frozen_pos, idx_rm, idx_input

    def build(self, input_shape):
        if input_shape[-1] != self._k_target:
            raise ValueError("Invalid input shape.")

    def call(self, bits):
        bits = _ffi.to_device(bits, torch.float32)
        if bits.shape[-1] != self._k_target:
            raise ValueError("Invalid input shape.")
        u_crc = self._enc_crc(bits.reshape(-1, self._k_target)).as_subclass(torch.Tensor)
        if self._channel_type == "downlink":
            if self._dev_iil is None:
                self._dev_iil = torch.from_numpy(np.asarray(self._ind_input_int, np.int64)).to(u_crc.device)
            u_crc = u_crc.index_select(1, self._dev_iil).contiguous()
        c = self._encode_2d(u_crc)
        return self._bits(c.reshape(tuple(bits.shape[:-1]) + (self._n_target,)))
