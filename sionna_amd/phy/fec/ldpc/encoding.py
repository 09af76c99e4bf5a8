"""5G-NR LDPC encoder with rate matching - host-side mirror of
``sionna.phy.fec.ldpc.LDPC5GEncoder`` (reference src/sionna/phy/fec/ldpc/encoding.py).

The reference lifts the base graph into flat gather lists and encodes with four
gather+reduce_sum passes (encoding.py:524-591).  Here the code stays in its
quasi-cyclic form: the handle passed to the C-ABI is just the list of base-graph entries
(row, column, shift) for the selected lifting set, and ``csrc/ldpc5g.hip`` encodes with
rotated XORs in LDS (RU method in closed form).  Everything at init time is NumPy, like
the reference's SciPy construction.
"""
import ctypes as C
import numbers
import os

import numpy as np
import scipy.sparse as sp
import torch

from .... import _ffi
from ...block import Block

_TABLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "codes", "bg_tables.npz")

# 38.212 Table 5.3.2-1: lifting sizes Z = a * 2^j per set index i_LS
_LIFT_A = (2, 3, 5, 7, 9, 11, 13, 15)
_BG_SHAPE = {"bg1": (46, 68), "bg2": (42, 52)}


def _select_base_graph(k, r, bg):
    """38.212 Sec. 7.2.2 selection as implemented by the reference (encoding.py:248-282)."""
    if bg is None:
        bg = "bg2" if (k <= 292 or (k <= 3824 and r <= 0.67) or r <= 0.25) else "bg1"
    elif bg not in ("bg1", "bg2"):
        raise ValueError("Basegraph must be bg1, bg2 or None.")
    if bg == "bg1" and k > 8448:
        raise ValueError("K is not supported by BG1 (too large).")
    if bg == "bg2" and k > 3840:
        raise ValueError(f"K is not supported by BG2 (too large) k ={k}.")
    if bg == "bg1" and r < 1 / 3:
        raise ValueError("Only coderate>1/3 supported for BG1. Remark: Repetition coding is currently not supported.")
    if bg == "bg2" and r < 1 / 5:
        raise ValueError("Only coderate>1/5 supported for BG2. Remark: Repetition coding is currently not supported.")
    return bg


def _select_lifting(k, bg):
    """Smallest k_b*Z >= k over all lifting sizes (encoding.py:354-409) -> (Z, i_LS, k_b)."""
    if bg == "bg1":
        kb_sel = 22
    else:
        kb_sel = 10 if k > 640 else 9 if k > 560 else 8 if k > 192 else 6
    best = None
    for i_ls, a in enumerate(_LIFT_A):
        zz = a
        while zz <= 384:
            if kb_sel * zz >= k and (best is None or kb_sel * zz < best[0]):
                best = (kb_sel * zz, zz, i_ls)     # strict '<' keeps the first set on ties
            zz *= 2
    return best[1], best[2], (22 if bg == "bg1" else 10)


class LDPC5GEncoder(Block):
    """``LDPC5GEncoder(k, n, num_bits_per_symbol=None, bg=None, precision=None)``
    (reference encoding.py:14-137); ``call(bits[..., k]) -> [..., n]``."""

    def __init__(self, k, n, num_bits_per_symbol=None, bg=None, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(k, numbers.Number):
            raise TypeError("k must be a number.")
        if not isinstance(n, numbers.Number):
            raise TypeError("n must be a number.")
        k, n = int(k), int(n)
        if k > 8448:
            raise ValueError("Unsupported code length (k too large).")
        if k < 12:
            raise ValueError("Unsupported code length (k too small).")
        if n > 316 * 384:
            raise ValueError("Unsupported code length (n too large).")
        if n < 0:
            raise ValueError("Unsupported code length (n negative).")
        self._k, self._n = k, n
        self._coderate = k / n
        if self._coderate > 948 / 1024:
            print(f"Warning: effective coderate r>948/1024 for n={n}, k={k}.")
        if self._coderate > 0.95:
            raise ValueError(f"Unsupported coderate (r>0.95) for n={n}, k={k}.")
        if self._coderate < 1 / 5:
            raise ValueError("Unsupported coderate (r<1/5).")
        self._bg = _select_base_graph(k, self._coderate, bg)
        self._z, self._i_ls, self._k_b = _select_lifting(k, self._bg)
        t = np.load(_TABLES)
        self._bg_rows = np.ascontiguousarray(t[f"{self._bg}_row"], dtype=np.int16)
        self._bg_cols = np.ascontiguousarray(t[f"{self._bg}_col"], dtype=np.int16)
        self._bg_shifts = np.ascontiguousarray(t[f"{self._bg}_shift"][:, self._i_ls], dtype=np.int16)
        mb, nb = _BG_SHAPE[self._bg]
        self._n_ldpc = nb * self._z
        self._k_ldpc = self._k_b * self._z
        self._num_bits_per_symbol = num_bits_per_symbol
        if num_bits_per_symbol is not None:
            self._out_int, self._out_int_inv = self.generate_out_int(self._n, num_bits_per_symbol)
        self._pcm = None
        self._handles = {}

    # ------------------------------------------------------------ properties (encoding.py:143-190)
    k = property(lambda self: self._k)
    n = property(lambda self: self._n)
    coderate = property(lambda self: self._coderate)
    k_ldpc = property(lambda self: self._k_ldpc)
    n_ldpc = property(lambda self: self._n_ldpc)
    z = property(lambda self: self._z)
    num_bits_per_symbol = property(lambda self: self._num_bits_per_symbol)
    out_int = property(lambda self: self._out_int)
    out_int_inv = property(lambda self: self._out_int_inv)

    @property
    def pcm(self):
        """Lifted parity-check matrix (scipy csr), row r*Z+i has a one in column
        c*Z + (i+shift) mod Z for every base entry (r,c) (encoding.py:322-352)."""
        if self._pcm is None:
            z = self._z
            i = np.arange(z)
            r = (self._bg_rows.astype(np.int64)[:, None] * z + i[None, :]).reshape(-1)
            c = (self._bg_cols.astype(np.int64)[:, None] * z
                 + (i[None, :] + self._bg_shifts.astype(np.int64)[:, None]) % z).reshape(-1)
            mb, nb = _BG_SHAPE[self._bg]
            self._pcm = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(mb * z, nb * z))
        return self._pcm

    def generate_out_int(self, n, num_bits_per_symbol):
        """38.212 Sec. 5.4.2.2 bit interleaver and its inverse (encoding.py:196-246)."""
        if n % 1 != 0:
            raise ValueError("n must be int.")
        if num_bits_per_symbol % 1 != 0:
            raise ValueError("num_bits_per_symbol must be int.")
        n, m = int(n), int(num_bits_per_symbol)
        if n <= 0:
            raise ValueError("n must be a positive integer.")
        if m <= 0:
            raise ValueError("num_bits_per_symbol must be a positive integer.")
        if n % m != 0:
            raise ValueError("n must be a multiple of num_bits_per_symbol.")
        o = np.arange(n)
        perm = (o % m) * (n // m) + o // m
        return perm, np.argsort(perm)

    # ------------------------------------------------------------ C-ABI handle
    def _handle(self, nb_pruned=0):
        """samd_ldpc5g_t for this code (one per pruning variant used by a decoder)."""
        gen = _ffi.options_generation()
        if getattr(self, "_handles_gen", gen) != gen:             # a development switch changed: handles capture them at creation
            self._retired = getattr(self, "_retired", []) + list(self._handles.values())   # may still be in flight: freed in __del__
            self._handles = {}
        self._handles_gen = gen
        if nb_pruned not in self._handles:
            h = C.c_void_p()
            m = 0 if self._num_bits_per_symbol is None else int(self._num_bits_per_symbol)
            _ffi.device()
            _ffi.check(_ffi.lib().samd_ldpc5g_create(
                1 if self._bg == "bg1" else 2, self._z,
                self._bg_rows.ctypes.data_as(C.c_void_p), self._bg_cols.ctypes.data_as(C.c_void_p),
                self._bg_shifts.ctypes.data_as(C.c_void_p), len(self._bg_rows), self._k, self._n, m,
                int(nb_pruned), C.byref(h)), "samd_ldpc5g_create")
            self._handles[nb_pruned] = h
        return self._handles[nb_pruned]

    def __del__(self):
        try:
            for h in list(self._handles.values()) + getattr(self, "_retired", []):
                _ffi.lib().samd_ldpc5g_destroy(h)
        except Exception:  # pylint: disable=broad-except
            pass

    # ------------------------------------------------------------ Block interface
    def build(self, input_shape):
        if input_shape[-1] != self._k:
            raise ValueError("Last dimension must be of length k.")

    def call(self, bits):
        """[..., k] float 0/1 -> [..., n] (encode, drop filler and first 2Z, keep n,
        interleave; encoding.py:599-668)."""
        bits = _ffi.to_device(bits, torch.float32)          # bits are exact in either precision
        if bits.shape[-1] != self._k:
            raise ValueError("Last dimension must be of length k.")
        lead = tuple(bits.shape[:-1])
        u = bits.reshape(-1, self._k)
        out = torch.empty((u.shape[0], self._n), dtype=torch.float32, device=u.device)
        if u.shape[0] > 0:
            _ffi.check(_ffi.lib().samd_ldpc5g_encode_f32(self._handle(0), _ffi.ptr(u), _ffi.ptr(out), u.shape[0],
                                                         _ffi.stream()), "LDPC5GEncoder")
        return out.reshape(lead + (self._n,)).to(self.rdtype)
