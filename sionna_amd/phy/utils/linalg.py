"""``inv_cholesky`` / ``matrix_pinv`` - mirrors of reference src/sionna/phy/utils/linalg.py:8-32, 35-66 on the per-item
kernels of csrc/mimo_linalg.hip (1 <= M, K <= 16; the precision follows the input, real inputs give real outputs)."""
import torch

from ... import _ffi
from ..block import wrap


def _as_complex(t):
    """(device tensor in complex64 / complex128, was_real)"""
    dt = str(getattr(t, "dtype", ""))
    dbl = dt.endswith("float64") or dt.endswith("complex128")
    real = not ("complex" in dt)
    return _ffi.to_device(t, torch.complex128 if dbl else torch.complex64).contiguous(), real, dbl


def _back(t, real):
    return wrap(t.real.contiguous() if real else t)


def inv_cholesky(tensor):
    """[..., M, M] Hermitian positive definite A = L L^H -> L^-1 (lower triangular)."""
    a, real, dbl = _as_complex(tensor)
    assert a.dim() >= 2 and a.shape[-1] == a.shape[-2], "the last two dimensions must be square"
    m = int(a.shape[-1])
    out = torch.empty_like(a)
    fn = _ffi.lib().samd_inv_cholesky_c128 if dbl else _ffi.lib().samd_inv_cholesky_c64
    _ffi.check(fn(_ffi.ptr(a), a.numel() // (m * m), m, _ffi.ptr(out), _ffi.stream()), "inv_cholesky")
    return _back(out, real)


def matrix_pinv(tensor):
    """[..., M, K] of full column rank -> (A^H A)^-1 A^H, [..., K, M]."""
    a, real, dbl = _as_complex(tensor)
    assert a.dim() >= 2, "rank >= 2 required"
    m, k = int(a.shape[-2]), int(a.shape[-1])
    out = torch.empty(tuple(a.shape[:-2]) + (k, m), dtype=a.dtype, device=a.device)
    fn = _ffi.lib().samd_matrix_pinv_c128 if dbl else _ffi.lib().samd_matrix_pinv_c64
    _ffi.check(fn(_ffi.ptr(a), a.numel() // (m * k), m, k, _ffi.ptr(out), _ffi.stream()), "matrix_pinv")
    return _back(out, real)
