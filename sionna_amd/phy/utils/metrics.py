"""Error metrics - mirror of reference src/sionna/phy/utils/metrics.py:9-144.

Device tensors are reduced by the HIP kernel ``samd_count_errors_f32`` (two int64
counters per call); host tensors (e.g. from a user-supplied ``mc_fun`` that runs on the
CPU) are counted with plain torch ops - that branch is host bookkeeping, not a compute
fallback of the hot path.
"""
import numpy as np
import torch

from ... import _ffi


def _as_tensor(x):
    if isinstance(x, torch.Tensor):
        return x
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x)))


def count_errors_into(b, b_hat, counters, soft=False):
    """counters[0] += #(b != b_hat); counters[1] += #rows (last dim = block) with an error.

    ``counters``: int64[>=2] tensor on the same device as ``b``.  Asynchronous on GPU.
    """
    b, b_hat = _as_tensor(b), _as_tensor(b_hat)
    if b.shape != b_hat.shape:
        raise ValueError("b and b_hat must have the same shape")
    if b.is_cuda:
        b = b.to(torch.float32).contiguous()
        b_hat = b_hat.to(device=b.device, dtype=torch.float32).contiguous()
        block_len = b.shape[-1] if b.dim() > 0 else 1
        num_blocks = b.numel() // max(block_len, 1)
        assert counters.is_cuda and counters.dtype == torch.int64 and counters.is_contiguous()
        _ffi.check(_ffi.lib().samd_count_errors_f32(_ffi.ptr(b), _ffi.ptr(b_hat), num_blocks, block_len,
                                                    int(bool(soft)), _ffi.ptr(counters), _ffi.stream()),
                   "count_errors")
    else:
        b_hat = b_hat.to(b.dtype)
        if soft:
            b_hat = (b_hat > 0).to(b.dtype)
        err = b != b_hat
        counters[0] += int(err.sum())
        counters[1] += int(err.reshape(-1, err.shape[-1]).any(dim=-1).sum()) if err.dim() > 0 else int(err)
    return counters


def _count(b, b_hat):
    b = _as_tensor(b)
    c = torch.zeros(2, dtype=torch.int64, device=b.device)
    count_errors_into(b, b_hat, c)
    return c


def count_errors(b, b_hat):
    """Number of bit errors (metrics.py:94-117)."""
    return _count(b, b_hat)[0]


def count_block_errors(b, b_hat):
    """Number of block errors; a block is the last dimension (metrics.py:119-144)."""
    return _count(b, b_hat)[1]


def compute_ber(b, b_hat, precision="double"):
    """Bit error rate (metrics.py:9-40)."""
    b = _as_tensor(b)
    dt = torch.float64 if precision == "double" else torch.float32
    return (count_errors(b, b_hat).to(torch.float64) / max(b.numel(), 1)).to(dt)


def compute_bler(b, b_hat, precision="double"):
    """Block error rate (metrics.py:66-92)."""
    b = _as_tensor(b)
    dt = torch.float64 if precision == "double" else torch.float32
    nblk = b.numel() // max(b.shape[-1], 1) if b.dim() > 0 else 1
    return (count_block_errors(b, b_hat).to(torch.float64) / max(nblk, 1)).to(dt)
