"""AWGN channel - mirror of reference src/sionna/phy/channel/awgn.py:10-78."""
import torch

from ... import _ffi
from ..block import Block
from ..config import config


class AWGN(Block):
    """``AWGN()(x, no)``: y = x + sqrt(no) * CN(0,1); ``no`` scalar or broadcastable to x."""

    def __init__(self, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)

    def call(self, x, no):
        if self.precision == "double":
            # the unit-variance draws come from the same float32 Philox / Box-Muller stream (the build's own RNG
            # specification, oracle/utils.py); scaling and addition in float64
            x = _ffi.to_device(x, torch.complex128)
            no = _ffi.to_device(no, torch.float64)
            while 1 < no.numel() and no.dim() < x.dim():
                no = no.unsqueeze(-1)
            w = torch.empty(x.shape, dtype=torch.complex64, device=x.device)
            one = torch.ones(1, dtype=torch.float32, device=x.device)
            rng = config.rng
            zero = torch.zeros_like(w)                     # (a named tensor: its storage must outlive the launch call)
            _ffi.check(_ffi.lib().samd_awgn_c64(_ffi.ptr(zero), _ffi.ptr(one), 1, rng.seed, rng.next_call(),
                                                w.numel(), _ffi.ptr(w), _ffi.stream()), "AWGN")
            return x + w.to(torch.complex128) * torch.sqrt(no)
        x = _ffi.to_device(x, torch.complex64)
        no = _ffi.to_device(no, torch.float32)
        if no.numel() == 1:
            no = no.reshape(1)
        else:
            # awgn.py:70-76: no is expanded to the rank of x from the right
            while no.dim() < x.dim():
                no = no.unsqueeze(-1)
            no = torch.broadcast_to(no, x.shape).contiguous()
        y = torch.empty_like(x)
        rng = config.rng
        _ffi.check(_ffi.lib().samd_awgn_c64(_ffi.ptr(x), _ffi.ptr(no), no.numel(), rng.seed, rng.next_call(),
                                            x.numel(), _ffi.ptr(y), _ffi.stream()), "AWGN")
        return y
