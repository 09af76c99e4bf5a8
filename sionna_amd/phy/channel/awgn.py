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
        dbl = self.precision == "double"          # float64: the float32 stream's uniforms, Box-Muller and scaling in double
        x = _ffi.to_device(x, self.cdtype)
        no = _ffi.to_device(no, self.rdtype)
        if no.numel() == 1:
            no = no.reshape(1)
        else:
            # awgn.py:70-76: no is expanded to the rank of x from the right
            while no.dim() < x.dim():
                no = no.unsqueeze(-1)
            no = torch.broadcast_to(no, x.shape).contiguous()
        return self._add(x, no, config.rng.next_call())

    def _add(self, x, no, call_id, out=None):
        """the launch with a given call id of the Philox stream (OFDMChannel hands in the one it drew); out = x: in place"""
        y = torch.empty_like(x) if out is None else out
        fn = _ffi.lib().samd_awgn_c128 if self.precision == "double" else _ffi.lib().samd_awgn_c64
        _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(no), no.numel(), config.rng.seed, call_id, x.numel(), _ffi.ptr(y), _ffi.stream()), "AWGN")
        return y
