"""``Object`` / ``Block`` protocol - mirror of reference src/sionna/phy/block.py:13-155.

``Block.__call__`` converts array inputs to device tensors of the block's precision,
calls ``build(*shapes)`` exactly once, then ``call(*args)`` (block.py:144-155).  Unknown
constructor kwargs are accepted and ignored like in the reference (block.py:25).
"""
from abc import ABC, abstractmethod

import numpy as np
import torch

from .config import config, dtypes


class Tensor(torch.Tensor):
    """torch.Tensor whose ``.numpy()`` also works for device tensors (notebooks call
    ``x.numpy()`` on block outputs)."""

    def numpy(self, *args, **kwargs):  # pylint: disable=arguments-differ
        return self.as_subclass(torch.Tensor).detach().cpu().numpy(*args, **kwargs)


def wrap(t):
    return t.as_subclass(Tensor) if isinstance(t, torch.Tensor) else t


class Object(ABC):
    # pylint: disable=unused-argument
    def __init__(self, *args, precision=None, **kwargs):
        if precision is None:
            self._precision = config.precision
        elif precision in ["single", "double"]:
            self._precision = precision
        else:
            raise ValueError("'precision' must be 'single' or 'double'")

    @property
    def precision(self):
        return self._precision

    @property
    def cdtype(self):
        return dtypes[self.precision]["torch"]["cdtype"]

    @property
    def rdtype(self):
        return dtypes[self.precision]["torch"]["rdtype"]

    def _require_single(self):
        """The HIP kernels compute in float32 (BASELINE north-star dtype)."""
        if self._precision != "single":
            raise NotImplementedError(
                f"{type(self).__name__}: the MI355X kernels implement precision='single' only")


class Block(Object):
    # pylint: disable=unused-argument
    def __init__(self, *args, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._built = False

    @property
    def built(self):
        return self._built

    def build(self, *arg_shapes, **kwarg_shapes):
        pass

    @abstractmethod
    def call(self, *args, **kwargs):
        raise NotImplementedError("Subclasses must implement this method.")

    def _convert_to_tensor(self, v):
        """block.py:122-131: arrays/tensors are cast to the block's real/complex dtype."""
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(np.ascontiguousarray(v))
        if isinstance(v, torch.Tensor):
            from .. import _ffi
            dt = v.dtype
            if dt.is_floating_point:
                dt = self.rdtype
            elif dt.is_complex:
                dt = self.cdtype
            if v.dtype != dt or v.device != _ffi.device():
                v = v.to(device=_ffi.device(), dtype=dt)
        return v

    @staticmethod
    def _get_shape(v):
        if hasattr(v, "shape"):
            return tuple(v.shape)
        try:
            return tuple(np.asarray(v).shape)
        except Exception:  # pylint: disable=broad-except
            return ()

    def __call__(self, *args, **kwargs):
        args = [self._convert_to_tensor(a) for a in args]
        kwargs = {k: self._convert_to_tensor(v) for k, v in kwargs.items()}
        if not self._built:
            self.build(*[self._get_shape(a) for a in args],
                       **{k: self._get_shape(v) for k, v in kwargs.items()})
            self._built = True
        out = self.call(*args, **kwargs)
        if isinstance(out, tuple):
            return tuple(wrap(o) for o in out)
        return wrap(out)
