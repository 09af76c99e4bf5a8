"""``Object`` / ``Block`` protocol - mirror of reference src/sionna/phy/block.py:13-155.

``Block.__call__`` converts array inputs to device tensors of the block's precision,
calls ``build(*shapes)`` exactly once, then ``call(*args)`` (block.py:144-155).  Unknown
constructor kwargs are accepted and ignored like in the reference (block.py:25).
"""
import weakref
from abc import ABC, abstractmethod

import numpy as np
import torch

from .config import config, dtypes


# ---------------------------------------------------------------------------------------------------------------------
# Deferred block outputs.  Some blocks of the reference exist only to hand a large tensor to the next block (the
# nearest-neighbour-interpolated channel estimate: 64 of config C4's 120 bytes per resource element, written by
# LSChannelEstimator and read back by LMMSEEqualizer).  Such a block returns its output tensor with allocated but UNFILLED
# storage and a ``Pending`` record saying how to fill it.  A consumer that can work from the recipe (the fused
# LS + LMMSE (+ demapper) kernel) does so and the tensor is never filled; ANY other use - a torch operation, ``.numpy()``,
# the C-ABI pointer of ``_ffi.ptr`` - fills it first (``__torch_function__`` below), so the tensor behaves like an
# ordinary one.  This is how the HIP path fuses across the reference's block boundaries without a tracing compiler.
# The recipe keeps references to its inputs; they must not be modified in place before the tensor is used.
# ---------------------------------------------------------------------------------------------------------------------
_PENDING = set()          # id() of every live deferred tensor (entries leave on fill or garbage collection)
# property getters / methods that only look at a tensor's metadata and therefore do not fill a deferred tensor
_META_GET = {"shape", "dtype", "device", "is_cuda", "ndim", "requires_grad", "layout", "names", "is_sparse", "is_quantized",
             "is_meta", "grad_fn", "is_leaf", "itemsize", "nbytes", "is_cpu", "is_nested", "grad", "output_nr", "_version"}
# ("type" only without arguments - t.type(dtype) converts the VALUES; not __repr__ / untyped_storage / is_set_to: they show or
# hand out the storage.  data_ptr stays: the address identifies the tensor, _ffi.ptr fills before it passes one on.)
_META_FN = {"dim", "size", "numel", "is_contiguous", "stride", "element_size", "ndimension", "is_floating_point", "is_complex",
            "nelement", "get_device", "storage_offset", "is_pinned", "__len__", "__hash__", "is_shared",
            "has_names", "is_same_size", "is_signed", "is_inference", "is_conj", "is_neg", "requires_grad_", "__class__",
            "_is_view", "data_ptr"}


class Pending:
    """How to fill a deferred tensor: ``fill(plain_tensor)`` launches the kernel(s) that write its storage; ``kind`` and
    the keyword attributes let a fusing consumer recognise the recipe."""

    def __init__(self, kind, fill, guard=(), **info):
        self.kind, self.fill = kind, fill
        # inputs the recipe will read LATER: (tensor, its version counter now).  An in-place modification between the
        # block's call and the first use of its deferred output would silently change that output - check_guard raises.
        self.guard = [(g, g._version) for g in guard if isinstance(g, torch.Tensor)]
        self.__dict__.update(info)

    def check_guard(self):
        for g, ver in self.guard:
            if g._version != ver:
                raise RuntimeError("an input of a deferred block output was modified in place before that output was used "
                                   "(e.g. a reused receive buffer): use the output first, or build the block with defer=False")


def pending_of(t):
    """The Pending record of a deferred tensor, else None (plain tensors, numpy arrays, scalars)."""
    return t.__dict__.get("_samd_pending") if isinstance(t, Tensor) else None


def materialize(t):
    """Fill a deferred tensor now (no-op for everything else); returns ``t``."""
    p = pending_of(t)
    if p is not None:
        p.check_guard()
        del t.__dict__["_samd_pending"]
        _PENDING.discard(id(t))
        p.fill(torch.Tensor._make_subclass(torch.Tensor, t) if type(t) is not torch.Tensor else t)
    return t


def defer(t, pending):
    """Mark the (unfilled) tensor ``t`` as deferred; returns it as a ``Tensor``."""
    t = t if type(t) is Tensor else t.as_subclass(Tensor)
    t.__dict__["_samd_pending"] = pending
    _PENDING.add(id(t))
    weakref.finalize(t, _PENDING.discard, id(t))
    return t


def _fill_all(objs):
    for o in objs:
        if isinstance(o, Tensor):
            if "_samd_pending" in o.__dict__:
                materialize(o)
        elif isinstance(o, (list, tuple)):
            _fill_all(o)


class Tensor(torch.Tensor):
    """torch.Tensor whose ``.numpy()`` also works for device tensors (notebooks call ``x.numpy()`` on block outputs) and
    whose storage may be filled on first use (see ``Pending``)."""

    def numpy(self, *args, **kwargs):  # pylint: disable=arguments-differ
        materialize(self)
        return self.as_subclass(torch.Tensor).detach().cpu().numpy(*args, **kwargs)

    def as_subclass(self, cls):  # pylint: disable=arguments-differ
        materialize(self)               # (as_subclass does not pass through __torch_function__: the alias would show unfilled storage)
        return super().as_subclass(cls)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if _PENDING:                                   # some tensor somewhere is deferred: is one of the operands?
            name = getattr(func, "__name__", "")
            meta = name in _META_FN or (name == "__get__" and getattr(getattr(func, "__self__", None), "__name__", "") in _META_GET) \
                or (name == "type" and len(args) <= 1 and not kwargs)
            if not meta:
                _fill_all(args)
                if kwargs:
                    _fill_all(kwargs.values())
        return super().__torch_function__(func, types, args, kwargs or {})


def wrap(t):
    if type(t) is Tensor:
        return t
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

    _np_cdtype = property(lambda self: dtypes[self.precision]["np"]["cdtype"])
    _np_rdtype = property(lambda self: dtypes[self.precision]["np"]["rdtype"])

    def _bits(self, t):
        """Output of a bit-domain block (bits are exact in either precision: the kernels carry them as float32)."""
        return t if self._precision == "single" else t.to(self.rdtype)

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
