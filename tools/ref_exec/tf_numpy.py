"""A NumPy-backed stand-in for the handful of TensorFlow calls that the reference's link-level code makes, so that the
reference's OWN source files can be executed in this container (TensorFlow is not installed, there is no network).

Purpose: pin ``oracle/`` to reference-EXECUTED outputs.  ``tools/ref_exec/loader.py`` imports reference files unmodified
from /root/reference with this module registered as ``tensorflow``; ``tools/gen_*_golden.py`` run them and store
input/output fixtures under tests/golden/.  Nothing here is product code, nothing here is shipped or imported by
``sionna_amd``; it never runs on the GPU box.

Numerical contract of the stand-in (what makes the fixtures meaningful):
  * every op computes in the dtype TensorFlow would (float32 stays float32; Python scalars are weak);
  * ``RaggedTensor`` = flat values + value_rowids (first axis ragged, like the reference uses it);
  * ragged ``reduce_sum/prod/min/max`` over the ragged axis follow TF-CPU's ``unsorted_segment_*`` kernels: one
    accumulator per row initialised with the identity, elements folded in storage order (for float32 sums the order
    matters; min/max/sign products are exact in any order);
  * add/sub/mul/abs/sign/min/max/where/clip are IEEE-exact in NumPy as in Eigen, so code that uses only those
    (vn_update_sum, cn_update_minsum, cn_update_offset_minsum, the whole min-sum BP loop) reproduces TF-CPU float32
    results bit for bit;  exp/log/tanh/atanh/cholesky go to NumPy's float32 routines, which are NOT bit-identical to
    Eigen's - fixtures that involve them are compared at 1e-5, never bit-exactly.
"""
import types

import numpy as np


class _Shape(list):
    """TensorShape stand-in: list-like (``[1]*rank + x.shape`` works), with ``rank`` and ``as_list``."""
    rank = property(len)
    ndims = property(len)

    def as_list(self):
        return list(self)

    def __hash__(self):
        return hash(tuple(self))


class DType:
    """tf.DType stand-in; NumPy accepts it wherever a dtype is expected (through the ``dtype`` attribute)."""

    def __init__(self, name):
        self._np = np.dtype(name)
        self.name = self._np.name

    dtype = property(lambda self: self._np)
    as_numpy_dtype = property(lambda self: self._np.type)
    is_complex = property(lambda self: self._np.kind == "c")
    is_floating = property(lambda self: self._np.kind == "f")
    is_integer = property(lambda self: self._np.kind in "iu")
    size = property(lambda self: self._np.itemsize)

    @property
    def real_dtype(self):
        return DType({"complex64": "float32", "complex128": "float64"}.get(self.name, self.name))

    def __eq__(self, other):
        try:
            return np.dtype(other._np if isinstance(other, DType) else other) == self._np
        except TypeError:
            return False

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash(self._np)

    def __repr__(self):
        return f"tf.{self.name}"

    def __getattr__(self, name):                                    # NumPy internals ask the ndarray subclass for dtype.type / .kind / ...
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self._np, name)


class Tensor(np.ndarray):
    """ndarray with the few tf.Tensor methods the reference calls."""

    @property
    def shape(self):
        return _Shape(np.ndarray.shape.__get__(self))

    @property
    def dtype(self):
        return DType(np.ndarray.dtype.__get__(self))

    def get_shape(self):
        return self.shape

    def numpy(self):
        return np.asarray(self)

    def __getitem__(self, key):                                  # x[i] of a vector is a 0-d TENSOR (has .numpy()), not a NumPy scalar
        r = np.ndarray.__getitem__(self, key)
        return r if isinstance(r, np.ndarray) else np.asarray(r).view(Tensor)


def _t(x, dtype=None):
    if isinstance(x, RaggedTensor):
        return x
    a = np.asarray(x, dtype=dtype)
    return a.view(Tensor)


class RaggedTensor:
    """[nrows, (ragged), ...] as flat values [nvals, ...] + value_rowids [nvals] (sorted ascending)."""

    def __init__(self, values, rowids, nrows):
        self.flat_values = _t(values)
        self._rowids = np.asarray(rowids, dtype=np.int64)
        self._nrows = int(nrows)
        assert np.all(np.diff(self._rowids) >= 0), "value_rowids must be sorted"

    @classmethod
    def from_value_rowids(cls, values, value_rowids, nrows=None):
        value_rowids = np.asarray(value_rowids)
        if nrows is None:
            nrows = int(value_rowids.max()) + 1 if len(value_rowids) else 0
        return cls(np.asarray(values), value_rowids, nrows)

    values = property(lambda self: self.flat_values)
    dtype = property(lambda self: self.flat_values.dtype)

    @property
    def shape(self):
        return _Shape((self._nrows, None) + tuple(self.flat_values.shape[1:]))

    def value_rowids(self):
        return _t(self._rowids)

    def nrows(self):
        return self._nrows

    def row_lengths(self):
        return np.bincount(self._rowids, minlength=self._nrows)

    def with_flat_values(self, v):
        return RaggedTensor(v, self._rowids, self._nrows)

    def _bin(self, other, fn):
        if isinstance(other, RaggedTensor):
            assert np.array_equal(other._rowids, self._rowids)
            return self.with_flat_values(fn(np.asarray(self.flat_values), np.asarray(other.flat_values)))
        return self.with_flat_values(fn(np.asarray(self.flat_values), other))

    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)
    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __pow__(self, o): return self._bin(o, lambda a, b: a ** b if not (np.isscalar(b) and b == -1) else 1 / a)
    def __neg__(self): return self.with_flat_values(-np.asarray(self.flat_values))
    def __eq__(self, o): return self._bin(o, lambda a, b: a == b)

    # --- reductions over the ragged axis (axis=1), TF-CPU unsorted_segment_* order
    def _reduce(self, ufunc, identity, keepdims):
        v = np.asarray(self.flat_values)
        out = np.full((self._nrows,) + v.shape[1:], identity, dtype=v.dtype)
        lens = self.row_lengths()
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
        for d in range(int(lens.max()) if len(lens) else 0):        # d-th element of every row that has one, in order
            rows = np.nonzero(lens > d)[0]
            out[rows] = ufunc(out[rows], v[starts[rows] + d])
        if keepdims:
            out = out[:, None]
        return _t(out)


def _elementwise(fn):
    def wrapped(x, *a, **k):
        k.pop("name", None)
        if isinstance(x, RaggedTensor):
            return x.with_flat_values(fn(np.asarray(x.flat_values), *a, **k))
        return _t(fn(np.asarray(x), *a, **k))
    return wrapped


def _weak(x):
    """Python scalars stay weakly typed (TF converts them to the other operand's dtype)."""
    return x if isinstance(x, (int, float, complex, bool)) else np.asarray(x)


def _dense(x):
    return np.asarray(x.flat_values) if isinstance(x, RaggedTensor) else np.asarray(x)


def make_tf():
    tf = types.ModuleType("tensorflow")
    tf.__doc__ = "NumPy stand-in (tools/ref_exec/tf_numpy.py)"
    tf.Tensor, tf.RaggedTensor = Tensor, RaggedTensor
    for name in ("float16", "float32", "float64", "int8", "int16", "int32", "int64", "uint8", "complex64", "complex128", "bool"):
        setattr(tf, name, DType(name))
    tf.bfloat16 = type("BF16", (), {"__eq__": lambda self, o: False, "__hash__": lambda self: 0, "__repr__": lambda self: "tf.bfloat16"})()   # (no NumPy twin; only ever compared)
    tf.DType, tf.dtypes = DType, types.SimpleNamespace(DType=DType)
    tf.newaxis = None

    # ---- construction / casting
    def constant(value, dtype=None, shape=None, name=None, **k):
        a = np.array(value, dtype=dtype)
        if dtype is None and not isinstance(value, np.ndarray):         # TF defaults: Python float -> float32, int -> int32
            a = a.astype({"f": np.float32, "i": np.int32, "c": np.complex64}.get(a.dtype.kind, a.dtype))
        return _t(a)
    tf.constant = constant
    tf.convert_to_tensor = lambda value, dtype=None, **k: constant(value, dtype)
    tf.cast = lambda x, dtype, name=None: (x.with_flat_values(np.asarray(x.flat_values).astype(dtype)) if isinstance(x, RaggedTensor)
                                           else _t(np.asarray(x).astype(dtype)))
    tf.zeros = lambda shape, dtype=np.float32, name=None: _t(np.zeros([int(s) for s in np.atleast_1d(shape)], dtype))
    tf.ones = lambda shape, dtype=np.float32, name=None: _t(np.ones([int(s) for s in np.atleast_1d(shape)], dtype))
    tf.zeros_like = _elementwise(np.zeros_like)
    tf.ones_like = _elementwise(np.ones_like)
    def eye(num_rows, num_columns=None, batch_shape=None, dtype=np.float32, name=None):
        e = np.eye(int(num_rows), None if num_columns is None else int(num_columns), dtype=dtype)
        if batch_shape is not None and len(batch_shape):
            e = np.broadcast_to(e, [int(b) for b in batch_shape] + list(e.shape)).copy()
        return _t(e)
    tf.eye = eye
    def _range(*a, start=None, limit=None, delta=None, dtype=None, name=None):
        a = list(a)
        if start is not None or limit is not None:
            a = [0 if start is None else start, limit] + ([delta] if delta is not None else [])
        elif delta is not None:
            a = (a + [None, None])[:2] + [delta] if len(a) == 2 else [0, a[0], delta]
        isf = any(isinstance(v, (float, np.floating)) or (hasattr(v, "dtype") and np.asarray(v).dtype.kind == "f") for v in a)
        if dtype is None:
            dtype = np.float32 if isf else np.int32
        return _t(np.arange(*[np.asarray(v).item() for v in a]).astype(dtype))
    tf.range = _range
    tf.is_tensor = lambda x: isinstance(x, (Tensor, RaggedTensor))
    tf.shape = lambda x, **k: _t(np.array(np.asarray(x).shape, dtype=np.int32))
    tf.rank = lambda x, **k: np.asarray(x).ndim
    tf.executing_eagerly = lambda: True
    tf.size = lambda x: np.asarray(x).size
    tf.identity = lambda x, **k: x
    tf.stop_gradient = lambda x: x
    tf.ensure_shape = lambda x, shape, **k: x
    def _complex(re, im, name=None):
        re, im = np.asarray(re), np.asarray(im)
        out = np.empty(np.broadcast(re, im).shape, np.complex64 if re.dtype == np.float32 else np.complex128)
        out.real, out.imag = re, im
        return _t(out)
    tf.complex = _complex
    tf.split = lambda value, num_or_size_splits, axis=0, **k: [_t(a) for a in (
        np.split(np.asarray(value), num_or_size_splits, axis=axis) if np.isscalar(num_or_size_splits)
        else np.split(np.asarray(value), np.cumsum(num_or_size_splits)[:-1], axis=axis))]

    # ---- shape manipulation (copies: the reference mutates with *= afterwards)
    tf.reshape = lambda x, shape, name=None: _t(np.reshape(np.asarray(x), [int(s) for s in np.atleast_1d(shape)]).copy())
    tf.transpose = lambda x, perm=None, conjugate=False, **k: _t(
        (np.conj(np.transpose(np.asarray(x), perm)) if conjugate else np.transpose(np.asarray(x), perm)).copy())
    tf.expand_dims = lambda x, axis, name=None: _t(np.expand_dims(np.asarray(x), axis))
    tf.squeeze = lambda x, axis=None, name=None: _t(np.squeeze(np.asarray(x), axis=axis if axis is None or np.isscalar(axis) else tuple(axis)))
    tf.concat = lambda values, axis, name=None: _t(np.concatenate([np.asarray(v) for v in values], axis=int(axis)))
    tf.stack = lambda values, axis=0, name=None: _t(np.stack([np.asarray(v) for v in values], axis=axis))
    tf.tile = lambda x, multiples, **k: _t(np.tile(np.asarray(x), [int(m) for m in multiples]))
    tf.broadcast_to = lambda x, shape, **k: _t(np.broadcast_to(np.asarray(x), [int(s) for s in shape]).copy())

    class TensorArray:
        """tf.TensorArray: write() returns the array (the reference rebinds it), read() the stored tensor (PolarBPDecoder)."""

        def __init__(self, dtype, size=0, dynamic_size=False, clear_after_read=True, **k):
            self._items = {}

        def write(self, index, value):
            self._items[int(np.asarray(index))] = _t(np.array(value, copy=True))
            return self

        def read(self, index):
            return self._items[int(np.asarray(index))]
    tf.TensorArray = TensorArray

    def _slice(x, begin, size, name=None):
        x = np.asarray(x)
        idx = tuple(slice(int(b), None if int(s) == -1 else int(b) + int(s)) for b, s in zip(begin, size))
        return _t(x[idx].copy())
    tf.slice = _slice

    def gather(params, indices, validate_indices=None, axis=None, batch_dims=0, name=None):
        axis = int(batch_dims) if axis is None else int(axis)
        if isinstance(indices, RaggedTensor):
            assert axis == 0 and not isinstance(params, RaggedTensor)
            return indices.with_flat_values(np.asarray(params)[np.asarray(indices.flat_values).astype(np.int64)])
        idx = np.asarray(indices).astype(np.int64)
        if isinstance(params, RaggedTensor):                               # select rows (layered schedules)
            assert axis == 0 and idx.ndim == 1
            lens = params.row_lengths()
            starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
            pos = np.concatenate([np.arange(starts[r], starts[r] + lens[r]) for r in idx]) if len(idx) else np.zeros(0, np.int64)
            rid = np.repeat(np.arange(len(idx)), lens[idx])
            return RaggedTensor(np.asarray(params.flat_values)[pos.astype(np.int64)], rid, len(idx))
        p = np.asarray(params)
        if batch_dims:
            bd = int(batch_dims)
            if axis < 0:
                axis += p.ndim
            assert tuple(idx.shape[:bd]) == tuple(p.shape[:bd]), "gather: batch dimensions must agree"
            if axis != bd:                                           # gather along a later axis: bring it next to the batch dims
                res = gather(np.moveaxis(p, axis, bd), idx, axis=bd, batch_dims=bd)
                k = idx.ndim - bd                                    # the gathered dims sit at [bd, bd + k): move them to `axis`
                r = np.asarray(res)
                for j in range(k):
                    r = np.moveaxis(r, bd + k - 1, axis + k - 1 - 0) if False else r
                src = list(range(bd, bd + k))
                dst = list(range(axis, axis + k))
                return _t(np.moveaxis(r, src, dst))
            inner = idx.shape[bd:]
            flat = idx.reshape(p.shape[:bd] + (-1,) + (1,) * (p.ndim - axis - 1))
            flat = np.broadcast_to(flat, p.shape[:bd] + (flat.shape[bd],) + p.shape[axis + 1:])
            out = np.take_along_axis(p, flat, axis=axis)
            return _t(out.reshape(p.shape[:bd] + tuple(inner) + p.shape[axis + 1:]))
        return _t(np.take(p, idx, axis=axis))
    tf.gather = gather

    def tensor_scatter_nd_update(tensor, indices, updates, name=None):
        out = np.array(tensor, copy=True)
        ind = np.asarray(indices)
        assert ind.ndim == 2                                              # [n, depth]: one index tuple per update slice
        out[tuple(ind[:, d] for d in range(ind.shape[1]))] = np.asarray(updates)
        return _t(out)
    tf.tensor_scatter_nd_update = tensor_scatter_nd_update

    def scatter_nd(indices, updates, shape, name=None):
        upd = np.asarray(updates)
        out = np.zeros([int(v) for v in shape], dtype=upd.dtype)
        ind = np.asarray(indices)
        depth = ind.shape[-1]                                             # indices [..., depth]: any number of leading dims
        ind2 = ind.reshape(-1, depth)
        np.add.at(out, tuple(ind2[:, d] for d in range(depth)), upd.reshape((ind2.shape[0],) + out.shape[depth:]))   # duplicates add up, like TF
        return _t(out)
    tf.scatter_nd = scatter_nd

    def tensor_scatter_nd_add(tensor, indices, updates, name=None):
        out = np.array(tensor, copy=True)
        ind = np.asarray(indices)
        depth = ind.shape[-1]
        ind2 = ind.reshape(-1, depth)
        np.add.at(out, tuple(ind2[:, d] for d in range(depth)), np.asarray(updates).reshape((ind2.shape[0],) + out.shape[depth:]))
        return _t(out)
    tf.tensor_scatter_nd_add = tensor_scatter_nd_add

    def argsort(values, axis=-1, direction="ASCENDING", stable=False, name=None):
        v = np.asarray(values)
        # TF sorts DESCENDING as the ascending order of -values; ties keep their index order (stable kernel)
        idx = np.argsort(-v if direction == "DESCENDING" else v, axis=axis, kind="stable")
        return _t(idx.astype(np.int32))
    tf.argsort = argsort
    tf.sort = lambda values, axis=-1, direction="ASCENDING", **k: _t(np.take_along_axis(np.asarray(values), np.asarray(argsort(values, axis, direction)), axis=axis))
    tf.gather_nd = lambda params, indices, batch_dims=0, **k: _t(np.asarray(params)[tuple(np.moveaxis(np.asarray(indices), -1, 0))])

    # ---- elementwise math (IEEE-exact ones first)
    tf.abs = _elementwise(np.abs)
    tf.sign = _elementwise(np.sign)
    tf.negative = _elementwise(np.negative)
    tf.square = _elementwise(np.square)
    tf.sqrt = _elementwise(np.sqrt)
    tf.tanh = _elementwise(np.tanh)
    tf.atanh = _elementwise(np.arctanh)
    tf.exp = _elementwise(np.exp)
    tf.floor = _elementwise(np.floor)
    tf.math = types.SimpleNamespace(
        log=_elementwise(np.log), exp=_elementwise(np.exp), abs=tf.abs, sign=tf.sign, tanh=tf.tanh, atanh=tf.atanh,
        sqrt=tf.sqrt, square=tf.square, real=_elementwise(np.real), imag=_elementwise(np.imag), conj=_elementwise(np.conj),
        reduce_logsumexp=None, log1p=_elementwise(np.log1p), softplus=None,
        divide_no_nan=lambda a, b: _t(np.where(np.asarray(b) == 0, 0, np.asarray(a) / np.where(np.asarray(b) == 0, 1, np.asarray(b))).astype(np.asarray(a).dtype)))
    def _lse(x, axis=None, keepdims=False, name=None):
        x = np.asarray(x)
        m = np.max(x, axis=axis, keepdims=True)
        m = np.where(np.isfinite(m), m, 0).astype(x.dtype)
        r = np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True)) + m     # tf.reduce_logsumexp (math_ops.py): same form
        return _t(r if keepdims else np.squeeze(r, axis=axis))
    tf.reduce_logsumexp = tf.math.reduce_logsumexp = _lse
    tf.math.log_sigmoid = _elementwise(lambda x: -np.logaddexp(np.zeros_like(x), -x))      # -softplus(-x)
    tf.math.softplus = _elementwise(lambda x: np.logaddexp(np.zeros_like(x), x))
    tf.math.softmax = lambda x, axis=-1, **k: _t(np.exp(np.asarray(x) - _lse(x, axis=axis, keepdims=True)))
    tf.nn = types.SimpleNamespace(log_softmax=lambda x, axis=-1, **k: _t(np.asarray(x) - _lse(x, axis=axis, keepdims=True)),
                                  softmax=tf.math.softmax, relu=_elementwise(lambda x: np.maximum(x, 0)))
    tf.sqrt = tf.math.sqrt = _elementwise(np.sqrt)
    tf.math.ceil = _elementwise(np.ceil)
    tf.math.floor = _elementwise(np.floor)
    tf.math.round = tf.round = _elementwise(np.round)                 # (both round half to even)
    tf.math.cos, tf.math.sin = _elementwise(np.cos), _elementwise(np.sin)
    tf.cos, tf.sin = tf.math.cos, tf.math.sin
    tf.math.mod = lambda a, b: _t(np.mod(np.asarray(a), b))
    tf.math.floormod = tf.math.mod

    def _binary(fn):
        def wrapped(a, b, name=None):
            if isinstance(a, RaggedTensor):
                return a._bin(b, fn)
            if isinstance(b, RaggedTensor):
                return b._bin(a, lambda y, x: fn(x, y))
            return _t(fn(_weak(a), _weak(b)))
        return wrapped
    tf.add = _binary(lambda a, b: a + b)
    tf.subtract = _binary(lambda a, b: a - b)
    tf.multiply = _binary(lambda a, b: a * b)
    tf.divide = _binary(lambda a, b: a / b)
    tf.maximum = _binary(np.maximum)
    tf.minimum = _binary(np.minimum)
    tf.equal = _binary(lambda a, b: a == b)
    tf.not_equal = _binary(lambda a, b: a != b)
    tf.less = _binary(lambda a, b: a < b)
    tf.greater = _binary(lambda a, b: a > b)
    tf.greater_equal = _binary(lambda a, b: a >= b)
    tf.less_equal = _binary(lambda a, b: a <= b)
    tf.logical_or = _binary(np.logical_or)
    tf.logical_and = _binary(np.logical_and)
    tf.pow = _binary(lambda a, b: a ** b)
    tf.bitwise = types.SimpleNamespace(bitwise_and=_binary(np.bitwise_and), bitwise_xor=_binary(np.bitwise_xor),
                                       left_shift=_binary(np.left_shift), right_shift=_binary(np.right_shift))
    tf.math.maximum, tf.math.minimum, tf.math.pow = tf.maximum, tf.minimum, tf.pow
    tf.math.multiply, tf.math.add, tf.math.subtract, tf.math.divide = tf.multiply, tf.add, tf.subtract, tf.divide

    def where(cond, x=None, y=None, name=None):
        if x is None and y is None:                                       # tf.where(cond): coordinates of the true entries
            return _t(np.argwhere(_dense(cond)).astype(np.int64))
        c = _dense(cond)
        xs, ys = _dense(x) if not np.isscalar(x) else x, _dense(y) if not np.isscalar(y) else y
        # TF: result dtype = dtype of x and y (both tensors of the same dtype); Python scalars adopt the tensor's dtype
        dt = next((np.asarray(v).dtype for v in (xs, ys) if not np.isscalar(v)),
                  np.bool_ if isinstance(xs, (bool, np.bool_)) and isinstance(ys, (bool, np.bool_)) else
                  np.int32 if isinstance(xs, int) and isinstance(ys, int) else np.float32)
        return _t(np.where(c, np.asarray(xs, dtype=dt) if np.isscalar(xs) else xs, np.asarray(ys, dtype=dt) if np.isscalar(ys) else ys))
    tf.where = where

    def clip_by_value(t, clip_value_min, clip_value_max, name=None):
        # TF: maximum(minimum(t, max), min)
        f = lambda v: np.maximum(np.minimum(v, np.asarray(clip_value_max, dtype=v.dtype)), np.asarray(clip_value_min, dtype=v.dtype))
        if isinstance(t, RaggedTensor):
            return t.with_flat_values(f(np.asarray(t.flat_values)))
        return _t(f(np.asarray(t)))
    tf.clip_by_value = clip_by_value

    # ---- reductions
    def _reduction(np_fn, ufunc, identity_of):
        def red(x, axis=None, keepdims=False, name=None):
            if isinstance(x, RaggedTensor):
                assert axis == 1, "only the ragged axis is reduced by the reference"
                return x._reduce(ufunc, identity_of(np.asarray(x.flat_values).dtype), keepdims)
            ax = axis if axis is None or np.isscalar(axis) else tuple(int(a) for a in axis)
            return _t(np_fn(np.asarray(x), axis=ax, keepdims=keepdims))
        return red
    tf.reduce_sum = _reduction(np.sum, np.add, lambda dt: 0)
    tf.reduce_prod = _reduction(np.prod, np.multiply, lambda dt: 1)
    tf.reduce_min = _reduction(np.min, np.minimum, lambda dt: np.finfo(dt).max if dt.kind == "f" else np.iinfo(dt).max)
    tf.reduce_max = _reduction(np.max, np.maximum, lambda dt: np.finfo(dt).min if dt.kind == "f" else np.iinfo(dt).min)
    tf.reduce_mean = lambda x, axis=None, keepdims=False, name=None: _t(np.mean(np.asarray(x), axis=axis, keepdims=keepdims))
    tf.reduce_any = lambda x, axis=None, **k: _t(np.any(np.asarray(x), axis=axis))
    tf.reduce_all = lambda x, axis=None, **k: _t(np.all(np.asarray(x), axis=axis))
    tf.argmax = lambda x, axis=None, output_type=np.int64, **k: _t(np.argmax(np.asarray(x), axis=axis).astype(output_type))
    tf.argmin = lambda x, axis=None, output_type=np.int64, **k: _t(np.argmin(np.asarray(x), axis=axis).astype(output_type))

    # ---- ragged namespace
    def map_flat_values(op, *args, **kwargs):
        first = next(a for a in list(args) + list(kwargs.values()) if isinstance(a, RaggedTensor))
        flat = lambda a: _t(np.asarray(a.flat_values)) if isinstance(a, RaggedTensor) else a
        out = op(*[flat(a) for a in args], **{k: flat(v) for k, v in kwargs.items()})
        return first.with_flat_values(np.asarray(out))
    tf.ragged = types.SimpleNamespace(map_flat_values=map_flat_values)

    # ---- control flow / misc
    def while_loop(cond, body, loop_vars, maximum_iterations=None, **k):
        v, n = tuple(loop_vars), 0
        while bool(cond(*v)) and (maximum_iterations is None or n < int(maximum_iterations)):
            v, n = tuple(body(*v)), n + 1
        return v
    tf.while_loop = while_loop
    tf.cond = lambda pred, true_fn, false_fn, **k: true_fn() if bool(np.asarray(pred)) else false_fn()
    tf.fill = lambda dims, value, **k: _t(np.full([int(d) for d in np.atleast_1d(dims)], value, dtype=np.float32 if isinstance(value, float) else None))

    def function(func=None, **k):
        return (lambda f: f) if func is None else func
    tf.function = function

    def py_function(func, inp, Tout, name=None):
        r = func(*[_t(np.asarray(a)) for a in inp])
        if isinstance(Tout, (list, tuple)):
            return [_t(np.asarray(v).astype(t._np if isinstance(t, DType) else np.dtype(t))) for v, t in zip(r, Tout)]
        return _t(np.asarray(r))
    tf.py_function = py_function
    class _Debugging:
        @staticmethod
        def assert_equal(a, b, message=None, **k):
            if not np.all(np.asarray(a) == np.asarray(b)):
                raise AssertionError(message or "assert_equal")

        def __getattr__(self, name):                                # the other tf.debugging.assert_* are argument checks
            if name.startswith("assert_"):
                return lambda *a, **k: None
            raise AttributeError(name)
    tf.debugging = _Debugging()
    def _lstsq(matrix, rhs, l2_regularizer=0.0, fast=True, name=None):
        """tf.linalg.lstsq(fast=False): the minimum-norm least-squares solution through a complete orthogonal
        decomposition - NumPy's pseudo-inverse (SVD) in the operands' precision."""
        a, b = np.asarray(matrix), np.asarray(rhs)
        return _t(np.matmul(np.linalg.pinv(a), b).astype(a.dtype))
    tf.linalg = types.SimpleNamespace(
        lstsq=_lstsq,
        cholesky=lambda a: _t(np.linalg.cholesky(np.asarray(a))),
        matmul=lambda a, b, adjoint_a=False, adjoint_b=False, transpose_a=False, transpose_b=False, **k: tf.matmul(
            a, b, adjoint_a=adjoint_a, adjoint_b=adjoint_b, transpose_a=transpose_a, transpose_b=transpose_b),
        diag_part=lambda a: _t(np.diagonal(np.asarray(a), axis1=-2, axis2=-1).copy()),
        diag=lambda d: _t(np.asarray(d)[..., None] * np.eye(np.asarray(d).shape[-1], dtype=np.asarray(d).dtype)),
        adjoint=lambda a: _t(np.conj(np.swapaxes(np.asarray(a), -1, -2))),
        inv=lambda a: _t(np.linalg.inv(np.asarray(a))),
        triangular_solve=lambda matrix, rhs, lower=True, adjoint=False, **k: _t(_tri_solve(np.asarray(matrix), np.asarray(rhs), lower, adjoint)),
        cholesky_solve=lambda chol, rhs, **k: _t(_tri_solve(np.asarray(chol), _tri_solve(np.asarray(chol), np.asarray(rhs), True, False), True, True)),
        eye=tf.eye, matvec=lambda a, b, **k: _t(np.einsum("...ij,...j->...i", np.asarray(a), np.asarray(b))))

    def matmul(a, b, transpose_a=False, transpose_b=False, adjoint_a=False, adjoint_b=False, **k):
        a, b = np.asarray(a), np.asarray(b)
        if transpose_a: a = np.swapaxes(a, -1, -2)
        if transpose_b: b = np.swapaxes(b, -1, -2)
        if adjoint_a: a = np.conj(np.swapaxes(a, -1, -2))
        if adjoint_b: b = np.conj(np.swapaxes(b, -1, -2))
        return _t(np.matmul(a, b))
    tf.matmul = matmul

    def _qr(a, full_matrices=False, **k):
        q, r = np.linalg.qr(np.asarray(a), mode="complete" if full_matrices else "reduced")
        return _t(q), _t(r)
    tf.linalg.qr = _qr

    def _top_k(x, k=1, sorted=True, **kw):                          # values / indices of the k largest entries of the last axis
        x = np.asarray(x)
        idx = np.argsort(-x, axis=-1, kind="stable")[..., :int(k)]
        return _t(np.take_along_axis(x, idx, -1)), _t(idx.astype(np.int32))
    tf.math.top_k = tf.nn.top_k = _top_k
    # ---- signal / misc array ops used by the OFDM time-domain path
    tf.signal = types.SimpleNamespace(
        fft=lambda x, **k: _t(np.fft.fft(np.asarray(x), axis=-1).astype(np.asarray(x).dtype)),
        ifft=lambda x, **k: _t(np.fft.ifft(np.asarray(x), axis=-1).astype(np.asarray(x).dtype)),
        fftshift=lambda x, axes=None, **k: _t(np.fft.fftshift(np.asarray(x), axes=axes)),
        ifftshift=lambda x, axes=None, **k: _t(np.fft.ifftshift(np.asarray(x), axes=axes)))

    def pad(tensor, paddings, mode="CONSTANT", constant_values=0, name=None):
        return _t(np.pad(np.asarray(tensor), [(int(a), int(b)) for a, b in np.asarray(paddings)], constant_values=constant_values))
    tf.pad = pad
    tf.repeat = lambda x, repeats, axis=None, **k: _t(np.repeat(np.asarray(x), repeats, axis=axis))
    tf.reverse = lambda x, axis, **k: _t(np.flip(np.asarray(x), axis=tuple(int(a) for a in np.atleast_1d(axis))))
    tf.linspace = lambda start, stop, num, **k: _t(np.linspace(start, stop, int(num)).astype(
        np.asarray(start).dtype if np.asarray(start).dtype.kind == "f" and not isinstance(start, float) else np.float32))
    tf.math.cumsum = tf.cumsum = lambda x, axis=0, **k: _t(np.cumsum(np.asarray(x), axis=axis))
    tf.experimental = types.SimpleNamespace(numpy=types.SimpleNamespace(
        swapaxes=lambda x, a, b: _t(np.swapaxes(np.asarray(x), a, b)),
        sinc=_elementwise(lambda x: np.sinc(x).astype(x.dtype)),
        log10=_elementwise(np.log10), log2=_elementwise(np.log2)))
    # ---- ops of the 3GPP channel-model code (tr38901/*.py)
    tf.acos = tf.math.acos = _elementwise(np.arccos)
    tf.asin = tf.math.asin = _elementwise(np.arcsin)
    tf.atan2 = tf.math.atan2 = _binary(np.arctan2)
    tf.math.angle = _elementwise(lambda z: np.angle(z).astype(np.float32 if z.dtype == np.complex64 else np.float64))
    tf.math.log10 = _elementwise(np.log10)
    tf.roll = lambda x, shift, axis, **k: _t(np.roll(np.asarray(x), shift, axis))
    tf.meshgrid = lambda *xs, indexing="xy", **k: [_t(a) for a in np.meshgrid(*[np.asarray(x) for x in xs], indexing=indexing)]
    tf.logical_not = tf.math.logical_not = _elementwise(np.logical_not)
    tf.einsum = lambda eq, *ops, **k: _t(np.einsum(eq, *[np.asarray(o) for o in ops]))
    tf.norm = lambda x, axis=None, keepdims=False, **k: _t(np.linalg.norm(np.asarray(x), axis=axis, keepdims=keepdims))
    tf.cumsum = tf.math.cumsum

    class _Generator:
        """config.tf_rng stand-in: NumPy draws of the same distributions (NOT TensorFlow's stream)."""
        def __init__(self, seed=0):
            self._g = np.random.default_rng(seed)
        def uniform(self, shape, minval=0.0, maxval=1.0, dtype=np.float32, **k):
            if np.dtype(dtype).kind in "iu":
                return _t(self._g.integers(int(minval), int(maxval), [int(v) for v in np.atleast_1d(shape)]).astype(dtype))
            lo, hi = np.asarray(minval, np.float64), np.asarray(maxval, np.float64)
            return _t((lo + (hi - lo) * self._g.random([int(v) for v in np.atleast_1d(shape)])).astype(dtype))
        def normal(self, shape, mean=0.0, stddev=1.0, dtype=np.float32, **k):
            return _t((mean + stddev * self._g.normal(size=[int(v) for v in np.atleast_1d(shape)])).astype(dtype))
    tf.random = types.SimpleNamespace(Generator=_Generator)
    tf.name_scope = lambda *a, **k: _NullCtx()
    class Variable(Tensor):
        """tf.Variable stand-in (an ndarray view; ``isinstance(x, tf.Variable)`` is False for plain tensors)."""
        def __new__(cls, initial_value, *a, dtype=None, **k):
            return np.array(initial_value, dtype=dtype).view(cls)

        def scatter_nd_add(self, indices, updates, **k):         # in place, like the TF variable method
            np.add.at(np.asarray(self), tuple(np.asarray(indices).reshape(-1, np.asarray(indices).shape[-1]).T), np.asarray(updates))
            return self
    tf.Variable = Variable
    tf.keras = types.SimpleNamespace(layers=types.SimpleNamespace(Layer=object))
    tf.config = types.SimpleNamespace(list_physical_devices=lambda *a: [], list_logical_devices=lambda *a: [])
    tf.math.is_nan = lambda x, **k: _t(np.isnan(np.asarray(x)))
    tf.math.is_inf = lambda x, **k: _t(np.isinf(np.asarray(x)))
    for _n in ("greater", "greater_equal", "less", "less_equal", "equal", "not_equal", "logical_or", "logical_and"):
        setattr(tf.math, _n, getattr(tf, _n))
    for _n in ("reduce_sum", "reduce_prod", "reduce_min", "reduce_max", "reduce_mean", "reduce_any", "reduce_all", "argmax", "argmin"):
        if hasattr(tf, _n):
            setattr(tf.math, _n, getattr(tf, _n))
    return tf


def _tri_solve(l, rhs, lower, adjoint):
    """Batched triangular solve by substitution in the matrices' own dtype (TF: MatrixTriangularSolve -> Eigen
    triangularView.solve; same recurrences, accumulation order may differ in the last bit)."""
    if adjoint:
        l, lower = np.conj(np.swapaxes(l, -1, -2)), not lower
    n = l.shape[-1]
    x = np.zeros(np.broadcast_shapes(l.shape[:-2], rhs.shape[:-2]) + rhs.shape[-2:], dtype=np.result_type(l, rhs))
    order = range(n) if lower else range(n - 1, -1, -1)
    for i in order:
        acc = rhs[..., i, :].astype(x.dtype)
        js = range(i) if lower else range(n - 1, i, -1)
        for j in js:
            acc = acc - l[..., i, j, None] * x[..., j, :]
        x[..., i, :] = acc / l[..., i, i, None]
    return x


class _NullCtx:
    def __enter__(self): return self
    def __exit__(self, *a): return False
