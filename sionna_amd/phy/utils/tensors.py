"""Tensor shape helpers - mirrors of reference src/sionna/phy/utils/tensors.py:9-211
(``expand_to_rank``, ``flatten_dims``, ``flatten_last_dims``, ``insert_dims``, ``split_dim``) and the
``log2`` / ``log10`` / ``db`` conveniences of utils/misc.py.  Pure views / reshapes of torch tensors."""
import numpy as np
import torch


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))


def expand_to_rank(tensor, target_rank, axis=-1):
    """Insert singleton dimensions at ``axis`` until the tensor has ``target_rank`` dimensions."""
    tensor = _t(tensor)
    return insert_dims(tensor, max(target_rank - tensor.dim(), 0), axis)


def insert_dims(tensor, num_dims, axis=-1):
    """Insert ``num_dims`` singleton dimensions starting at ``axis``."""
    tensor = _t(tensor)
    assert num_dims >= 0, "`num_dims` must be nonnegative."
    rank = tensor.dim()
    assert -(rank + 1) <= axis <= rank, "`axis` is out of range `[-(D+1), D]`)"
    axis = axis if axis >= 0 else rank + axis + 1
    shape = tuple(tensor.shape)
    return tensor.reshape(shape[:axis] + (1,) * num_dims + shape[axis:])


def flatten_dims(tensor, num_dims, axis):
    """Merge ``num_dims`` dimensions starting at ``axis`` into one."""
    tensor = _t(tensor)
    assert num_dims >= 2, "`num_dims` must be >= 2"
    assert 0 <= axis <= tensor.dim() - 1, "0<= `axis` <= rank(tensor)-1"
    assert num_dims + axis <= tensor.dim(), "`num_dims`+`axis` <= rank(`tensor`)"
    shape = tuple(tensor.shape)
    return tensor.reshape(shape[:axis] + (-1,) + shape[axis + num_dims:])


def flatten_last_dims(tensor, num_dims=2):
    """Merge the last ``num_dims`` dimensions into one."""
    tensor = _t(tensor)
    assert num_dims >= 2, "`num_dims` must be >= 2"
    assert num_dims <= tensor.dim(), "`num_dims` must <= rank(`tensor`)"
    return tensor.reshape(tuple(tensor.shape[:-num_dims]) + (-1,))


def split_dim(tensor, shape, axis):
    """Reshape dimension ``axis`` into the dimensions ``shape``."""
    tensor = _t(tensor)
    assert 0 <= axis <= tensor.dim() - 1, "0<= `axis` <= rank(tensor)-1"
    s = tuple(tensor.shape)
    return tensor.reshape(s[:axis] + tuple(int(v) for v in shape) + s[axis + 1:])


def log2(x):
    return torch.log2(_t(x))


def log10(x):
    return torch.log10(_t(x))


def db(x):
    """10 log10(x)."""
    return 10.0 * torch.log10(_t(x))
