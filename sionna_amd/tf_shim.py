"""A minimal stand-in for the ``tensorflow`` module, for BER notebooks that decorate their models with
``@tf.function`` and wrap scalars in ``tf.constant``.

This is host-side glue for running an existing notebook on ``sionna_amd`` WITHOUT TensorFlow installed - not a
TensorFlow compatibility layer: no kernel goes through it.  ``function`` is an identity decorator (the blocks launch
pre-compiled HIP kernels, there is nothing to trace), ``constant`` / ``cast`` / ``convert_to_tensor`` pass Python
scalars through and turn arrays into torch tensors, dtypes map to torch dtypes, and the device-configuration calls
notebooks make at the top (``tf.config...``, ``tf.get_logger().setLevel``) are accepted and ignored.
``sionna_amd.install_as_sionna(tf_shim=True)`` registers it as ``tensorflow`` when the real package is absent."""
import types

import numpy as np
import torch

float16, float32, float64 = torch.float16, torch.float32, torch.float64
int8, int16, int32, int64, uint8 = torch.int8, torch.int16, torch.int32, torch.int64, torch.uint8
complex64, complex128 = torch.complex64, torch.complex128
bool = torch.bool  # pylint: disable=redefined-builtin
dtypes = types.SimpleNamespace(float32=float32, float64=float64, int32=int32, int64=int64, complex64=complex64,
                               complex128=complex128)
newaxis = None


def function(func=None, **_kwargs):
    """``@tf.function``, ``@tf.function()``, ``@tf.function(jit_compile=True)`` and ``tf.function(model, ...)``."""
    if func is None:
        return lambda f: f
    return func


def _scalar_or_tensor(value, dtype=None):
    if isinstance(value, torch.Tensor):
        return value.to(dtype) if dtype is not None else value
    if isinstance(value, (int, float, complex, np.number)) or (isinstance(value, np.ndarray) and value.ndim == 0):
        v = value.item() if isinstance(value, (np.number, np.ndarray)) else value
        if dtype in (int8, int16, int32, int64, uint8):
            return int(v)
        if dtype in (float16, float32, float64):
            return float(v)
        return v
    return torch.as_tensor(np.asarray(value), dtype=dtype)


def constant(value, dtype=None, shape=None, name=None):  # pylint: disable=unused-argument
    return _scalar_or_tensor(value, dtype)


def convert_to_tensor(value, dtype=None, **_kwargs):
    return _scalar_or_tensor(value, dtype)


def cast(x, dtype):
    return _scalar_or_tensor(x, dtype)


def zeros(shape, dtype=float32):
    return torch.zeros(tuple(int(s) for s in shape), dtype=dtype)


def ones(shape, dtype=float32):
    return torch.ones(tuple(int(s) for s in shape), dtype=dtype)


def zeros_like(x, dtype=None):
    return torch.zeros_like(torch.as_tensor(x), dtype=dtype)


def ones_like(x, dtype=None):
    return torch.ones_like(torch.as_tensor(x), dtype=dtype)


def shape(x):
    return tuple(x.shape)


def rank(x):
    return len(x.shape)


def is_tensor(x):
    return isinstance(x, torch.Tensor)


def reshape(x, shape):  # pylint: disable=redefined-outer-name
    return torch.as_tensor(x).reshape(tuple(int(s) for s in shape))


class _Logger:
    def setLevel(self, *_a):  # noqa: N802  (TensorFlow's spelling)
        return None


def get_logger():
    return _Logger()


class _Experimental:
    @staticmethod
    def set_memory_growth(*_a, **_k):
        return None


class _Config:
    experimental = _Experimental()

    @staticmethod
    def list_physical_devices(kind=None):  # pylint: disable=unused-argument
        return []

    @staticmethod
    def set_visible_devices(*_a, **_k):
        return None

    @staticmethod
    def run_functions_eagerly(*_a, **_k):
        return None


config = _Config()


class _Errors:
    InvalidArgumentError = ValueError


errors = _Errors()
__version__ = "0.0-sionna_amd-shim"
