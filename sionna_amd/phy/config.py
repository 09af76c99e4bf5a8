"""Global configuration singleton - mirror of ``sionna.phy.config``
(reference src/sionna/phy/config.py:34-201): ``config.seed``, ``config.precision``,
``config.np_rng`` / ``config.py_rng`` and, instead of ``tf_rng``, the device generator
``config.rng`` (counter-based Philox stream, see oracle/utils.py for its specification).
"""
import os
import random

import numpy as np
import torch

dtypes = {
    "single": {"np": {"rdtype": np.float32, "cdtype": np.complex64},
               "torch": {"rdtype": torch.float32, "cdtype": torch.complex64}},
    "double": {"np": {"rdtype": np.float64, "cdtype": np.complex128},
               "torch": {"rdtype": torch.float64, "cdtype": torch.complex128}},
}

_MASK64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


class PhiloxGenerator:
    """Host-side state of a device random stream: (seed, call counter).

    Each kernel launch that draws random numbers consumes one ``call`` value; the element
    index inside the launch is the Philox counter.  ``rank`` decorrelates the streams of
    the data-parallel replicas (a tested requirement of the reference,
    test/unit/utils/test_utils.py:112-127).
    """

    def __init__(self, seed=None, rank=None):
        if rank is None:
            rank = int(os.environ.get("RANK", "0"))
        self.rank = rank
        self.reset(seed)

    def reset(self, seed=None):
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        self.base_seed = int(seed) & _MASK64
        # rank 0 keeps the user's seed (single-GPU runs reproduce the oracle stream)
        self.seed = self.base_seed if self.rank == 0 else _splitmix64(self.base_seed ^ _splitmix64(self.rank))
        self.call = 0

    def next_call(self):
        c = self.call
        self.call += 1
        return c


class Config:
    _instance = None

    def __new__(cls):
        if cls._instance is None:
            cls._instance = object.__new__(cls)
        return cls._instance

    def __init__(self):
        self._seed = None
        self._py_rng = None
        self._np_rng = None
        self._rng = None
        self._precision = None
        self.precision = "single"

    @property
    def py_rng(self):
        if self._py_rng is None:
            self._py_rng = random.Random()
        return self._py_rng

    @property
    def np_rng(self):
        if self._np_rng is None:
            self._np_rng = np.random.default_rng()
        return self._np_rng

    @property
    def rng(self):
        """Device random stream (replaces ``tf_rng``)."""
        if self._rng is None:
            self._rng = PhiloxGenerator(self._seed)
        return self._rng

    tf_rng = rng  # notebooks that only pass it around keep working

    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, seed):
        if seed is not None:
            seed = int(seed)
        self._seed = seed
        self.rng.reset(seed)
        self.py_rng.seed(seed)
        self._np_rng = np.random.default_rng(seed)

    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, v):
        if v not in ["single", "double"]:
            raise ValueError("Precision must be ``single`` or ``double``.")
        self._precision = v

    @property
    def np_rdtype(self):
        return dtypes[self.precision]["np"]["rdtype"]

    @property
    def np_cdtype(self):
        return dtypes[self.precision]["np"]["cdtype"]

    @property
    def rdtype(self):
        return dtypes[self.precision]["torch"]["rdtype"]

    @property
    def cdtype(self):
        return dtypes[self.precision]["torch"]["cdtype"]

    tf_rdtype = rdtype
    tf_cdtype = cdtype

    @property
    def device(self):
        from .. import _ffi
        return _ffi.device()


config = Config()
