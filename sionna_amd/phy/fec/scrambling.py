"""``Scrambler``, ``TB5GScrambler``, ``Descrambler`` - mirrors of reference
src/sionna/phy/fec/scrambling.py:10-579 on ``samd_scramble_f32`` / ``samd_nr_prng_seq_f32``.

The reference's random ``Scrambler`` draws from TensorFlow's stateless RNG; here the sequence is
the library's Philox bit stream ``BinarySource(seed, call=1337)`` (specification:
oracle/scrambling.py::random_scrambling_sequence), so scrambler / descrambler pairs and explicit
seeds behave as in the reference while the realisations differ."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, wrap
from ..config import config

_CALL = 1337          # the reference seeds with (1337, seed), scrambling.py:124


def _apply(x, seq, period, binary, rdtype=torch.float32):
    x = _ffi.to_device(x, rdtype)
    out = torch.empty_like(x)
    if x.numel() == 0:
        return wrap(out)
    fn = _ffi.lib().samd_scramble_f64 if rdtype == torch.float64 else _ffi.lib().samd_scramble_f32
    _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(seq), x.numel(), int(period), int(bool(binary)), _ffi.ptr(out), _ffi.stream()),
               "scramble")
    return wrap(out)


def _check_binary_flag(binary):
    if not isinstance(binary, (bool, np.bool_)):
        raise TypeError("binary must be bool.")
    return bool(binary)


class Scrambler(Block):
    """``Scrambler(seed=None, keep_batch_constant=False, binary=True, sequence=None,
    keep_state=True)(x, seed=None, binary=None)``."""

    def __init__(self, seed=None, keep_batch_constant=False, binary=True, sequence=None, keep_state=True,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(keep_batch_constant, bool):
            raise TypeError("keep_batch_constant must be bool.")
        self._keep_batch_constant = keep_batch_constant
        if seed is not None:
            if sequence is not None:
                print("Note: explicit scrambling sequence provided. Seed will be ignored.")
            if not isinstance(seed, int):
                raise TypeError("seed must be int.")
        else:
            seed = int(config.np_rng.uniform(0, 2 ** 31 - 1))
        self._binary = _check_binary_flag(binary)
        if not isinstance(keep_state, bool):
            raise TypeError("keep_state must be bool.")
        self._keep_state = keep_state
        self._seed = seed
        self._sequence = None
        if sequence is not None:
            seq = np.asarray(sequence.cpu() if isinstance(sequence, torch.Tensor) else sequence, np.float32)
            if not np.all((seq == 0) | (seq == 1)):
                raise ValueError("Scrambling sequence must be binary.")
            self._sequence = seq

    seed = property(lambda self: self._seed)
    keep_state = property(lambda self: self._keep_state)
    sequence = property(lambda self: self._sequence)

    def _random_sequence(self, shape, seed):
        shp = tuple(shape[1:]) if self._keep_batch_constant else tuple(shape)
        n = int(np.prod(shp)) if len(shp) else 1
        seq = torch.empty(max(n, 1), dtype=torch.float32, device=_ffi.device())
        _ffi.check(_ffi.lib().samd_binary_source_f32(int(seed) & (2 ** 64 - 1), _CALL, n, _ffi.ptr(seq), _ffi.stream()),
                   "Scrambler sequence")
        return seq, n

    def call(self, x, seed=None, binary=None):
        return self._run(x, seed, binary, self.rdtype)

    def _run(self, x, seed, binary, rdtype):
        """the scrambling itself in the I/O precision `rdtype` - a Descrambler of another precision hands in its own
        (scrambling.py:573-579 casts to it) instead of changing anything on this block, which an encoder may share"""
        binary = self._binary if binary is None else _check_binary_flag(binary)
        x = _ffi.to_device(x, rdtype)
        if seed is not None:
            s = int(seed)
        elif self._keep_state:
            s = self._seed
        else:
            s = int(config.np_rng.integers(0, 2 ** 31 - 1))
        if self._sequence is not None:
            seq = self._sequence
            while seq.ndim > 0 and seq.shape[0] == 1:          # leading broadcast dims carry no data
                seq = seq[0]
            if seq.ndim > x.dim():
                raise ValueError("sequence has more dimensions than the input")
            seq = np.broadcast_to(seq, tuple(x.shape)[x.dim() - seq.ndim:]) if seq.ndim else seq.reshape(1)
            seq_d = _ffi.to_device(np.ascontiguousarray(seq, np.float32).reshape(-1), torch.float32)
            return _apply(x, seq_d, seq_d.numel(), binary, rdtype)
        seq_d, period = self._random_sequence(x.shape, s)
        return _apply(x, seq_d, period, binary, rdtype)


class TB5GScrambler(Block):
    """``TB5GScrambler(n_rnti=1, n_id=1, binary=True, channel_type="PUSCH", codeword_index=0)``
    - 38.211 Sec. 6.3.1.1 / 7.3.1.1 scrambling; lists of (n_rnti, n_id) scramble axis -2 per stream."""

    def __init__(self, n_rnti=1, n_id=1, binary=True, channel_type="PUSCH", codeword_index=0, precision=None,
                 **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(binary, bool):
            raise TypeError("binary must be bool.")
        self._binary = binary
        if channel_type not in ("PDSCH", "PUSCH"):
            raise TypeError("Unsupported channel_type.")
        if codeword_index not in (0, 1):
            raise ValueError("codeword_index must be 0 or 1.")
        if isinstance(n_rnti, (list, tuple)):
            if not isinstance(n_id, (list, tuple)):
                raise TypeError("n_id must be a list of same length as n_rnti.")
            if len(n_rnti) != len(n_id):
                raise ValueError("n_rnti and n_id must be of same length.")
            self._multi_stream = True
            n_rnti, n_id = list(n_rnti), list(n_id)
        else:
            n_rnti, n_id = [n_rnti], [n_id]
            self._multi_stream = False
        for idx, (nr, ni) in enumerate(zip(n_rnti, n_id)):
            if not nr % 1 == 0:
                raise ValueError("n_rnti must be integer.")
            if nr not in range(2 ** 16):
                raise ValueError("n_rnti must be in [0, 65535].")
            n_rnti[idx] = int(nr)
            if not ni % 1 == 0:
                raise ValueError("n_rnti must be integer.")
            if ni not in range(2 ** 10):
                raise ValueError("n_id must be in [0, 1023].")
            n_id[idx] = int(ni)
        if channel_type == "PUSCH":
            self._c_init = [nr * 2 ** 15 + ni for nr, ni in zip(n_rnti, n_id)]
        else:
            self._c_init = [nr * 2 ** 15 + codeword_index * 2 ** 14 + ni for nr, ni in zip(n_rnti, n_id)]
        self._seq_len = None
        self._sequence = None

    keep_state = property(lambda self: True)

    def _build_sequence(self, n):
        seq = torch.empty((len(self._c_init), n), dtype=torch.float32, device=_ffi.device())
        for i, c in enumerate(self._c_init):
            _ffi.check(_ffi.lib().samd_nr_prng_seq_f32(c, n, _ffi.ptr(seq[i]), _ffi.stream()), "generate_prng_seq")
        self._sequence, self._seq_len = seq, n

    def call(self, x, /, *, binary=None):
        return self._run(x, None, binary, self.rdtype)

    def _run(self, x, seed, binary, rdtype):               # (see Scrambler._run; the sequence of 38.211 has no seed)
        binary = self._binary if binary is None else _check_binary_flag(binary)
        x = _ffi.to_device(x, rdtype)
        if self._multi_stream:
            assert x.dim() >= 2 and x.shape[-2] == len(self._c_init), \
                "Dimension of axis=-2 must be equal to len(n_rnti)."
        if self._seq_len != x.shape[-1]:
            self._build_sequence(int(x.shape[-1]))
        return _apply(x, self._sequence, self._sequence.numel(), binary, rdtype)


class Descrambler(Block):
    """``Descrambler(scrambler, binary=True)(x, seed=None)``: re-applies the scrambler's sequence."""

    def __init__(self, scrambler, binary=True, precision=None, **kwargs):
        if not isinstance(scrambler, (Scrambler, TB5GScrambler)):
            raise TypeError("scrambler must be an instance of Scrambler.")
        self._scrambler = scrambler
        super().__init__(precision=scrambler.precision if precision is None else precision, **kwargs)
        self._binary = _check_binary_flag(binary)
        if scrambler.keep_state is False:
            print("Warning: scrambler uses random sequences that cannot be access by descrambler. Please use "
                  "keep_state=True and provide explicit random seed as input to call function.")

    scrambler = property(lambda self: self._scrambler)

    def call(self, x, /, *, seed=None):
        scr = self._scrambler
        s = (seed if seed is not None else scr.seed) if isinstance(scr, Scrambler) else None
        if isinstance(s, torch.Tensor):
            s = int(s)
        return scr._run(x, s, self._binary, self.rdtype)   # the descrambler's own precision (scrambling.py:573-579 casts to it)
