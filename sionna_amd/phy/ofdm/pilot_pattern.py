"""Pilot patterns - mirror of reference src/sionna/phy/ofdm/pilot_pattern.py:17-378.
Host-side (NumPy) objects; masks and pilots are uploaded by the blocks that use them."""
import numpy as np
import torch

from ..block import Object
from ..config import dtypes
from ..mapping import qam
from ... import _ffi


class PilotPattern(Object):
    """``PilotPattern(mask [tx,s,T,F_eff] bool, pilots [tx,s,num_pilots], normalize=False)``"""

    def __init__(self, mask, pilots, normalize=False, precision=None):
        super().__init__(precision=precision)
        self._mask = np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask).astype(bool)
        self.pilots = pilots
        self.normalize = normalize
        self._check_settings()

    num_tx = property(lambda self: self._mask.shape[0])
    num_streams_per_tx = property(lambda self: self._mask.shape[1])
    num_ofdm_symbols = property(lambda self: self._mask.shape[2])
    num_effective_subcarriers = property(lambda self: self._mask.shape[3])
    num_pilot_symbols = property(lambda self: self._pilots.shape[-1])
    mask = property(lambda self: self._mask)

    @property
    def num_data_symbols(self):
        return self._mask.shape[-1] * self._mask.shape[-2] - self.num_pilot_symbols

    @property
    def normalize(self):
        return self._normalize

    @normalize.setter
    def normalize(self, value):
        self._normalize = bool(value)

    @property
    def pilots(self):
        """[num_tx, num_streams_per_tx, num_pilots] complex ndarray, normalised to unit mean energy
        per sequence if ``normalize`` (pilot_pattern.py:119-132)."""
        p = self._pilots
        if self._normalize and p.shape[-1] > 0:
            scale = 1 / np.sqrt(np.mean(np.abs(p) ** 2, axis=-1, keepdims=True))
            p = (scale * p).astype(p.dtype)
        return p

    @pilots.setter
    def pilots(self, v):
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        self._pilots = np.asarray(v).astype(dtypes[self.precision]["np"]["cdtype"])

    def _check_settings(self):
        assert self._mask.ndim == 4, "`mask` must have four dimensions."
        assert self._pilots.ndim == 3, "`pilots` must have three dimensions."
        assert self._mask.shape[:2] == self._pilots.shape[:2], \
            "The first two dimensions of `mask` and `pilots` must be equal."
        num_pilots = self._mask.sum(axis=(-2, -1))
        assert num_pilots.min() == num_pilots.max(), \
            "The number of nonzero elements in the masks for all transmitters and streams must be identical."
        assert self.num_pilot_symbols == num_pilots.max(), \
            "The shape of the last dimension of `pilots` must equal the number of non-zero entries within the last two dimensions of `mask`."
        return True


class EmptyPilotPattern(PilotPattern):
    def __init__(self, num_tx, num_streams_per_tx, num_ofdm_symbols, num_effective_subcarriers, precision=None):
        assert num_tx > 0, "`num_tx` must be positive`."
        assert num_streams_per_tx > 0, "`num_streams_per_tx` must be positive`."
        assert num_ofdm_symbols > 0, "`num_ofdm_symbols` must be positive`."
        assert num_effective_subcarriers > 0, "`num_effective_subcarriers` must be positive`."
        shape = [num_tx, num_streams_per_tx, num_ofdm_symbols, num_effective_subcarriers]
        super().__init__(np.zeros(shape, bool), np.zeros(shape[:2] + [0], np.complex64), normalize=False,
                         precision=precision)


def _qpsk_sequence(seed, call, n):
    """n QPSK symbols on the device Philox stream (seed, call) - the reference draws the Kronecker
    pilots from ``QAMSource(2, seed=seed)`` (pilot_pattern.py:365); specification: oracle/ofdm.py."""
    bits = torch.empty(2 * n, dtype=torch.float32, device=_ffi.device())
    _ffi.check(_ffi.lib().samd_binary_source_f32(int(seed), int(call), 2 * n, _ffi.ptr(bits), _ffi.stream()),
               "pilot bits")
    b = bits.cpu().numpy().astype(np.int64)
    return qam(2, precision="single")[b[0::2] * 2 + b[1::2]]


class KroneckerPilotPattern(PilotPattern):
    """Orthogonal comb pilots: stream q of the num_tx*num_streams_per_tx sequences owns the
    subcarriers q, q+num_seq, ... of every pilot-carrying OFDM symbol (pilot_pattern.py:269-378)."""

    def __init__(self, resource_grid, pilot_ofdm_symbol_indices, normalize=True, seed=0, precision=None):
        num_tx, ns = resource_grid.num_tx, resource_grid.num_streams_per_tx
        n_sym, n_sc = resource_grid.num_ofdm_symbols, resource_grid.num_effective_subcarriers
        n_ps = len(pilot_ofdm_symbol_indices)
        num_seq = num_tx * ns
        assert (n_sc / num_seq) % 1 == 0, \
            "`num_effective_subcarriers` must be an integer multiple of `num_tx`*`num_streams_per_tx`."
        per_sym = n_sc // num_seq
        mask = np.zeros([num_tx, ns, n_sym, n_sc], bool)
        mask[..., list(pilot_ofdm_symbol_indices), :] = True
        pilots = np.zeros([num_tx, ns, n_ps, n_sc], np.complex64)
        for q in range(num_seq):
            i, j = divmod(q, ns)
            pilots[i, j, :, q::num_seq] = _qpsk_sequence(seed, q, n_ps * per_sym).reshape(n_ps, per_sym)
        super().__init__(mask, pilots.reshape(num_tx, ns, -1), normalize=normalize, precision=precision)
