"""OFDM resource grid and the (de)mapping blocks - mirror of reference
src/sionna/phy/ofdm/resource_grid.py:15-552.  The grid object is host-side bookkeeping
(NumPy); the blocks run the HIP kernels ``samd_rg_map_c64`` / ``samd_gather3``."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, Object
from .pilot_pattern import PilotPattern, EmptyPilotPattern, KroneckerPilotPattern


class ResourceGrid(Object):
    def __init__(self, num_ofdm_symbols, fft_size, subcarrier_spacing, num_tx=1, num_streams_per_tx=1,
                 cyclic_prefix_length=0, num_guard_carriers=(0, 0), dc_null=False, pilot_pattern=None,
                 pilot_ofdm_symbol_indices=None, precision=None):
        super().__init__(precision=precision)
        self._num_ofdm_symbols = num_ofdm_symbols
        self._fft_size = fft_size
        self._subcarrier_spacing = subcarrier_spacing
        self._cyclic_prefix_length = int(cyclic_prefix_length)
        self._num_tx = num_tx
        self._num_streams_per_tx = num_streams_per_tx
        self._num_guard_carriers = np.array(num_guard_carriers)
        self._dc_null = dc_null
        self._pilot_ofdm_symbol_indices = pilot_ofdm_symbol_indices
        self._check_settings()
        self.pilot_pattern = pilot_pattern

    cyclic_prefix_length = property(lambda self: self._cyclic_prefix_length)
    num_tx = property(lambda self: self._num_tx)
    num_streams_per_tx = property(lambda self: self._num_streams_per_tx)
    num_ofdm_symbols = property(lambda self: self._num_ofdm_symbols)
    num_guard_carriers = property(lambda self: self._num_guard_carriers)
    fft_size = property(lambda self: self._fft_size)
    subcarrier_spacing = property(lambda self: self._subcarrier_spacing)
    dc_null = property(lambda self: self._dc_null)

    @property
    def num_resource_elements(self):
        return self._fft_size * self._num_ofdm_symbols

    @property
    def num_effective_subcarriers(self):
        return int(self._fft_size - self._dc_null - np.sum(self._num_guard_carriers))

    @property
    def dc_ind(self):
        """(fft_size-1)/2 for odd, fft_size/2 for even sizes (resource_grid.py:171-179)."""
        return int(self._fft_size / 2 - (self._fft_size % 2 == 1) / 2)

    @property
    def effective_subcarrier_ind(self):
        sc = np.arange(self._num_guard_carriers[0], self._fft_size - self._num_guard_carriers[1])
        return sc[sc != self.dc_ind] if self._dc_null else sc

    @property
    def num_pilot_symbols(self):
        return self.pilot_pattern.num_pilot_symbols

    @property
    def num_data_symbols(self):
        return int(self.num_effective_subcarriers * self._num_ofdm_symbols - self.num_pilot_symbols)

    @property
    def num_zero_symbols(self):
        return int((self._fft_size - self.num_effective_subcarriers) * self._num_ofdm_symbols)

    @property
    def ofdm_symbol_duration(self):
        return (1. + self._cyclic_prefix_length / self._fft_size) / self._subcarrier_spacing

    @property
    def bandwidth(self):
        return self._fft_size * self._subcarrier_spacing

    @property
    def num_time_samples(self):
        return (self._fft_size + self._cyclic_prefix_length) * self._num_ofdm_symbols

    @property
    def pilot_pattern(self):
        return self._pilot_pattern

    @pilot_pattern.setter
    def pilot_pattern(self, value):
        if value is None or (isinstance(value, str) and value == "empty"):
            value = EmptyPilotPattern(self._num_tx, self._num_streams_per_tx, self._num_ofdm_symbols,
                                      self.num_effective_subcarriers, precision=self.precision)
        elif isinstance(value, PilotPattern):
            pass
        elif isinstance(value, str):
            assert value in ["kronecker", "empty"], "Unknown pilot pattern"
            assert self._pilot_ofdm_symbol_indices is not None, "You must provide pilot_ofdm_symbol_indices."
            value = KroneckerPilotPattern(self, self._pilot_ofdm_symbol_indices, precision=self.precision)
        else:
            raise ValueError("Unsupported pilot_pattern")
        self._pilot_pattern = value

    def _check_settings(self):
        assert self._num_ofdm_symbols > 0, "`num_ofdm_symbols` must be positive`."
        assert self._fft_size > 0, "`fft_size` must be positive`."
        assert self._cyclic_prefix_length >= 0, "`cyclic_prefix_length must be nonnegative."
        assert self._cyclic_prefix_length <= self._fft_size, "`cyclic_prefix_length cannot be longer than `fft_size`."
        assert self._num_tx > 0, "`num_tx` must be positive`."
        assert self._num_streams_per_tx > 0, "`num_streams_per_tx` must be positive`."
        assert len(self._num_guard_carriers) == 2, "`num_guard_carriers` must have two elements."
        assert np.all(self._num_guard_carriers >= 0), "`num_guard_carriers` must have nonnegative entries."
        assert np.sum(self._num_guard_carriers) <= self._fft_size - self._dc_null, \
            "Total number of guardcarriers cannot be larger than `fft_size`."
        return True

    def build_type_grid(self):
        """[num_tx, num_streams_per_tx, num_ofdm_symbols, fft_size] int32: 0 data, 1 pilot, 2 guard
        carrier, 3 DC carrier (resource_grid.py:283-311)."""
        t = np.full((self._num_tx, self._num_streams_per_tx, self._num_ofdm_symbols, self._fft_size), 2, np.int32)
        if self._dc_null:
            t[..., self.dc_ind] = 3
        t[..., self.effective_subcarrier_ind] = self.pilot_pattern.mask.astype(np.int32)
        return t

    # ---- index tables shared by the blocks (host int32 arrays)
    def _positions(self):
        """(data_pos, pilot_pos) on the FULL grid [S, T*fft]: running index of the data / pilot symbol
        carried by each RE in row-major (t, f) order, -1 elsewhere."""
        t = self.build_type_grid().reshape(self._num_tx * self._num_streams_per_tx, -1)
        data_pos = np.where(t == 0, np.cumsum(t == 0, axis=1) - 1, -1).astype(np.int32)
        pilot_pos = np.where(t == 1, np.cumsum(t == 1, axis=1) - 1, -1).astype(np.int32)
        return data_pos, pilot_pos

    def _data_ind_eff(self):
        """[S, num_data] indices (effective grid, row-major) of the data REs = first num_data
        entries of argsort(mask) (resource_grid.py:455-459)."""
        m = self.pilot_pattern.mask.reshape(self._num_tx * self._num_streams_per_tx, -1)
        return np.argsort(m, axis=-1, kind="stable")[:, :self.pilot_pattern.num_data_symbols].astype(np.int32)


def _dev_i32(a):
    return _ffi.to_device(np.ascontiguousarray(a, dtype=np.int32), torch.int32)


class ResourceGridMapper(Block):
    """[batch, num_tx, num_streams_per_tx, num_data_symbols] -> [batch, num_tx, num_streams_per_tx,
    num_ofdm_symbols, fft_size] with pilots inserted (resource_grid.py:350-412)."""

    def __init__(self, resource_grid, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._resource_grid = resource_grid
        self._tables = None

    def call(self, inputs):
        rg = self._resource_grid
        if self._tables is None:
            dp, pp = rg._positions()
            pil = np.asarray(rg.pilot_pattern.pilots).reshape(dp.shape[0], -1)
            self._tables = (_dev_i32(dp), _dev_i32(pp), _ffi.to_device(pil, self.cdtype))
        dp, pp, pil = self._tables
        x = _ffi.to_device(inputs, self.cdtype)
        b = x.shape[0]
        s, nd, npil = dp.shape[0], rg.num_data_symbols, pil.shape[1]
        assert tuple(x.shape[1:]) == (rg.num_tx, rg.num_streams_per_tx, nd), "unexpected input shape"
        out = torch.empty((b, rg.num_tx, rg.num_streams_per_tx, rg.num_ofdm_symbols, rg.fft_size),
                          dtype=self.cdtype, device=x.device)
        fn = _ffi.lib().samd_rg_map_c128 if self.precision == "double" else _ffi.lib().samd_rg_map_c64
        _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(pil) if npil else None, _ffi.ptr(dp), _ffi.ptr(pp), b, s, dp.shape[1], nd, npil,
                      _ffi.ptr(out), _ffi.stream()), "ResourceGridMapper")
        return out


class RemoveNulledSubcarriers(Block):
    """Gather of the effective subcarriers along the last axis (resource_grid.py:522-552)."""

    def __init__(self, resource_grid, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._sc_ind = np.asarray(resource_grid.effective_subcarrier_ind, np.int32)
        self._fft_size = resource_grid.fft_size
        self._dev = None

    def call(self, inputs):
        x = inputs
        cplx = x.dtype.is_complex
        x = _ffi.to_device(x, self.cdtype if cplx else self.rdtype)
        el = (2 if cplx else 1) * (2 if self.precision == "double" else 1)       # floats per element
        assert x.shape[-1] == self._fft_size
        if self._dev is None:
            self._dev = (_dev_i32(self._sc_ind.reshape(1, -1)), _dev_i32(np.zeros(1)))
        idx, grp = self._dev
        rows = x.numel() // self._fft_size
        out = torch.empty(tuple(x.shape[:-1]) + (len(self._sc_ind),), dtype=x.dtype, device=x.device)
        _ffi.check(_ffi.lib().samd_gather3(_ffi.ptr(x), _ffi.ptr(grp), _ffi.ptr(idx), rows, 1, self._fft_size, 1,
                                           len(self._sc_ind), el, _ffi.ptr(out), _ffi.stream()),
                   "RemoveNulledSubcarriers")
        return out


class ResourceGridDemapper(Block):
    """[batch, num_rx, num_streams_per_rx, num_ofdm_symbols, fft_size(, data_dim)] -> data symbols
    [batch, num_tx, num_streams_per_tx, num_data_symbols(, data_dim)] (resource_grid.py:414-520)."""

    def __init__(self, resource_grid, stream_management, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._rg, self._sm = resource_grid, stream_management
        self._dev = None

    def call(self, y):  # pylint: disable=arguments-renamed
        rg, sm = self._rg, self._sm
        cplx = y.dtype.is_complex
        y = _ffi.to_device(y, self.cdtype if cplx else self.rdtype)
        el = (2 if cplx else 1) * (2 if self.precision == "double" else 1)       # floats per element
        extra = y.dim() == 6
        if extra:   # [b, rx, s, T, fft, D] -> treat D as part of the batch
            d = y.shape[-1]
            y = y.permute(0, 5, 1, 2, 3, 4).reshape((-1,) + tuple(y.shape[1:5])).contiguous()
        b = y.shape[0]
        g_in = y.shape[1] * y.shape[2]
        n_in = rg.num_ofdm_symbols * rg.fft_size
        if self._dev is None:
            sc = np.asarray(rg.effective_subcarrier_ind)
            di = rg._data_ind_eff()                                          # effective-grid indices
            t, f = np.divmod(di, rg.num_effective_subcarriers)
            self._dev = (_dev_i32(sm.stream_ind), _dev_i32(t * rg.fft_size + sc[f]))
        grp, idx = self._dev
        g_out, n_out = idx.shape
        out = torch.empty((b, sm.num_tx, sm.num_streams_per_tx, n_out), dtype=y.dtype, device=y.device)
        _ffi.check(_ffi.lib().samd_gather3(_ffi.ptr(y), _ffi.ptr(grp), _ffi.ptr(idx), b, g_in, n_in, g_out, n_out,
                                           el, _ffi.ptr(out), _ffi.stream()), "ResourceGridDemapper")
        if extra:
            out = out.reshape((-1, d) + tuple(out.shape[1:])).permute(0, 2, 3, 4, 1).contiguous()
        return out
