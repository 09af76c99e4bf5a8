"""LS channel estimation with nearest-neighbour or linear interpolation - mirror of reference
src/sionna/phy/ofdm/channel_estimation.py (``BaseChannelEstimator.call`` :138-173,
``LSChannelEstimator`` :175-285, ``NearestNeighborInterpolator`` :323-435, ``LinearInterpolator``
:437-733).

The reference gathers the pilot REs, divides by the pilots and then gathers again to spread
the estimates over the grid (plus ~6 transposes).  Estimation and spreading commute with the
per-pilot division, so ONE kernel does both: h_hat[.., t, f] = y[.., nearest pilot RE] * (1 /
pilot).  Linear interpolation gathers its two supports per axis through ``samd_lin_interp_c64``;
the LMMSE interpolator is outside the hot path."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, Object, wrap, Pending, defer


class NearestNeighborInterpolator(Object):
    """For every RE the index of the closest (Manhattan distance over (symbol, subcarrier),
    first index on ties) pilot with non-zero energy (channel_estimation.py:364-411)."""

    def __init__(self, pilot_pattern):
        super().__init__()
        assert pilot_pattern.num_pilot_symbols > 0, "The pilot pattern cannot be empty"
        mask = np.asarray(pilot_pattern.mask)
        pilots = np.asarray(pilot_pattern.pilots)
        s = mask.shape[0] * mask.shape[1]
        m = mask.reshape(s, mask.shape[2], mask.shape[3])
        pil = pilots.reshape(s, -1)
        assert np.max(np.sum(np.abs(pil) == 0, -1)) < pil.shape[-1], \
            "Each pilot sequence must have at least one nonzero entry"
        ii, jj = np.meshgrid(np.arange(mask.shape[2]), np.arange(mask.shape[3]), indexing="ij")
        g = np.zeros(m.shape, np.int32)
        for a in range(s):
            ip, jp = np.nonzero(m[a])                                          # row-major pilot order
            d = np.abs(ii[..., None] - ip) + np.abs(jj[..., None] - jp)        # [T, F, num_pilots]
            d[..., np.abs(pil[a]) == 0] = mask.shape[2] + mask.shape[3]        # never pick empty pilots
            g[a] = np.argmin(d, axis=-1)
        self._gather_ind = g.reshape(mask.shape)

    gather_ind = property(lambda self: self._gather_ind)

    def __call__(self, h_hat, err_var):
        """Spread estimates at the pilot positions [..., tx, s, num_pilots] over the grid
        [..., tx, s, T, F] (generic torch indexing; the fused estimator does not go through here)."""
        g = torch.from_numpy(self._gather_ind.reshape(self._gather_ind.shape[:2] + (-1,))).to(h_hat.device).long()
        def spread(x):
            x = torch.broadcast_to(x, tuple(x.shape[:-3]) + tuple(g.shape[:2]) + (x.shape[-1],))
            idx = g.expand(tuple(x.shape[:-1]) + (g.shape[-1],))
            return torch.gather(x, -1, idx).reshape(tuple(x.shape[:-1]) + tuple(self._gather_ind.shape[2:]))
        return wrap(spread(h_hat)), wrap(spread(err_var))


class LinearInterpolator(Object):
    """``LinearInterpolator(pilot_pattern, time_avg=False)(h_hat, err_var)``: estimates at the pilots
    [..., num_tx, num_streams, num_pilots] -> [..., num_tx, num_streams, num_ofdm_symbols,
    num_effective_subcarriers], first across subcarriers then across OFDM symbols
    (channel_estimation.py:437-733)."""

    def __init__(self, pilot_pattern, time_avg=False):
        super().__init__()
        assert pilot_pattern.num_pilot_symbols > 0, "The pilot pattern cannot be empty"
        self._time_avg = bool(time_avg)
        mask = np.asarray(pilot_pattern.mask)
        self._mask_shape = tuple(mask.shape)
        s = mask.shape[0] * mask.shape[1]
        t_, f_ = mask.shape[2:]
        m = mask.reshape(s, t_, f_)
        pil = np.asarray(pilot_pattern.pilots).reshape(s, -1)
        assert np.max(np.sum(np.abs(pil) == 0, -1)) < pil.shape[-1], \
            "Each pilot sequence must have at least one nonzero entry"
        fi0 = np.zeros((s, t_, f_), np.int32); fi1 = np.zeros_like(fi0)        # 0 = zero pad
        fx0 = np.full((s, t_, f_), -1, np.float32); fx1 = fx0.copy()
        t0 = np.zeros((s, t_), np.int32); t1 = np.zeros_like(t0)
        npil = np.ones(s, np.float32)
        sub = np.arange(f_)
        for a in range(s):
            ti, fj = np.nonzero(m[a])                            # row-major = pilot order
            nz = np.abs(pil[a]) > 0
            syms = []
            for t in range(t_):
                sel = np.flatnonzero((ti == t) & nz)             # pilot numbers with energy in this symbol
                if len(sel) == 0:
                    continue
                syms.append(t)
                cols = fj[sel]
                if len(sel) == 1:
                    k0 = k1 = np.zeros(f_, np.int64)
                else:                                            # right support: first pilot >= f, at least the 2nd
                    k1 = np.clip(np.searchsorted(cols, sub, side="left"), 1, len(sel) - 1)
                    k0 = k1 - 1
                fi0[a, t], fi1[a, t] = sel[k0] + 1, sel[k1] + 1
                fx0[a, t], fx1[a, t] = cols[k0], cols[k1]
            syms = np.asarray(syms)
            npil[a] = len(syms)
            if len(syms) == 1:
                t0[a] = t1[a] = syms[0]
            elif len(syms) >= 2:
                k1 = np.clip(np.searchsorted(syms, np.arange(t_), side="left"), 1, len(syms) - 1)
                t0[a], t1[a] = syms[k1 - 1], syms[k1]
        self._tables = (fi0, fi1, fx0, fx1, t0, t1, npil)
        self._dev = None
        self._num_pilots = pil.shape[-1]

    def _interpolate(self, x):
        x = _ffi.to_device(x, torch.complex64)
        s, t_, f_ = self._tables[0].shape
        assert x.shape[-1] == self._num_pilots and x.shape[-3] * x.shape[-2] == s, \
            "inputs must have shape [..., num_tx, num_streams_per_tx, num_pilots]"
        if self._dev is None:
            fi0, fi1, fx0, fx1, t0, t1, npil = self._tables
            self._dev = tuple(_ffi.to_device(a, torch.int32 if a.dtype == np.int32 else torch.float32)
                              for a in (fi0, fi1, fx0, fx1, t0, t1, npil))
        lead = tuple(x.shape[:-3])
        rows = int(np.prod(lead)) if lead else 1
        out = torch.empty(lead + self._mask_shape, dtype=torch.complex64, device=x.device)
        d = self._dev
        _ffi.check(_ffi.lib().samd_lin_interp_c64(_ffi.ptr(x), _ffi.ptr(d[0]), _ffi.ptr(d[1]), _ffi.ptr(d[2]), _ffi.ptr(d[3]),
                                                  _ffi.ptr(d[4]), _ffi.ptr(d[5]), _ffi.ptr(d[6]), rows, s, self._num_pilots,
                                                  t_, f_, int(self._time_avg), _ffi.ptr(out), _ffi.stream()),
                   "LinearInterpolator")
        return out

    def __call__(self, h_hat, err_var):
        h = self._interpolate(h_hat)
        ev = _ffi.to_device(err_var, torch.float32)
        ev = self._interpolate(torch.complex(ev, torch.zeros_like(ev))).real      # :726-730
        return wrap(h), wrap(ev.contiguous())


class LSChannelEstimator(Block):
    """``LSChannelEstimator(resource_grid, interpolation_type="nn")(y, no) -> (h_hat, err_var)``;
    y [batch, num_rx, num_rx_ant, num_ofdm_symbols, fft_size]; h_hat [batch, num_rx, num_rx_ant, num_tx,
    num_streams_per_tx, num_ofdm_symbols, num_effective_subcarriers]; err_var broadcastable to it."""

    def __init__(self, resource_grid, interpolation_type="nn", interpolator=None, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert interpolation_type in ["nn", "lin", "lin_time_avg", None], "Unsupported `interpolation_type`"
        self._rg = resource_grid
        self._lin = None
        if interpolator is not None and not isinstance(interpolator, (NearestNeighborInterpolator, LinearInterpolator)):
            # any object with the interface of BaseChannelInterpolator (channel_estimation.py:287-321): called with the LS
            # estimates and error variances at the pilots (device tensors), returns them for the whole grid (:160-167)
            assert callable(interpolator), "interpolator must be callable: (h_hat, err_var) at the pilots -> whole grid"
            self._lin, interpolation_type = interpolator, None
        elif isinstance(interpolator, LinearInterpolator):
            self._lin, interpolation_type = interpolator, None
        elif isinstance(interpolator, NearestNeighborInterpolator):
            interpolation_type = "nn"
        elif interpolation_type in ("lin", "lin_time_avg"):
            self._lin = LinearInterpolator(resource_grid.pilot_pattern, time_avg=interpolation_type == "lin_time_avg")
            interpolation_type = None
        self._interpolation_type = interpolation_type
        pp = resource_grid.pilot_pattern
        s = pp.mask.shape[0] * pp.mask.shape[1]
        m = pp.mask.reshape(s, -1)
        # pilot REs in row-major order = argsort(mask, DESCENDING)[:num_pilots] (channel_estimation.py:108-112)
        pilot_re = np.stack([np.flatnonzero(m[a]) for a in range(s)])         # [S, num_pilots] effective grid
        pil = np.asarray(pp.pilots).reshape(s, -1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.where(pil != 0, 1 / np.where(pil != 0, pil, 1), 0).astype(np.complex64)      # divide_no_nan
            ev = np.where(pil != 0, 1 / np.where(pil != 0, np.abs(pil) ** 2, 1), 0).astype(np.float32)
        if interpolation_type == "nn":
            g = NearestNeighborInterpolator(pp).gather_ind.reshape(s, -1)     # [S, T*F] pilot number
            pilot_re = np.take_along_axis(pilot_re, g, axis=1)
            inv = np.take_along_axis(inv, g, axis=1)
            ev = np.take_along_axis(ev, g, axis=1)
            self._out_shape = tuple(pp.mask.shape)
        else:
            self._out_shape = tuple(pp.mask.shape[:2]) + (pp.num_pilot_symbols,)
        sc = np.asarray(resource_grid.effective_subcarrier_ind)
        t, f = np.divmod(pilot_re, resource_grid.num_effective_subcarriers)
        self._src = (t * resource_grid.fft_size + sc[f]).astype(np.int32)      # index into the full grid
        self._inv, self._ev = inv, ev
        self._dev = None
        self._defer = bool(kwargs.get("defer", True))          # defer=False: always materialise h_hat (tests compare the two)

    def estimate_at_pilot_locations(self, y_pilots, no):
        """LS estimates at the pilot-carrying resource elements (channel_estimation.py:257-285): y_pilots
        [batch, num_rx, num_rx_ant, num_tx, num_streams_per_tx, num_pilot_symbols] -> (h_ls = y_pilots / pilots with
        divide_no_nan, err_var = no / |pilots|^2 broadcastable to it).  The scaling kernel of ``call`` with an identity gather."""
        self._require_single()
        pp = self._rg.pilot_pattern
        yp = _ffi.to_device(y_pilots, torch.complex64).contiguous()
        s, npil = pp.mask.shape[0] * pp.mask.shape[1], pp.num_pilot_symbols
        assert yp.dim() == 6 and tuple(yp.shape[3:]) == (pp.mask.shape[0], pp.mask.shape[1], npil), \
            "y_pilots must have shape [batch, num_rx, num_rx_ant, num_tx, num_streams_per_tx, num_pilot_symbols]"
        pil = np.asarray(pp.pilots).reshape(s, -1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.where(pil != 0, 1 / np.where(pil != 0, pil, 1), 0).astype(np.complex64)
            ev = np.where(pil != 0, 1 / np.where(pil != 0, np.abs(pil) ** 2, 1), 0).astype(np.float32)
        src = np.arange(s * npil, dtype=np.int32).reshape(s, npil)
        rows = yp.shape[0] * yp.shape[1] * yp.shape[2]
        h_ls = torch.empty_like(yp)
        src_d, inv_d = _ffi.to_device(src, torch.int32), _ffi.to_device(inv, torch.complex64)   # (alive until the launch is queued)
        if rows and npil:
            _ffi.check(_ffi.lib().samd_ls_gather_scale_c64(_ffi.ptr(yp), _ffi.ptr(src_d), _ffi.ptr(inv_d), rows, s, npil, s * npil,
                                                           _ffi.ptr(h_ls), _ffi.stream()), "LSChannelEstimator.estimate_at_pilot_locations")
        no = _ffi.to_device(no, torch.float32)
        no = no.reshape(tuple(no.shape) + (1,) * (6 - no.dim()))
        err_var = no * _ffi.to_device(ev, torch.float32).reshape(tuple(pp.mask.shape[:2]) + (npil,))
        from ..block import wrap
        return wrap(h_ls), wrap(err_var)

    def call(self, y, no):
        self._require_single()
        rg = self._rg
        y = _ffi.to_device(y, torch.complex64)
        assert y.dim() == 5 and y.shape[-2:] == (rg.num_ofdm_symbols, rg.fft_size), \
            "y must have shape [batch, num_rx, num_rx_ant, num_ofdm_symbols, fft_size]"
        if self._dev is None:
            self._dev = (_ffi.to_device(self._src, torch.int32), _ffi.to_device(self._inv, torch.complex64),
                         _ffi.to_device(self._ev, torch.float32))
        src, inv, ev = self._dev
        s, n_out = src.shape
        rows = y.shape[0] * y.shape[1] * y.shape[2]
        h_hat = torch.empty(tuple(y.shape[:3]) + self._out_shape, dtype=torch.complex64, device=y.device)

        def fill_h(out):
            _ffi.check(_ffi.lib().samd_ls_gather_scale_c64(_ffi.ptr(y), _ffi.ptr(src), _ffi.ptr(inv), rows, s, n_out,
                                                           rg.num_ofdm_symbols * rg.fft_size, _ffi.ptr(out),
                                                           _ffi.stream()), "LSChannelEstimator")
        # err_var = no / |pilot|^2, broadcastable to h_hat (channel_estimation.py:276-283); `no` has the
        # first n <= 3 dims of [batch, num_rx, num_rx_ant] - a handful of elements, plain broadcasting
        no = _ffi.to_device(no, torch.float32)
        no3 = no.reshape(tuple(no.shape) + (1,) * (3 - no.dim()))
        no = no3.reshape(tuple(no3.shape) + (1,) * len(self._out_shape))
        if self._interpolation_type == "nn" and self._defer:
            # nearest-neighbour interpolation: h_hat is a gather of the LS estimates at the pilots.  It is returned DEFERRED
            # (block.py): the fused LS + LMMSE (+ demapper) kernel of LMMSEEqualizer / LinearDetector works from the
            # recipe and never materialises it; any other use fills it with the gather kernel first.
            if no3.numel() == 1:
                err_var = torch.clamp_min(no * ev.reshape(self._out_shape), 0.)           # [.., S.., T, F] table: a few KB
            else:
                shape = tuple(torch.broadcast_shapes(tuple(no.shape), self._out_shape))
                err_var = defer(torch.empty(shape, dtype=torch.float32, device=y.device),
                                Pending("ls_err_var", lambda out: torch.clamp_min(no * ev.reshape(self._out_shape), 0., out=out)))
            rec = Pending("ls_nn", fill_h, y=y, src=src, coef=inv, ev=ev, no=no3.reshape(-1) if no3.numel() == 1 else
                          torch.broadcast_to(no3, tuple(y.shape[:3])).contiguous().reshape(-1), rg=rg, err_var=err_var)
            return defer(h_hat, rec), err_var
        fill_h(h_hat)
        err_var = no * ev.reshape(self._out_shape)
        if self._lin is not None:
            # a foreign interpolator sees err_var broadcast to h_hat's shape like in the reference (channel_estimation.py:160-163);
            # the built-in one keeps the leading [batch, num_rx, num_rx_ant] dims unexpanded (its kernel is linear in them)
            lead = tuple(err_var.shape[:3]) if isinstance(self._lin, LinearInterpolator) else tuple(h_hat.shape[:3])
            err_var = torch.broadcast_to(err_var, lead + self._out_shape)
            h_hat, err_var = self._lin(h_hat, err_var.contiguous())
            h_hat, err_var = _ffi.to_device(h_hat, torch.complex64), _ffi.to_device(err_var, torch.float32)
        return h_hat, torch.clamp_min(err_var, 0.)
