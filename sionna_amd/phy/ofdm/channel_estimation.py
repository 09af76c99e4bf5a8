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
        dbl = str(getattr(x, "dtype", "")).endswith("complex128")          # the precision follows the estimates handed in
        cdt = torch.complex128 if dbl else torch.complex64
        x = _ffi.to_device(x, cdt)
        s, t_, f_ = self._tables[0].shape
        assert x.shape[-1] == self._num_pilots and x.shape[-3] * x.shape[-2] == s, \
            "inputs must have shape [..., num_tx, num_streams_per_tx, num_pilots]"
        if self._dev is None:
            fi0, fi1, fx0, fx1, t0, t1, npil = self._tables
            self._dev = tuple(_ffi.to_device(a, torch.int32 if a.dtype == np.int32 else torch.float32)
                              for a in (fi0, fi1, fx0, fx1, t0, t1, npil))
        lead = tuple(x.shape[:-3])
        rows = int(np.prod(lead)) if lead else 1
        out = torch.empty(lead + self._mask_shape, dtype=cdt, device=x.device)
        d = self._dev
        fn = _ffi.lib().samd_lin_interp_c128 if dbl else _ffi.lib().samd_lin_interp_c64
        _ffi.check(fn(_ffi.ptr(x), _ffi.ptr(d[0]), _ffi.ptr(d[1]), _ffi.ptr(d[2]), _ffi.ptr(d[3]), _ffi.ptr(d[4]), _ffi.ptr(d[5]),
                      _ffi.ptr(d[6]), rows, s, self._num_pilots, t_, f_, int(self._time_avg), _ffi.ptr(out), _ffi.stream()),
                   "LinearInterpolator")
        return out

    def __call__(self, h_hat, err_var):
        h = self._interpolate(h_hat)
        ev = _ffi.to_device(err_var, torch.float64 if str(getattr(err_var, "dtype", "")).endswith("float64") else torch.float32)
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
            inv = np.where(pil != 0, 1 / np.where(pil != 0, pil, 1), 0).astype(self._np_cdtype)      # divide_no_nan
            ev = np.where(pil != 0, 1 / np.where(pil != 0, np.abs(pil) ** 2, 1), 0).astype(self._np_rdtype)
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
        pp = self._rg.pilot_pattern
        yp = _ffi.to_device(y_pilots, self.cdtype).contiguous()
        s, npil = pp.mask.shape[0] * pp.mask.shape[1], pp.num_pilot_symbols
        assert yp.dim() == 6 and tuple(yp.shape[3:]) == (pp.mask.shape[0], pp.mask.shape[1], npil), \
            "y_pilots must have shape [batch, num_rx, num_rx_ant, num_tx, num_streams_per_tx, num_pilot_symbols]"
        pil = np.asarray(pp.pilots).reshape(s, -1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.where(pil != 0, 1 / np.where(pil != 0, pil, 1), 0).astype(self._np_cdtype)
            ev = np.where(pil != 0, 1 / np.where(pil != 0, np.abs(pil) ** 2, 1), 0).astype(self._np_rdtype)
        src = np.arange(s * npil, dtype=np.int32).reshape(s, npil)
        rows = yp.shape[0] * yp.shape[1] * yp.shape[2]
        h_ls = torch.empty_like(yp)
        src_d, inv_d = _ffi.to_device(src, torch.int32), _ffi.to_device(inv, self.cdtype)   # (alive until the launch is queued)
        if rows and npil:
            _ffi.check(self._ls_fn()(_ffi.ptr(yp), _ffi.ptr(src_d), _ffi.ptr(inv_d), rows, s, npil, s * npil,
                                     _ffi.ptr(h_ls), _ffi.stream()), "LSChannelEstimator.estimate_at_pilot_locations")
        no = _ffi.to_device(no, self.rdtype)
        no = no.reshape(tuple(no.shape) + (1,) * (6 - no.dim()))
        err_var = no * _ffi.to_device(ev, self.rdtype).reshape(tuple(pp.mask.shape[:2]) + (npil,))
        from ..block import wrap
        return wrap(h_ls), wrap(err_var)

    def _ls_fn(self):
        return _ffi.lib().samd_ls_gather_scale_c128 if self.precision == "double" else _ffi.lib().samd_ls_gather_scale_c64

    def call(self, y, no):
        rg = self._rg
        dbl = self.precision == "double"      # float64: csrc/f64_ofdm.hip, h_hat always materialised (the fused kernels are float32)
        y = _ffi.to_device(y, self.cdtype)
        assert y.dim() == 5 and y.shape[-2:] == (rg.num_ofdm_symbols, rg.fft_size), \
            "y must have shape [batch, num_rx, num_rx_ant, num_ofdm_symbols, fft_size]"
        if self._dev is None:
            self._dev = (_ffi.to_device(self._src, torch.int32), _ffi.to_device(self._inv, self.cdtype),
                         _ffi.to_device(self._ev, self.rdtype))
        src, inv, ev = self._dev
        s, n_out = src.shape
        rows = y.shape[0] * y.shape[1] * y.shape[2]
        h_hat = torch.empty(tuple(y.shape[:3]) + self._out_shape, dtype=self.cdtype, device=y.device)

        def fill_h(out):
            _ffi.check(self._ls_fn()(_ffi.ptr(y), _ffi.ptr(src), _ffi.ptr(inv), rows, s, n_out,
                                     rg.num_ofdm_symbols * rg.fft_size, _ffi.ptr(out), _ffi.stream()), "LSChannelEstimator")
        # err_var = no / |pilot|^2, broadcastable to h_hat (channel_estimation.py:276-283); `no` has the
        # first n <= 3 dims of [batch, num_rx, num_rx_ant] - a handful of elements, plain broadcasting
        no = _ffi.to_device(no, self.rdtype)
        no3 = no.reshape(tuple(no.shape) + (1,) * (3 - no.dim()))
        no = no3.reshape(tuple(no3.shape) + (1,) * len(self._out_shape))
        if self._interpolation_type == "nn" and self._defer and not dbl:
            # nearest-neighbour interpolation: h_hat is a gather of the LS estimates at the pilots.  It is returned DEFERRED
            # (block.py): the fused LS + LMMSE (+ demapper) kernel of LMMSEEqualizer / LinearDetector works from the
            # recipe and never materialises it; any other use fills it with the gather kernel first.
            if no3.numel() == 1:
                err_var = torch.clamp_min(no * ev.reshape(self._out_shape), 0.)           # [.., S.., T, F] table: a few KB
            else:
                shape = tuple(torch.broadcast_shapes(tuple(no.shape), self._out_shape))
                err_var = defer(torch.empty(shape, dtype=torch.float32, device=y.device),
                                Pending("ls_err_var", lambda out: torch.clamp_min(no * ev.reshape(self._out_shape), 0., out=out), guard=(no,)))
            rec = Pending("ls_nn", fill_h, guard=(y, no3), y=y, src=src, coef=inv, ev=ev, no=no3.reshape(-1) if no3.numel() == 1 else
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
            h_hat, err_var = _ffi.to_device(h_hat, self.cdtype), _ffi.to_device(err_var, self.rdtype)
        return h_hat, torch.clamp_min(err_var, 0.)


# ---------------------------------------------------------------------------------------------------------------------
# LMMSE interpolation (reference channel_estimation.py:736-1853)
#
# Unlike the blocks above this is not a streaming pass over the grid but dense small-matrix algebra whose matrices depend
# on the error variances handed in: per (example, receive antenna) and per group of grid rows with the same pilot
# positions one Hermitian system "pilot covariance + diag(error variance)" is solved against the covariance columns, and
# the resulting interpolation matrix is applied to the pilot estimates.  These are library-shaped operations (batched
# LU solves and GEMMs: rocSOLVER / rocBLAS behind torch.linalg.solve / matmul on the device the estimates live on), so
# there is no hand-written kernel here.  The covariance systems of the 38.901 models are ill-conditioned (1e4 and worse),
# which single precision resolves poorly: the algebra runs in complex128 / float64 on the device and the results are cast
# to the block's precision - the reference's float32 pseudo-inverse agrees with it to ~1e-4 of scale
# (tests/test_lmmse_interpolator.py).
def _c128(x, device=None):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    return t.to(device=device if device is not None else t.device, dtype=torch.complex128)


def _rescale(h, err, hv, h_var):
    """re-scaling after an intermediate step (channel_estimation.py:1129-1153, 1347-1363), complex arithmetic, divide_no_nan"""
    den = hv + h_var - err
    zero = den == 0
    s = torch.where(zero, torch.zeros_like(den), 2. * h_var / torch.where(zero, torch.ones_like(den), den))
    h = s * h
    err = torch.real(s * (s - 1.) * hv + (1. - s) * h_var + s * err)
    return h, torch.clamp_min(err, 0.)


class _LMMSE1D:
    """LMMSEInterpolator1D (channel_estimation.py:736-1155) on rows grouped by their pilot positions (the reference pads
    every row to the largest pilot count and takes the minimum-norm least-squares solution, which is the same numbers
    while pilot covariance + error variance is non-singular)."""

    def __init__(self, pilot_mask, cov, last_step):
        self._cov = np.asarray(cov, np.complex128)
        self._last = last_step
        ntx, ns, O, I = pilot_mask.shape
        groups = {}
        for tx in range(ntx):
            for st in range(ns):
                for o in range(O):
                    groups.setdefault(tuple(np.flatnonzero(pilot_mask[tx, st, o] == 1)), []).append((tx * ns + st) * O + o)
        self._groups = [(np.asarray(p, np.int64), np.asarray(r, np.int64)) for p, r in groups.items() if len(p)]
        self._dev = {}

    def _tables(self, device):
        if device not in self._dev:
            cov = torch.from_numpy(self._cov).to(device)
            self._dev[device] = (cov, [(torch.from_numpy(p).to(device), torch.from_numpy(r).to(device)) for p, r in self._groups])
        return self._dev[device]

    def __call__(self, h, err):
        """h complex128, err float64: [..., tx, st, O, I] -> the same shapes"""
        cov, groups = self._tables(h.device)
        shp = h.shape
        R, I = shp[-4] * shp[-3] * shp[-2], shp[-1]
        h2, e2 = h.reshape(shp[:-4] + (R, I)), err.reshape(shp[:-4] + (R, I))
        h_var = torch.diagonal(cov)
        out_h = torch.zeros_like(h2)
        out_e = torch.clamp_min(torch.real(h_var), 0.).expand(e2.shape).clone()
        out_hv = torch.zeros_like(h2)
        for p, rows in groups:
            hp = h2.index_select(-2, rows).index_select(-1, p)                 # [..., G, np]
            ep = e2.index_select(-2, rows).index_select(-1, p)
            cpp = cov.index_select(0, p).index_select(1, p)
            b = cov.index_select(0, p)                                          # [np, I]
            a = cpp + torch.diag_embed(ep).to(torch.complex128)
            x = torch.linalg.solve(a, b.expand(a.shape[:-2] + b.shape))
            ext = x.conj().transpose(-1, -2)                                    # [..., G, I, np]
            hn = torch.matmul(ext, hp.unsqueeze(-1)).squeeze(-1)
            en = torch.clamp_min(torch.real(h_var - torch.sum(ext * b.transpose(0, 1), dim=-1)), 0.)
            hv = torch.sum(ext * torch.matmul(ext.conj(), cpp.transpose(0, 1)), dim=-1) \
                + torch.sum(ext * ext.conj() * ep.unsqueeze(-2).to(torch.complex128), dim=-1)
            out_h.index_copy_(-2, rows, hn)
            out_e.index_copy_(-2, rows, en)
            out_hv.index_copy_(-2, rows, hv)
        if not self._last:
            out_h, out_e = _rescale(out_h, out_e.to(torch.complex128), out_hv, h_var)
        return out_h.reshape(shp), out_e.reshape(shp)


class _SpatialFilter:
    """SpatialChannelFilter (channel_estimation.py:1157-1365): LMMSE smoothing across the receive antennas of every
    resource element; the antenna dimension is the last one of the inputs."""

    def __init__(self, cov, last_step):
        self._cov, self._last, self._dev = np.asarray(cov, np.complex128), last_step, {}

    def __call__(self, h, err):
        if h.device not in self._dev:
            self._dev[h.device] = torch.from_numpy(self._cov).to(h.device)
        cov = self._dev[h.device]
        a = cov + torch.diag_embed(err).to(torch.complex128)
        w = torch.linalg.solve(a, cov.expand(a.shape)).conj().transpose(-1, -2)           # (A^-1 C)^H (:1301-1306)
        hn = torch.matmul(w, h.unsqueeze(-1)).squeeze(-1)
        h_var = torch.diagonal(cov)
        en = torch.clamp_min(torch.real(h_var - torch.sum(cov.transpose(0, 1) * w, dim=-1)), 0.)
        if not self._last:
            hv = torch.sum(w * torch.matmul(w.conj(), cov.transpose(0, 1)), dim=-1) \
                + torch.sum(w * w.conj() * err.unsqueeze(-2).to(torch.complex128), dim=-1)
            hn, en = _rescale(hn, en.to(torch.complex128), hv, h_var)
        return hn, en


class LMMSEInterpolator(Object):
    """``LMMSEInterpolator(pilot_pattern, cov_mat_time, cov_mat_freq, cov_mat_space=None, order="t-f")(h_hat, err_var)``:
    LMMSE interpolation of the channel estimates at the pilots [batch, num_rx, num_rx_ant, num_tx, num_streams_per_tx,
    num_pilots] over the resource grid, dimension by dimension in the given ``order`` ("t" time, "f" frequency, optional
    "s" smoothing across the receive antennas), with the re-scaling between steps; returns (h_hat, err_var)
    [batch, num_rx, num_rx_ant, num_tx, num_streams_per_tx, num_ofdm_symbols, num_effective_subcarriers]
    (reference channel_estimation.py:1367-1853).  Use as ``LSChannelEstimator(rg, interpolator=LMMSEInterpolator(...))``."""

    def __init__(self, pilot_pattern, cov_mat_time, cov_mat_freq, cov_mat_space=None, order="t-f"):
        super().__init__()
        order = order.split("-")
        assert 2 <= len(order) <= 3, "Invalid order for interpolation."
        seen = set()
        for o in order:
            assert o in ("s", "f", "t"), f"Uknown dimension {o}"
            assert o not in seen, {"s": "Spatial smoothing can be specified at most once",
                                   "t": "Temporal interpolation can be specified once only",
                                   "f": "Frequency interpolation can be specified once only"}[o]
            seen.add(o)
        if "s" in seen:
            assert cov_mat_space is not None, "A spatial covariance matrix is required for spatial smoothing"
        assert "f" in seen, "Frequency interpolation is required"
        assert "t" in seen, "Time interpolation is required"
        self._order = order
        host = lambda c: None if c is None else np.asarray(c.detach().cpu() if isinstance(c, torch.Tensor) else c, np.complex128)
        ct, cf, cs = host(cov_mat_time), host(cov_mat_freq), host(cov_mat_space)
        mask, pilots = np.asarray(pilot_pattern.mask), np.asarray(pilot_pattern.pilots)
        ntx, ns, T, F = mask.shape
        # 0 data / unused, 1 pilot with energy, 2 zero-power pilot (_build_pilot_mask, :1704-1734)
        pm = np.zeros([ntx, ns, T, F], int)
        self._src, self._dst = [], []              # scatter of the inputs onto the grid (_build_inputs2rg_indices, :1736-1761)
        for tx in range(ntx):
            for st in range(ns):
                pos = np.flatnonzero(mask[tx, st].reshape(-1))
                live = np.abs(pilots[tx, st, :len(pos)]) > 0.0
                pm[tx, st].reshape(-1)[pos] = np.where(live, 1, 2)
                self._src.append((tx * ns + st) * pilots.shape[-1] + np.flatnonzero(live))
                self._dst.append((tx * ns + st) * T * F + pos[live])
        self._src, self._dst = np.concatenate(self._src), np.concatenate(self._dst)
        self._grid = (ntx, ns, T, F)
        self._num_pilots = pilots.shape[-1]
        self._steps = []
        for i, o in enumerate(order):
            last = i == len(order) - 1
            if o == "f":
                step = _LMMSE1D(pm, cf, last)
                pm = np.where(np.any(pm == 1, axis=-1, keepdims=True), 1, pm)
            elif o == "t":
                pmt = np.swapaxes(pm, -1, -2)
                step = _LMMSE1D(pmt, ct, last)
                pm = np.swapaxes(np.where(np.any(pmt == 1, axis=-1, keepdims=True), 1, pmt), -1, -2)
            else:
                step = _SpatialFilter(cs, last)
            self._steps.append((o, step, (pm == 1).astype(np.float64)))
        self._dev = {}

    def _interpolate(self, h_hat, err_var):
        """the algebra on whatever device the inputs live on (complex128 / float64 inside); h_hat
        [B, rx, ra, tx, st, num_pilots], err_var broadcastable to it"""
        dev = h_hat.device
        ntx, ns, T, F = self._grid
        if dev not in self._dev:
            self._dev[dev] = (torch.from_numpy(self._src).to(dev), torch.from_numpy(self._dst).to(dev),
                              [torch.from_numpy(m).to(dev) for _, _, m in self._steps])
        src, dst, masks = self._dev[dev]
        lead = tuple(h_hat.shape[:3])
        assert tuple(h_hat.shape[3:]) == (ntx, ns, self._num_pilots), \
            "h_hat must have shape [batch, num_rx, num_rx_ant, num_tx, num_streams_per_tx, num_pilots]"
        hp = h_hat.to(torch.complex128).reshape(lead + (-1,))
        ep = torch.broadcast_to(err_var, h_hat.shape).to(torch.float64).reshape(lead + (-1,))
        h = torch.zeros(lead + (ntx * ns * T * F,), dtype=torch.complex128, device=dev).index_copy_(-1, dst, hp.index_select(-1, src))
        e = torch.zeros(lead + (ntx * ns * T * F,), dtype=torch.float64, device=dev).index_copy_(-1, dst, ep.index_select(-1, src))
        h, e = h.reshape(lead + (ntx, ns, T, F)), e.reshape(lead + (ntx, ns, T, F))
        for (o, step, _), m in zip(self._steps, masks):
            if o == "f":
                h, e = step(h, e)
            elif o == "t":
                h, e = step(h.transpose(-1, -2).contiguous(), e.transpose(-1, -2).contiguous())
                h, e = h.transpose(-1, -2), e.transpose(-1, -2)
            else:
                h, e = step(h.movedim(2, -1).contiguous(), e.movedim(2, -1).contiguous())
                h, e = h.movedim(-1, 2), e.movedim(-1, 2)
            e = e * m
        return h.contiguous(), e.contiguous()

    def __call__(self, h_hat, err_var):
        h_hat = _ffi.to_device(h_hat, self.cdtype)              # config.precision (the block takes no precision argument)
        err_var = _ffi.to_device(err_var, self.rdtype)
        h, e = self._interpolate(h_hat, err_var)
        return wrap(h.to(self.cdtype)), wrap(e.to(self.rdtype))


def _tdl_pdp(model):
    import json
    import os
    assert model in ("A", "B", "C", "D", "E"), "Invalid TDL model"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "channel", "tr38901", "tdl_models.json")
    with open(path) as f:
        p = json.load(f)[model]
    return bool(p["los"]), np.asarray(p["delays"], float), np.power(10.0, np.asarray(p["powers"], float) / 10.0)


def tdl_freq_cov_mat(model, subcarrier_spacing, fft_size, delay_spread, precision=None):
    """Frequency covariance matrix of a TDL model, R[u, v] = sum_l P_l exp(-j 2 pi tau_l df (u - v)), [fft_size, fft_size]
    (reference channel_estimation.py:1856-1953; host constant, like there)."""
    from ..config import config, dtypes
    los, delays, pw = _tdl_pdp(model)
    delays = delays * delay_spread
    if los:                                            # the specular and the first scattered tap share a delay (:1928-1931)
        pw = np.concatenate([[pw[0] + pw[1]], pw[2:]])
        delays = delays[1:]
    pw = pw / pw.sum()
    ph = np.exp(1j * (-2. * np.pi * subcarrier_spacing * np.arange(fft_size))[None, :] * delays[:, None])
    cov = np.einsum("l,lu,lv->uv", pw, ph, np.conj(ph))
    return wrap(torch.from_numpy(cov).to(dtypes[precision or config.precision]["torch"]["cdtype"]))


def tdl_time_cov_mat(model, speed, carrier_frequency, ofdm_symbol_duration, num_ofdm_symbols, los_angle_of_arrival=np.pi / 4.,
                     precision=None):
    """Time covariance matrix of a TDL model: Jakes' J0(nu dt (u - v)) of the Doppler spread nu = 2 pi v / c f_c, plus the
    specular term of the LoS models, [num_ofdm_symbols, num_ofdm_symbols] (reference channel_estimation.py:1956-2070)."""
    from scipy.special import jv
    from ..config import config, dtypes
    los, _, pw = _tdl_pdp(model)
    pw = pw / pw.sum()
    nu = 2. * np.pi * speed / 299792458. * carrier_frequency
    i = np.arange(num_ofdm_symbols)
    e = nu * ofdm_symbol_duration * (i[:, None] - i[None, :])
    cov = jv(0.0, e) * (pw[1:].sum() if los else pw.sum()) + (np.exp(1j * e * np.cos(los_angle_of_arrival)) * pw[0] if los else 0.)
    return wrap(torch.from_numpy(np.asarray(cov, np.complex128)).to(dtypes[precision or config.precision]["torch"]["cdtype"]))
