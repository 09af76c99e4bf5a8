"""OFDM MIMO equalisation - mirror of reference src/sionna/phy/ofdm/equalization.py
(``OFDMEqualizer`` :17-275, ``LMMSEEqualizer`` :277-344).  The whole ``call`` (layout changes,
interference-plus-noise covariance, per-RE LMMSE solve, stream re-ordering, data-symbol gather)
is ONE HIP kernel, ``samd_ofdm_lmmse_c64`` (csrc/mimo.hip)."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, pending_of


# kernel modes of samd_ofdm_lmmse_c64 / samd_lmmse_equalizer_c64 (csrc/mimo.hip)
MODE_LMMSE_NO_WHITENING, MODE_LMMSE, MODE_ZF, MODE_MF = 0, 1, 2, 3
# err_var argument forms of the fused per-RE kernels
EV_NONE, EV_TABLE, EV_FULL = 0, 1, 2


def _is_host_zero(err_var):
    """True if ``err_var`` is a HOST scalar equal to zero (Python / NumPy number, 0-d or 1-element NumPy array, CPU
    tensor).  Device tensors are never inspected: that would be a device-to-host synchronisation on every call of
    the config-4 hot path; a zero device tensor simply takes the table path and adds exact zeros."""
    if isinstance(err_var, (int, float, np.number)):
        return float(err_var) == 0.0
    if isinstance(err_var, np.ndarray):
        return err_var.size == 1 and float(err_var.reshape(-1)[0]) == 0.0
    if isinstance(err_var, torch.Tensor) and err_var.device.type == "cpu":
        return err_var.numel() == 1 and float(err_var.reshape(-1)[0]) == 0.0
    return False


def _same_tensor(a, b):
    """``a`` is (a view-free alias of) the tensor ``b``: a block's ``__call__`` re-wraps its outputs, so identity of the
    Python objects is too strict; the storage address, shape and dtype identify the estimator's own err_var."""
    return (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.shape == b.shape and a.dtype == b.dtype
            and a.device == b.device and a.data_ptr() == b.data_ptr())


class OFDMEqualizer(Block):
    """Base class kept for API parity.  Only the LMMSE equaliser has a HIP path; a user-supplied
    ``equalizer`` callable would have to run per resource element on the host, which this build
    deliberately does not offer."""

    def __init__(self, equalizer, resource_grid, stream_management, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if equalizer not in ("lmmse", "zf", "mf"):
            raise NotImplementedError("OFDMEqualizer: the fused kernel implements the 'lmmse', 'zf' and 'mf' equalisers")
        self._rg, self._sm = resource_grid, stream_management
        self._mode = {"lmmse": MODE_LMMSE, "zf": MODE_ZF, "mf": MODE_MF}[equalizer]
        self._dev = None

    def _tables(self):
        if self._dev is None:
            rg, sm = self._rg, self._sm
            s = rg.num_tx * rg.num_streams_per_tx
            nre = rg.num_ofdm_symbols * rg.num_effective_subcarriers
            di = rg._data_ind_eff()                                              # [S, num_data]
            data_pos = np.full((s, nre), -1, np.int32)
            np.put_along_axis(data_pos, di, np.arange(di.shape[1], dtype=np.int32)[None, :], axis=1)
            desired = np.asarray(sm.rx_stream_ids, np.int32)                     # [RX, K]
            all_ids = np.arange(s)
            undesired = np.stack([np.setdiff1d(all_ids, desired[i]) for i in range(sm.num_rx)]).astype(np.int32)
            i32 = lambda a: _ffi.to_device(np.ascontiguousarray(a, np.int32), torch.int32)
            self._dev = (i32(rg.effective_subcarrier_ind), i32(desired), i32(undesired) if undesired.size else None,
                         i32(data_pos), undesired.shape[1])
            # every stream detected by some receiver: the kernel writes every output element, no zero fill needed
            self._covers_all = np.array_equal(np.unique(desired), all_ids)
        return self._dev

    def _prepare(self, y, h_hat, err_var, no):
        """Device tensors + scalar arguments shared by the fused per-RE kernels."""
        rg, sm = self._rg, self._sm
        y = _ffi.to_device(y, torch.complex64)
        h_hat = _ffi.to_device(h_hat, torch.complex64)
        b, rx, m = y.shape[:3]
        s = rg.num_tx * rg.num_streams_per_tx
        t, f = rg.num_ofdm_symbols, rg.num_effective_subcarriers
        assert tuple(h_hat.shape) == (b, rx, m, rg.num_tx, rg.num_streams_per_tx, t, f), "unexpected h_hat shape"
        # err_var: broadcastable to h_hat.  Recognise the two compact forms, expand anything else.
        if _is_host_zero(err_var):
            ev_mode, ev_arg = EV_NONE, None
        else:
            ev = _ffi.to_device(err_var, torch.float32)
            ev = ev.reshape((1,) * (7 - ev.dim()) + tuple(ev.shape)) if ev.dim() < 7 else ev
            if ev.shape[:3] == (1, 1, 1):
                ev_mode = EV_TABLE
                ev_arg = torch.broadcast_to(ev, (1, 1, 1, rg.num_tx, rg.num_streams_per_tx, t, f)).contiguous()
            else:
                ev_mode, ev_arg = EV_FULL, torch.broadcast_to(ev, tuple(h_hat.shape)).contiguous()
        no = _ffi.to_device(no, torch.float32)
        no = torch.broadcast_to(no.reshape(tuple(no.shape) + (1,) * (3 - no.dim())), (b, rx, m)).contiguous()
        sc_ind, desired, undesired, data_pos, n_und = self._tables()
        keep = (y, h_hat, ev_arg, no)                       # keeps the temporaries alive until the launch
        head = (_ffi.ptr(y), _ffi.ptr(h_hat), _ffi.ptr(ev_arg), ev_mode, _ffi.ptr(no))
        tabs = (_ffi.ptr(sc_ind), _ffi.ptr(desired), _ffi.ptr(undesired), _ffi.ptr(data_pos))
        dims = (b, rx, m, s, sm.num_streams_per_rx, n_und, t, f, rg.fft_size, rg.num_data_symbols)
        return keep, head, tabs, dims

    def _double_inputs(self, y, h_hat, err_var, no):
        """precision="double": the pre-processing of the reference's OFDM receivers (ofdm/equalization.py:107-230 =
        ofdm/detection.py:229-287) as device tensor operations in complex128 -> (y_dt [B,rx,T,F,M], hd [B,rx,T,F,M,K],
        s [B,rx,T,F,M,M], extract, scatter): ``extract`` takes per-RE results [B,rx,T,F,K,...] to the data symbols of the
        streams [B,tx,s,num_data,...] (:232-275), ``scatter`` is its inverse with zeros at the other resource elements."""
        rg, sm = self._rg, self._sm
        dev = _ffi.device()
        y = _ffi.to_device(y, torch.complex128)
        h_hat = _ffi.to_device(h_hat, torch.complex128)
        ev = _ffi.to_device(err_var, torch.float64) if not _is_host_zero(err_var) else torch.zeros((), dtype=torch.float64, device=dev)
        no = _ffi.to_device(no, torch.float64)
        sc = torch.as_tensor(np.asarray(rg.effective_subcarrier_ind), dtype=torch.int64, device=dev)
        y_eff = y.index_select(-1, sc)                                            # remove nulled subcarriers
        y_dt = y_eff.permute(0, 1, 3, 4, 2)                                        # [B,rx,T,F,M]
        ev = torch.broadcast_to(ev, h_hat.shape).permute(0, 1, 5, 6, 2, 3, 4)
        ev = ev.reshape(tuple(ev.shape[:5]) + (-1,))                               # [B,rx,T,F,M,tx*s]
        h_dt = h_hat.permute(1, 3, 4, 0, 2, 5, 6)
        h_dt = h_dt.reshape((-1,) + tuple(h_dt.shape[3:]))                         # [rx*tx*s,B,M,T,F]
        ind_d = torch.as_tensor(np.asarray(sm.detection_desired_ind), dtype=torch.int64, device=dev)
        ind_u = torch.as_tensor(np.asarray(sm.detection_undesired_ind), dtype=torch.int64, device=dev)
        hd = h_dt.index_select(0, ind_d).reshape((sm.num_rx, sm.num_streams_per_rx) + tuple(h_dt.shape[1:]))
        hd = hd.permute(2, 0, 4, 5, 3, 1)                                          # [B,rx,T,F,M,K]
        m = y_dt.shape[-1]
        no_dt = no.reshape(tuple(no.shape) + (1,) * (3 - no.dim()))
        no_dt = torch.broadcast_to(no_dt, tuple(y.shape[:3]))[..., None, None]
        no_dt = torch.broadcast_to(no_dt, tuple(y_eff.shape)).permute(0, 1, 3, 4, 2)   # [B,rx,T,F,M]
        eye = torch.eye(m, dtype=torch.float64, device=dev)
        s = (no_dt[..., None] * eye + ev.sum(-1)[..., None] * eye).to(torch.complex128)
        if ind_u.numel() > 0:
            hu = h_dt.index_select(0, ind_u).reshape((sm.num_rx, -1) + tuple(h_dt.shape[1:])).permute(2, 0, 4, 5, 3, 1)
            s = s + hu @ hu.conj().transpose(-1, -2)
        stream_ind = torch.as_tensor(np.asarray(sm.stream_ind), dtype=torch.int64, device=dev)
        di = torch.as_tensor(np.asarray(rg._data_ind_eff()), dtype=torch.int64, device=dev)       # [tx*s, num_data]
        di = di.reshape(rg.num_tx, rg.num_streams_per_tx, -1)

        def extract(z):                                                            # [B,rx,T,F,K,...] -> [B,tx,s,num_data,...]
            z = z.as_subclass(torch.Tensor)
            tail = tuple(z.shape[5:])
            z = z.reshape(tuple(z.shape[:5]) + (-1,)).permute(1, 4, 2, 3, 0, 5)    # [rx,K,T,F,B,E]
            z = z.reshape((-1,) + tuple(z.shape[2:])).index_select(0, stream_ind)  # streams in transmitter order
            z = z.reshape((rg.num_tx, rg.num_streams_per_tx, -1) + tuple(z.shape[-2:]))             # [tx,s,T*F,B,E]
            z = torch.gather(z, 2, di[..., None, None].expand(-1, -1, -1, z.shape[-2], z.shape[-1]))
            return z.permute(3, 0, 1, 2, 4).reshape((z.shape[3], rg.num_tx, rg.num_streams_per_tx, -1) + tail).contiguous()

        def scatter(z):                                                            # [B,tx,s,num_data,...] -> [B,rx,T,F,K,...]
            z = z.as_subclass(torch.Tensor)
            tail = tuple(z.shape[4:])
            b = z.shape[0]
            z = z.reshape(tuple(z.shape[:4]) + (-1,)).permute(1, 2, 3, 0, 4)       # [tx,s,ND,B,E]
            t_, f_ = y_dt.shape[2], y_dt.shape[3]
            full = torch.zeros((rg.num_tx, rg.num_streams_per_tx, t_ * f_) + tuple(z.shape[-2:]), dtype=z.dtype, device=dev)
            full.scatter_(2, di[..., None, None].expand(-1, -1, -1, z.shape[-2], z.shape[-1]), z)
            full = full.reshape((-1, t_, f_) + tuple(z.shape[-2:]))                # [tx*s,T,F,B,E]
            inv = torch.empty_like(stream_ind)
            inv[stream_ind] = torch.arange(stream_ind.numel(), device=dev)
            full = full.index_select(0, inv).reshape((sm.num_rx, sm.num_streams_per_rx, t_, f_) + tuple(z.shape[-2:]))
            return full.permute(4, 0, 2, 3, 1, 5).reshape((b, sm.num_rx, t_, f_, sm.num_streams_per_rx) + tail).contiguous()

        return y_dt.contiguous(), hd.contiguous(), s.contiguous(), extract, scatter

    def _call_double(self, y, h_hat, err_var, no):
        """precision="double": ``_double_inputs`` around the complex128 equaliser kernel (the single-precision path fuses
        all of it into one launch; this one exists for analysis runs)."""
        from ..mimo.equalization import _equalize_f64
        from ..block import wrap
        y_dt, hd, s, extract, _ = self._double_inputs(y, h_hat, err_var, no)
        x_hat, no_eff = _equalize_f64(y_dt, hd, s, int(self._mode), type(self).__name__)
        return wrap(extract(x_hat)), wrap(extract(no_eff))

    def _fused_lsnn(self, y, h_hat, err_var, no, demap=None):
        """``h_hat`` still deferred by ``LSChannelEstimator(interpolation_type="nn")`` of the same resource grid, its own
        ``err_var``, LMMSE, no interfering streams: ONE launch of samd_ofdm_lsnn_lmmse_c64 forms the estimate where it
        is used (the estimator's own product, hence the bits of the separate launches) and h_hat is never written.
        ``demap`` = (num_bits_per_symbol, maxlog, hard_out, pam_levels): the LLRs instead of (x_hat, no_eff).
        Returns None when the recipe does not apply (the caller then takes the general path, which fills h_hat)."""
        pend = pending_of(h_hat)
        if (pend is None or pend.kind != "ls_nn" or pend.rg is not self._rg or self._mode != MODE_LMMSE
                or self.precision != "single" or not _same_tensor(err_var, pend.err_var)):
            return None
        pend.check_guard()                                    # the recipe's inputs (y, no) unchanged since the estimator ran
        rg, sm = self._rg, self._sm
        sc_ind, desired, undesired, data_pos, n_und = self._tables()
        if n_und:
            return None
        y = _ffi.to_device(y, torch.complex64)
        b, rx, m = y.shape[:3]
        if tuple(y.shape) != tuple(pend.y.shape):
            return None
        no = _ffi.to_device(no, torch.float32)
        no = torch.broadcast_to(no.reshape(tuple(no.shape) + (1,) * (3 - no.dim())), (b, rx, m)).contiguous()
        s, nd = rg.num_tx * rg.num_streams_per_tx, rg.num_data_symbols
        alloc = torch.empty if self._covers_all else torch.zeros
        x_hat = no_eff = llr = None
        if demap is None:
            nbits, maxlog, hard, lev = 0, 0, 0, None
            x_hat = alloc((b, rg.num_tx, rg.num_streams_per_tx, nd), dtype=torch.complex64, device=y.device)
            no_eff = alloc((b, rg.num_tx, rg.num_streams_per_tx, nd), dtype=torch.float32, device=y.device)
        else:
            nbits, maxlog, hard, lev = demap
            llr = alloc((b, rg.num_tx, rg.num_streams_per_tx, nd * nbits), dtype=torch.float32, device=y.device)
        rc = _ffi.lib().samd_ofdm_lsnn_lmmse_c64(
            _ffi.ptr(y), _ffi.ptr(pend.y), _ffi.ptr(pend.src), _ffi.ptr(pend.coef), _ffi.ptr(pend.ev), _ffi.ptr(pend.no),
            pend.no.numel(), _ffi.ptr(no), _ffi.ptr(sc_ind), _ffi.ptr(desired), _ffi.ptr(data_pos), b, rx, m, s,
            sm.num_streams_per_rx, rg.num_ofdm_symbols, rg.num_effective_subcarriers, rg.fft_size, nd, int(nbits), int(maxlog),
            int(hard), _ffi.ptr(lev), _ffi.ptr(x_hat), _ffi.ptr(no_eff), _ffi.ptr(llr), _ffi.stream())
        if rc == _ffi.ERR_UNSUPPORTED:
            return None
        _ffi.check(rc, type(self).__name__ + "(fused LS-NN)")
        return llr if demap is not None else (x_hat, no_eff)

    def call(self, y, h_hat, err_var, no):
        if self.precision == "double":
            return self._call_double(y, h_hat, err_var, no)
        fused = self._fused_lsnn(y, h_hat, err_var, no)
        if fused is not None:
            return fused
        rg = self._rg
        keep, head, tabs, dims = self._prepare(y, h_hat, err_var, no)
        b, nd, dev = dims[0], rg.num_data_symbols, keep[0].device
        alloc = torch.empty if self._covers_all else torch.zeros
        x_hat = alloc((b, rg.num_tx, rg.num_streams_per_tx, nd), dtype=torch.complex64, device=dev)
        no_eff = alloc((b, rg.num_tx, rg.num_streams_per_tx, nd), dtype=torch.float32, device=dev)
        _ffi.check(_ffi.lib().samd_ofdm_lmmse_c64(*head, *tabs, *dims, int(self._mode), _ffi.ptr(x_hat),
                                                  _ffi.ptr(no_eff), _ffi.stream()), "LMMSEEqualizer")
        return x_hat, no_eff


class LMMSEEqualizer(OFDMEqualizer):
    """``LMMSEEqualizer(resource_grid, stream_management, whiten_interference=True)``
    ``(y, h_hat, err_var, no) -> (x_hat, no_eff)`` [batch, num_tx, num_streams, num_data_symbols]."""

    def __init__(self, resource_grid, stream_management, whiten_interference=True, precision=None, **kwargs):
        super().__init__("lmmse", resource_grid, stream_management, precision=precision, **kwargs)
        self._mode = MODE_LMMSE if whiten_interference else MODE_LMMSE_NO_WHITENING


class ZFEqualizer(OFDMEqualizer):
    """``ZFEqualizer(resource_grid, stream_management)(y, h_hat, err_var, no)`` (ofdm/equalization.py:346-403)."""

    def __init__(self, resource_grid, stream_management, precision=None, **kwargs):
        super().__init__("zf", resource_grid, stream_management, precision=precision, **kwargs)


class MFEqualizer(OFDMEqualizer):
    """``MFEqualizer(resource_grid, stream_management)(y, h_hat, err_var, no)`` (ofdm/equalization.py:405-460)."""

    def __init__(self, resource_grid, stream_management, precision=None, **kwargs):
        super().__init__("mf", resource_grid, stream_management, precision=precision, **kwargs)
