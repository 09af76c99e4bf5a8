"""Constellations, mapper, demapper and the binary source - host-side mirror of
``sionna.phy.mapping`` (reference src/sionna/phy/mapping.py) for the hot path.

Constellation tables are built on the host with NumPy (init time); Mapper / Demapper /
BinarySource call the HIP kernels in ``csrc/mapping.hip`` / ``csrc/channel.hip`` through
the C-ABI.  Out of scope: LLRs2SymbolLogits,
SymbolLogits2Moments, QAM2PAM.
"""
import numpy as np
import torch

from .. import _ffi
from .block import Block, Object
from .config import config, dtypes, PhiloxGenerator


def _gray_pam_levels(num_bits):
    """Gray-labelled PAM level for every label 0..2^num_bits-1 (MSB first).

    Closed iteration of the recursion of 38.211 Sec. 5.1 (reference mapping.py:15-42):
    the last bit b gives 1-2b; prepending bit b maps v -> (1-2b) * (2^j - v), j = number of
    bits already consumed.
    """
    labels = np.arange(2 ** num_bits)
    bits = (labels[:, None] >> np.arange(num_bits - 1, -1, -1)) & 1      # [2^nb, nb], MSB first
    v = 1 - 2 * bits[:, -1]
    for j in range(1, num_bits):
        v = (1 - 2 * bits[:, num_bits - 1 - j]) * (2 ** j - v)
    return v


def pam_gray(b):
    """PAM level of the bit vector ``b`` (reference mapping.py:15-42)."""
    b = np.asarray(b, dtype=np.int64)
    label = int(np.sum(b << np.arange(len(b) - 1, -1, -1)))
    return int(_gray_pam_levels(len(b))[label])


def _precision_dtypes(precision):
    p = config.precision if precision is None else precision
    return dtypes[p]["np"]["rdtype"], dtypes[p]["np"]["cdtype"]


def qam(num_bits_per_symbol, normalize=True, precision=None):
    """QAM constellation, label of point i = binary representation of i; even label bits
    drive the real axis, odd ones the imaginary axis (reference mapping.py:44-118)."""
    if num_bits_per_symbol % 2 != 0 or num_bits_per_symbol <= 0:
        raise ValueError("num_bits_per_symbol must be a multiple of 2")
    assert isinstance(normalize, bool), "normalize must be boolean"
    rdtype, cdtype = _precision_dtypes(precision)
    m = int(num_bits_per_symbol)
    n = m // 2
    idx = np.arange(2 ** m)
    bits = (idx[:, None] >> np.arange(m - 1, -1, -1)) & 1
    w = 1 << np.arange(n - 1, -1, -1)
    lev = _gray_pam_levels(n)
    c = (lev[bits[:, 0::2] @ w] + 1j * lev[bits[:, 1::2] @ w]).astype(cdtype)
    if normalize:
        qam_var = 1 / (2 ** (n - 2)) * np.sum(np.linspace(1, 2 ** n - 1, 2 ** (n - 1), dtype=rdtype) ** 2)
        c /= np.sqrt(qam_var)
    return c


def pam(num_bits_per_symbol, normalize=True, precision=None):
    """PAM constellation (reference mapping.py:120-193)."""
    if num_bits_per_symbol <= 0:
        raise ValueError("num_bits_per_symbol must be positive")
    assert isinstance(normalize, bool), "normalize must be boolean"
    rdtype, cdtype = _precision_dtypes(precision)
    n = int(num_bits_per_symbol)
    c = _gray_pam_levels(n).astype(cdtype)
    if normalize:
        pam_var = 1 / (2 ** (n - 1)) * np.sum(np.linspace(1, 2 ** n - 1, 2 ** (n - 1), dtype=rdtype) ** 2)
        c /= np.sqrt(pam_var)
    return c


class Constellation(Block):
    """reference mapping.py:195-420"""

    def __init__(self, constellation_type, num_bits_per_symbol, points=None, normalize=False,
                 center=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if constellation_type not in ("qam", "pam", "custom"):
            raise ValueError(f"Wrong `constellation_type` {constellation_type}")
        self._constellation_type = constellation_type
        if num_bits_per_symbol is None:
            raise ValueError("No value for `num_bits_per_symbol`")
        n = num_bits_per_symbol
        if (n <= 0) or (n % 1 != 0):
            raise ValueError("`num_bits_per_symbol` must be a positive integer")
        if constellation_type == "qam" and n % 2 != 0:
            raise ValueError("`num_bits_per_symbol` must be a positive integer multiple of 2")
        self._num_bits_per_symbol = int(n)
        self._num_points = 2 ** self._num_bits_per_symbol
        self.normalize = normalize
        self.center = center
        if points is not None and constellation_type != "custom":
            raise ValueError("`points` can only be provided for `constellation_type`='custom'")
        if points is None and constellation_type == "custom":
            raise ValueError("You must provide a value for `points`")
        self._points = None
        self._dev = None
        self._dev64 = None
        if constellation_type == "qam":
            points = qam(self._num_bits_per_symbol, normalize=True, precision=self.precision)
        elif constellation_type == "pam":
            points = pam(self._num_bits_per_symbol, normalize=True, precision=self.precision)
        self.points = points

    constellation_type = property(lambda self: self._constellation_type)
    num_bits_per_symbol = property(lambda self: self._num_bits_per_symbol)
    num_points = property(lambda self: self._num_points)

    @property
    def normalize(self):
        return self._normalize

    @normalize.setter
    def normalize(self, value):
        assert isinstance(value, bool), "`normalize` must be boolean"
        self._normalize = value
        self._dev = None
        self._dev64 = None

    @property
    def center(self):
        return self._center

    @center.setter
    def center(self, value):
        assert isinstance(value, bool), "`center` must be boolean"
        self._center = value
        self._dev = None
        self._dev64 = None

    @property
    def points(self):
        """[2**num_bits_per_symbol] complex ndarray (host copy)."""
        return self._points

    @points.setter
    def points(self, v):
        if self._points is not None and self._constellation_type != "custom":
            raise ValueError("`points` can only be modified for custom constellations")
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.shape != (2 ** self._num_bits_per_symbol,):
            raise ValueError("`points` must have shape [2**num_bits_per_symbol]")
        self._points = v.astype(dtypes[self.precision]["np"]["cdtype"])
        self._dev = None
        self._dev64 = None

    def _host_points(self, raw=False):
        x = self._points
        if self._constellation_type == "custom" and not raw:
            if self._center:
                x = x - np.mean(x)
            if self._normalize:
                x = x / np.sqrt(np.mean(np.abs(x) ** 2)).astype(x.real.dtype)
        return x.astype(self._points.dtype)

    def _raw_differs(self):
        return self._constellation_type == "custom" and (self._center or self._normalize)

    def pam_levels(self, raw=False):
        """Device float32[2^(m/2)] PAM levels if the constellation is a square QAM whose label
        interleaves two identical PAM axes (true for ``qam()``), else None.  ``raw``: see ``device_points``."""
        if raw and self._raw_differs():
            return None                          # custom points: the generic 2^m-point kernel on the raw points
        if getattr(self, "_pam_valid", False) and self._dev is not None:
            return self._pam_dev
        self._pam_dev, self._pam_valid = None, True
        m = self._num_bits_per_symbol
        if m % 2 != 0:
            return None
        pts = self._host_points().astype(np.complex64)
        nb = m // 2
        idx = np.arange(2 ** m)
        bits = (idx[:, None] >> np.arange(m - 1, -1, -1)) & 1
        w = 1 << np.arange(nb - 1, -1, -1)
        li, lq = bits[:, 0::2] @ w, bits[:, 1::2] @ w
        lev = np.zeros(2 ** nb, np.float32)
        lev[li] = pts.real                       # level of each I label (last write wins; checked below)
        self.device_points()                     # (re)creates _dev, the cache-validity marker
        if np.array_equal(lev[li], pts.real) and np.array_equal(lev[lq], pts.imag):
            self._pam_dev = _ffi.to_device(lev, torch.float32)
        return self._pam_dev

    def device_points(self, double=False, raw=False):
        """Device copy (complex64, or complex128 for precision="double") used by the kernels; rebuilt after a
        setter call.  ``raw=True``: the stored ``points`` WITHOUT centring / normalisation - what the reference's
        ``Demapper`` and ``SymbolDemapper`` measure distances to (mapping.py:667-668, 777: ``constellation.points``),
        while its ``Mapper`` transmits ``constellation()`` (mapping.py:514).  They differ only for a "custom"
        constellation with ``normalize`` or ``center`` set."""
        if raw and self._raw_differs():
            pts = self._host_points(raw=True)
            if double:
                return _ffi.to_device(np.asarray(pts, np.complex128), torch.complex128)
            return _ffi.to_device(pts.astype(np.complex64), torch.complex64)
        if double:
            if getattr(self, "_dev64", None) is None or self._dev is None:
                self._dev64 = _ffi.to_device(np.asarray(self._host_points(), np.complex128), torch.complex128)
        if self._dev is None:
            self._dev = _ffi.to_device(self._host_points().astype(np.complex64), torch.complex64)
        return self._dev64 if double else self._dev

    def __call__(self):
        return self.call()

    def call(self):
        """(Possibly) centred and normalised constellation points."""
        return self._host_points()

    @staticmethod
    def check_or_create(*, constellation_type=None, num_bits_per_symbol=None, constellation=None,
                        precision=None):
        if isinstance(constellation, Constellation):
            return constellation
        if constellation_type in ["qam", "pam"]:
            return Constellation(constellation_type, num_bits_per_symbol, precision=precision)
        raise ValueError("You must provide a valid `constellation`")


class Mapper(Block):
    """bits [..., n] -> symbols [..., n/num_bits_per_symbol] (reference mapping.py:422-519)."""

    def __init__(self, constellation_type=None, num_bits_per_symbol=None, constellation=None,
                 return_indices=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        self._return_indices = return_indices

    constellation = property(lambda self: self._constellation)

    def call(self, bits):
        m = self._constellation.num_bits_per_symbol
        bits = _ffi.to_device(bits, torch.float32)
        if bits.shape[-1] % m != 0:
            raise ValueError("last dimension must be a multiple of num_bits_per_symbol")
        out_shape = tuple(bits.shape[:-1]) + (bits.shape[-1] // m,)
        if self.precision == "double":
            # a table look-up, no arithmetic: the complex128 points are gathered by the symbol index
            w = (1 << torch.arange(m - 1, -1, -1, device=bits.device)).to(torch.int64)
            ind = (bits.reshape(out_shape + (m,)).to(torch.int64) * w).sum(-1)
            x = self._constellation.device_points(double=True)[ind]
            return (x, ind.to(torch.int32)) if self._return_indices else x
        x = torch.empty(out_shape, dtype=torch.complex64, device=bits.device)
        ns = x.numel()
        _ffi.check(_ffi.lib().samd_qam_map_c64(_ffi.ptr(bits), _ffi.ptr(self._constellation.device_points()),
                                               m, ns, _ffi.ptr(x), _ffi.stream()), "Mapper")
        if self._return_indices:
            w = (1 << torch.arange(m - 1, -1, -1, device=bits.device)).to(torch.int32)
            ind = (bits.reshape(out_shape + (m,)).to(torch.int32) * w).sum(-1).to(torch.int32)
            return x, ind
        return x


class Demapper(Block):
    """LLRs (logits) for every bit of every received symbol (reference mapping.py:521-691,
    794-967).  ``call(y, no, prior=None)``; ``prior``: LLRs [num_bits_per_symbol] or [..., n, num_bits_per_symbol]."""

    def __init__(self, demapping_method, constellation_type=None, num_bits_per_symbol=None,
                 constellation=None, hard_out=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert demapping_method in ("app", "maxlog"), "Unknown demapping method"
        self._method = demapping_method
        self._hard_out = hard_out
        # square-QAM fast path (per-axis evaluation); `separable=False` forces the generic
        # 2^m-point kernel (used by the tests to cross-check the two)
        self._separable = bool(kwargs.get("separable", True))
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)

    constellation = property(lambda self: self._constellation)

    def _call_double(self, y, no, prior):
        """precision="double": float64 kernel samd_qam_demap_f64 (csrc/f64.hip)."""
        m = self._constellation.num_bits_per_symbol
        y = _ffi.to_device(y, torch.complex128)
        no = _ffi.to_device(no, torch.float64)
        no = no.reshape(1) if no.numel() == 1 else torch.broadcast_to(no, y.shape).contiguous()
        out = torch.empty(tuple(y.shape[:-1]) + (y.shape[-1] * m,), dtype=torch.float64, device=y.device)
        if prior is not None:
            prior = _ffi.to_device(prior, torch.float64)
            if prior.dim() == 1:
                assert prior.numel() == m, "prior must have num_bits_per_symbol entries"
            else:
                prior = torch.broadcast_to(prior, tuple(y.shape) + (m,)).contiguous()
        _ffi.check(_ffi.lib().samd_qam_demap_f64(
            _ffi.ptr(y), _ffi.ptr(no), no.numel(), _ffi.ptr(self._constellation.device_points(double=True, raw=True)), m, y.numel(),
            _ffi.ptr(prior), 0 if prior is None else prior.numel(), 0 if self._method == "app" else 1,
            int(bool(self._hard_out)), _ffi.ptr(out), _ffi.stream()), "Demapper(double)")
        return out

    def call(self, y, no, prior=None):
        if self.precision == "double":
            return self._call_double(y, no, prior)
        m = self._constellation.num_bits_per_symbol
        y = _ffi.to_device(y, torch.complex64)
        no = _ffi.to_device(no, torch.float32)
        if no.numel() == 1:
            no = no.reshape(1)
        else:
            no = torch.broadcast_to(no, y.shape).contiguous()
        out = torch.empty(tuple(y.shape[:-1]) + (y.shape[-1] * m,), dtype=torch.float32, device=y.device)
        if prior is not None:                                     # generic 2^m-point kernel with the a-priori term
            prior = _ffi.to_device(prior, torch.float32)
            if prior.dim() == 1:
                assert prior.numel() == m, "prior must have num_bits_per_symbol entries"
                prior = prior.contiguous()
            else:
                prior = torch.broadcast_to(prior, tuple(y.shape) + (m,)).contiguous()
            _ffi.check(_ffi.lib().samd_qam_demap_prior_f32(
                _ffi.ptr(y), _ffi.ptr(no), no.numel(), _ffi.ptr(self._constellation.device_points(raw=True)), m, y.numel(),
                _ffi.ptr(prior), prior.numel(), 0 if self._method == "app" else 1, int(bool(self._hard_out)), _ffi.ptr(out),
                _ffi.stream()), "Demapper(prior)")
            return out
        lev = self._constellation.pam_levels(raw=True) if self._separable else None
        if lev is not None:
            _ffi.check(_ffi.lib().samd_square_qam_demap_f32(
                _ffi.ptr(y), _ffi.ptr(no), no.numel(), _ffi.ptr(lev), m, y.numel(),
                0 if self._method == "app" else 1, int(bool(self._hard_out)), _ffi.ptr(out), _ffi.stream()),
                "Demapper")
            return out
        _ffi.check(_ffi.lib().samd_qam_demap_f32(
            _ffi.ptr(y), _ffi.ptr(no), no.numel(), _ffi.ptr(self._constellation.device_points(raw=True)), m,
            y.numel(), 0 if self._method == "app" else 1, int(bool(self._hard_out)), _ffi.ptr(out),
            _ffi.stream()), "Demapper")
        return out


class SymbolDemapper(Block):
    """``SymbolDemapper(constellation_type=None, num_bits_per_symbol=None, constellation=None, hard_out=False)``
    ``(y, no, prior=None)``: normalised logits (log-probabilities) of the constellation points for every received symbol,
    [..., n, num_points] - or, with ``hard_out``, the index of the most likely point, [..., n] int32 (reference
    mapping.py:693-792).  ``prior``: log-probabilities [num_points] or [..., n, num_points]."""

    def __init__(self, constellation_type=None, num_bits_per_symbol=None, constellation=None, hard_out=False,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._hard_out = hard_out
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)

    constellation = property(lambda self: self._constellation)

    def call(self, y, no, prior=None):
        dbl = self.precision == "double"       # float64: csrc/f64_mapping.hip
        m = self._constellation.num_bits_per_symbol
        npts = 1 << m
        y = _ffi.to_device(y, self.cdtype)
        no = _ffi.to_device(no, self.rdtype)
        no = no.reshape(1) if no.numel() == 1 else torch.broadcast_to(no, y.shape).contiguous()
        if prior is not None:
            prior = _ffi.to_device(prior, self.rdtype)
            prior = prior.contiguous() if prior.dim() == 1 else torch.broadcast_to(prior, tuple(y.shape) + (npts,)).contiguous()
            assert prior.shape[-1] == npts, "prior must have num_points entries"
        hard = bool(self._hard_out)
        out = None if hard else torch.empty(tuple(y.shape) + (npts,), dtype=self.rdtype, device=y.device)
        idx = torch.empty(tuple(y.shape), dtype=torch.int32, device=y.device) if hard else None
        pts = _ffi.to_device(np.asarray(self._constellation.points, np.complex128), torch.complex128) if dbl \
            else self._constellation.device_points(raw=True)
        _ffi.check((_ffi.lib().samd_symbol_demap_f64 if dbl else _ffi.lib().samd_symbol_demap_f32)(
            _ffi.ptr(y), _ffi.ptr(no), no.numel(), _ffi.ptr(pts), m, y.numel(),
            _ffi.ptr(prior), 0 if prior is None else prior.numel(), int(hard), _ffi.ptr(out), _ffi.ptr(idx), _ffi.stream()),
            "SymbolDemapper")
        return idx if hard else out


class SymbolLogits2LLRs(Block):
    """``SymbolLogits2LLRs(method, num_bits_per_symbol, *, hard_out=False)(logits [..., n, 2^m], prior=None)`` ->
    LLRs or hard decisions [..., n, m] (reference mapping.py:794-967): for every bit the reduction (logsumexp for "app",
    max for "maxlog") over the points whose label has the bit set minus the one over the others; ``prior`` [m] or
    [..., n, m] LLRs enter through log_sigmoid(+-prior_i).  One launch of ``samd_symbol_logits2llrs_f32``."""

    def __init__(self, method, num_bits_per_symbol, *, hard_out=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert method in ("app", "maxlog"), "Unknown demapping method"
        self._method, self._hard_out, self._num_bits_per_symbol = method, hard_out, int(num_bits_per_symbol)

    num_bits_per_symbol = property(lambda self: self._num_bits_per_symbol)

    def call(self, logits, prior=None):
        dbl = self.precision == "double"
        m = self._num_bits_per_symbol
        z = _ffi.to_device(logits, self.rdtype).contiguous()
        assert z.shape[-1] == 1 << m, "the last dimension of logits must be 2**num_bits_per_symbol"
        rows = z.numel() // (1 << m)
        out = torch.empty(tuple(z.shape[:-1]) + (m,), dtype=self.rdtype, device=z.device)
        pr, plen = None, 0
        if prior is not None:
            pr = _ffi.to_device(prior, self.rdtype)
            if pr.dim() > 1 or rows == 1:
                pr = torch.broadcast_to(pr, tuple(z.shape[:-1]) + (m,))
            pr = pr.contiguous()
            plen = pr.numel()
            assert plen in (m, rows * m), "prior must be [num_bits_per_symbol] or broadcastable to [..., n, num_bits_per_symbol]"
        _ffi.check((_ffi.lib().samd_symbol_logits2llrs_f64 if dbl else _ffi.lib().samd_symbol_logits2llrs_f32)(_ffi.ptr(z), m, rows, _ffi.ptr(pr) if pr is not None else None, plen,
                                                          1 if self._method == "maxlog" else 0, 1 if self._hard_out else 0,
                                                          _ffi.ptr(out), _ffi.stream()), "SymbolLogits2LLRs")
        return out


class LLRs2SymbolLogits(Block):
    """``LLRs2SymbolLogits(num_bits_per_symbol, hard_out=False)(llrs [..., n, m])`` -> logits (unnormalised
    log-probabilities) of the 2^m constellation points, [..., n, 2^m], or with ``hard_out`` the index of the most likely
    point, [..., n] int32 (reference mapping.py:969-1058): logit_c = sum_j log_sigmoid(l(c)_j llr_j) with the label of c =
    binary representation of c, 0 replaced by -1.  One launch of ``samd_llrs2symbol_logits_f32``."""

    def __init__(self, num_bits_per_symbol, hard_out=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._hard_out = hard_out
        self._num_bits_per_symbol = int(num_bits_per_symbol)

    num_bits_per_symbol = property(lambda self: self._num_bits_per_symbol)

    def call(self, llrs):
        dbl = self.precision == "double"
        m = self._num_bits_per_symbol
        x = _ffi.to_device(llrs, self.rdtype).contiguous()
        assert x.shape[-1] == m, "the last dimension of llrs must be num_bits_per_symbol"
        rows = x.numel() // m
        hard = bool(self._hard_out)
        out = None if hard else torch.empty(tuple(x.shape[:-1]) + (1 << m,), dtype=self.rdtype, device=x.device)
        idx = torch.empty(tuple(x.shape[:-1]), dtype=torch.int32, device=x.device) if hard else None
        _ffi.check((_ffi.lib().samd_llrs2symbol_logits_f64 if dbl else _ffi.lib().samd_llrs2symbol_logits_f32)(_ffi.ptr(x), m, rows, int(hard), _ffi.ptr(out), _ffi.ptr(idx),
                                                          _ffi.stream()), "LLRs2SymbolLogits")
        return idx if hard else out


class SymbolLogits2Moments(Block):
    """``SymbolLogits2Moments(constellation_type=None, num_bits_per_symbol=None, constellation=None)``
    ``(logits [..., n, num_points])`` -> (mean [..., n], var [..., n]) of the constellation under softmax(logits)
    (reference mapping.py:1061-1138; the mean is real-valued float like the reference's ``tf.reduce_sum(p * points)``
    when the constellation is real - here always returned as the block's complex dtype, the reference casts p to complex
    too).  One launch of ``samd_symbol_logits2moments_c64``."""

    def __init__(self, constellation_type=None, num_bits_per_symbol=None, constellation=None, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)

    def call(self, logits):
        dbl = self.precision == "double"
        m = self._constellation.num_bits_per_symbol
        z = _ffi.to_device(logits, self.rdtype).contiguous()
        assert z.shape[-1] == 1 << m, "the last dimension of logits must be the number of constellation points"
        rows = z.numel() >> m
        mean = torch.empty(tuple(z.shape[:-1]), dtype=self.cdtype, device=z.device)
        var = torch.empty(tuple(z.shape[:-1]), dtype=self.rdtype, device=z.device)
        pts = _ffi.to_device(np.asarray(self._constellation.points, np.complex128), torch.complex128) if dbl \
            else self._constellation.device_points()
        _ffi.check((_ffi.lib().samd_symbol_logits2moments_c128 if dbl else _ffi.lib().samd_symbol_logits2moments_c64)(_ffi.ptr(z), _ffi.ptr(pts), m, rows, _ffi.ptr(mean),
                                                             _ffi.ptr(var), _ffi.stream()), "SymbolLogits2Moments")
        return mean, var


def _bit_labels(num_bits):
    """[2^num_bits, num_bits]: binary representation of every index, MSB first (mapping.py:1165-1170)."""
    idx = np.arange(1 << num_bits)
    return ((idx[:, None] >> np.arange(num_bits - 1, -1, -1)[None, :]) & 1)


class SymbolInds2Bits(Block):
    """``SymbolInds2Bits(num_bits_per_symbol)(symbol_ind int [...])`` -> [..., num_bits_per_symbol] float: the binary
    representation of the indices (reference mapping.py:1141-1178: a gather from the label table; index plumbing, no
    arithmetic)."""

    def __init__(self, num_bits_per_symbol, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._num_bits_per_symbol = int(num_bits_per_symbol)
        self._labels = None

    def call(self, symbol_ind):
        ind = _ffi.to_device(symbol_ind, torch.int64)
        if self._labels is None:
            self._labels = _ffi.to_device(_bit_labels(self._num_bits_per_symbol).astype(np.float32), torch.float32).to(self.rdtype)
        return self._labels.index_select(0, ind.reshape(-1)).reshape(tuple(ind.shape) + (self._num_bits_per_symbol,))


class QAM2PAM(Object):
    """``QAM2PAM(num_bits_per_symbol)(ind_qam int [...])`` -> (ind_pam1, ind_pam2) int32: the indices of the real and the
    imaginary PAM component of QAM point indices - the even / odd bits of the label (reference mapping.py:1181-1231)."""

    def __init__(self, num_bits_per_symbol, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        lab = _bit_labels(int(num_bits_per_symbol))
        base = 1 << np.arange(int(num_bits_per_symbol) // 2 - 1, -1, -1)
        self._tab = (np.sum(lab[:, 0::2] * base, -1).astype(np.int32), np.sum(lab[:, 1::2] * base, -1).astype(np.int32))
        self._dev = None

    def __call__(self, ind_qam):
        ind = _ffi.to_device(ind_qam, torch.int64)
        if self._dev is None:
            self._dev = tuple(_ffi.to_device(t, torch.int32) for t in self._tab)
        flat = ind.reshape(-1)
        return tuple(t.index_select(0, flat).reshape(ind.shape) for t in self._dev)


class PAM2QAM(Object):
    """``PAM2QAM(num_bits_per_symbol, hard_in_out=True)(pam1, pam2)``: indices (int, [...]) or logits (float,
    [..., 2^(num_bits_per_symbol/2)]) of the two PAM constellations -> indices / logits [..., 2^num_bits_per_symbol] of the
    QAM constellation whose labels interleave theirs (reference mapping.py:1234-1314).  Indices are a table lookup; logits
    are one launch of ``samd_pam2qam_logits_f32`` (the reference's flatten-and-gather, literally)."""

    def __init__(self, num_bits_per_symbol, hard_in_out=True, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        nb = int(num_bits_per_symbol)
        assert nb % 2 == 0, "num_bits_per_symbol must be even"
        self._nb, self._hard_in_out = nb, hard_in_out
        P = 1 << (nb // 2)
        lab = _bit_labels(nb // 2)
        b = np.zeros([P, P, nb], np.int64)
        b[:, :, 0::2] = lab[:, None, :]
        b[:, :, 1::2] = lab[None, :, :]
        self._qam_ind = np.sum(b * (1 << np.arange(nb - 1, -1, -1)), -1).astype(np.int32)       # [P, P]
        self._dev = None

    def __call__(self, pam1, pam2):
        P = 1 << (self._nb // 2)
        if self._hard_in_out:
            i1, i2 = _ffi.to_device(pam1, torch.int64), _ffi.to_device(pam2, torch.int64)
            if self._dev is None:
                self._dev = _ffi.to_device(self._qam_ind.reshape(-1), torch.int32)
            return self._dev.index_select(0, (i1 * P + i2).reshape(-1)).reshape(i1.shape)
        dbl = self.precision == "double"
        rdt = torch.float64 if dbl else torch.float32
        a, b = _ffi.to_device(pam1, rdt).contiguous(), _ffi.to_device(pam2, rdt).contiguous()
        assert a.shape == b.shape and a.shape[-1] == P, "logits must have 2**(num_bits_per_symbol/2) entries"
        out = torch.empty(tuple(a.shape[:-1]) + (P * P,), dtype=rdt, device=a.device)
        _ffi.check((_ffi.lib().samd_pam2qam_logits_f64 if dbl else _ffi.lib().samd_pam2qam_logits_f32)(
            _ffi.ptr(a), _ffi.ptr(b), self._nb, a.numel() // P, _ffi.ptr(out), _ffi.stream()), "PAM2QAM")
        return out


class BinarySource(Block):
    """Random bits as float (reference mapping.py:1317-1352) on the device Philox stream."""

    def __init__(self, precision=None, seed=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._seed = seed
        self._rng = PhiloxGenerator(seed) if seed is not None else None

    def __call__(self, inputs):          # the argument is a shape, not a tensor
        return super().__call__(tuple(int(s) for s in np.asarray(inputs).reshape(-1)))

    def _convert_to_tensor(self, v):
        return v

    def call(self, inputs):
        rng = self._rng if self._rng is not None else config.rng
        out = torch.empty(tuple(inputs), dtype=torch.float32, device=_ffi.device())
        _ffi.check(_ffi.lib().samd_binary_source_f32(rng.seed, rng.next_call(), out.numel(), _ffi.ptr(out),
                                                     _ffi.stream()), "BinarySource")
        return out.to(self.rdtype)                     # bits are exact in either precision


class SymbolSource(Block):
    """``SymbolSource(constellation_type=None, num_bits_per_symbol=None, constellation=None,
    return_indices=False, return_bits=False, seed=None)(shape)``: random constellation symbols of
    the given shape (BinarySource -> Mapper, reference mapping.py:1354-1450); optionally also the
    symbol indices and/or the bits [..., num_bits_per_symbol]."""

    def __init__(self, constellation_type=None, num_bits_per_symbol=None, constellation=None, return_indices=False,
                 return_bits=False, seed=None, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        self._num_bits_per_symbol = constellation.num_bits_per_symbol
        self._return_indices, self._return_bits = return_indices, return_bits
        self._binary_source = BinarySource(seed=seed, precision=precision)
        self._mapper = Mapper(constellation=constellation, return_indices=return_indices, precision=precision)

    def __call__(self, inputs):          # the argument is a shape, not a tensor
        return super().__call__(tuple(int(s) for s in np.asarray(inputs).reshape(-1)))

    def _convert_to_tensor(self, v):
        return v

    def call(self, inputs):
        b = self._binary_source(tuple(inputs) + (self._num_bits_per_symbol,))
        if self._return_indices:
            x, ind = self._mapper(b)
        else:
            x = self._mapper(b)
        result = x.squeeze(-1)
        if self._return_indices or self._return_bits:
            result = [result]
        if self._return_indices:
            result.append(ind.squeeze(-1))
        if self._return_bits:
            result.append(b)
        return tuple(result) if isinstance(result, list) else result


class QAMSource(SymbolSource):
    """``QAMSource(num_bits_per_symbol, return_indices=False, return_bits=False, seed=None)`` (:1452-1514)."""

    def __init__(self, num_bits_per_symbol=None, return_indices=False, return_bits=False, seed=None, precision=None,
                 **kwargs):
        super().__init__(constellation_type="qam", num_bits_per_symbol=num_bits_per_symbol, return_indices=return_indices,
                         return_bits=return_bits, seed=seed, precision=precision, **kwargs)


class PAMSource(SymbolSource):
    """``PAMSource(num_bits_per_symbol, return_indices=False, return_bits=False, seed=None)`` (:1516-1576)."""

    def __init__(self, num_bits_per_symbol=None, return_indices=False, return_bits=False, seed=None, precision=None,
                 **kwargs):
        super().__init__(constellation_type="pam", num_bits_per_symbol=num_bits_per_symbol, return_indices=return_indices,
                         return_bits=return_bits, seed=seed, precision=precision, **kwargs)
