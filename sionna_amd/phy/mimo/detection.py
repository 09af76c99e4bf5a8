"""``mimo.LinearDetector`` - equaliser followed by a demapper (reference src/sionna/phy/mimo/detection.py:24-143) -,
``mimo.MMSEPICDetector`` (:1314-1643) on ``samd_mmse_pic_f32``, ``mimo.EPDetector`` (:1039-1312) on ``samd_ep_f32`` and
``mimo.KBestDetector`` (:539-1037) on ``samd_kbest_f32``; bit and symbol output (logits / indices of the constellation
points: the kernels' bit or PAM-logit outputs through ``SymbolLogits2LLRs`` / ``LLRs2SymbolLogits`` / ``PAM2QAM``)."""
import numpy as np
import torch

from ... import _ffi
from ..block import Block, wrap
from ..mapping import Demapper, SymbolDemapper, Constellation, SymbolLogits2LLRs, LLRs2SymbolLogits, PAM2QAM
from .equalization import lmmse_equalizer, zf_equalizer, mf_equalizer


class LinearDetector(Block):
    def __init__(self, equalizer, output, demapping_method, constellation_type=None, num_bits_per_symbol=None,
                 constellation=None, hard_out=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert output in ("bit", "symbol"), "Unknown output"
        assert demapping_method in ("app", "maxlog"), "Unknown demapping method"
        self._output = output
        if isinstance(equalizer, str):                        # detection.py:101-111
            assert equalizer in ("lmmse", "zf", "mf"), "Unknown equalizer."
            self._equalizer = {"lmmse": lmmse_equalizer, "zf": zf_equalizer, "mf": mf_equalizer}[equalizer]
        else:
            self._equalizer = equalizer
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        if output == "bit":
            self._demapper = Demapper(demapping_method, constellation=self._constellation, hard_out=hard_out,
                                      precision=precision)
        else:                                                 # logits / indices of the constellation points (:127-131)
            self._demapper = SymbolDemapper(constellation=self._constellation, hard_out=hard_out, precision=precision)

    def call(self, y, h, s):
        x_hat, no_eff = self._equalizer(y, h, s, precision=self.precision)   # detection.py:134
        z = self._demapper(x_hat, no_eff)                     # bit: [..., K*m]; symbol: [..., K, num_points] or [..., K]
        if self._output == "symbol":
            return z
        m = self._constellation.num_bits_per_symbol
        return z.reshape(tuple(x_hat.shape) + (m,))           # [..., K, m] (detection.py:139-143)


class MMSEPICDetector(Block):
    """``MMSEPICDetector(output, demapping_method="maxlog", num_iter=1, constellation_type=None,
    num_bits_per_symbol=None, constellation=None, hard_out=False)(y, h, s, prior)``:
    y [...,M], h [...,M,K], s [...,M,M], prior [...,K,num_bits_per_symbol] (LLRs) ->
    extrinsic LLRs [...,K,num_bits_per_symbol]."""

    def __init__(self, output, demapping_method="maxlog", num_iter=1, constellation_type=None,
                 num_bits_per_symbol=None, constellation=None, hard_out=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert isinstance(num_iter, int), "num_iter must be an integer"
        assert output in ("bit", "symbol"), "Unknown output"
        assert demapping_method in ("app", "maxlog"), "Unknown demapping method"
        self._num_iter, self._output, self._demapping_method, self._hard_out = num_iter, output, demapping_method, hard_out
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        if output == "symbol":
            # priors arrive as logits on the points and leave as logits / indices (detection.py:1478-1486, 1523-1524, 1636-1637):
            # the kernel works on bit LLRs in between
            nb = self._constellation.num_bits_per_symbol
            self._symbol_logits_2_llrs = SymbolLogits2LLRs(demapping_method, nb, precision=precision)
            self._llr_2_symbol_logits_output = LLRs2SymbolLogits(nb, hard_out=hard_out, precision=precision)

    constellation = property(lambda self: self._constellation)

    def _kernel_params(self):
        pts = _ffi.to_device(np.asarray(self._constellation.points, np.complex64), torch.complex64)
        hard = int(bool(self._hard_out)) if self._output == "bit" else 0     # symbol output: soft extrinsic LLRs from the kernel
        return pts, self._constellation.num_bits_per_symbol, int(self._demapping_method == "maxlog"), self._num_iter, hard

    def call(self, y, h, s, prior):
        self._require_single()
        y = _ffi.to_device(y, torch.complex64)
        h = _ffi.to_device(h, torch.complex64)
        s = _ffi.to_device(s, torch.complex64)
        prior = _ffi.to_device(prior, torch.float32)
        m, k = h.shape[-2], h.shape[-1]
        lead = tuple(h.shape[:-2])
        pts, nb, maxlog, num_iter, hard = self._kernel_params()
        if self._output == "symbol":
            assert tuple(prior.shape[-2:]) == (k, 1 << nb), "prior must have shape [..., num_streams, num_points]"
            prior = self._symbol_logits_2_llrs(prior).as_subclass(torch.Tensor)
        assert tuple(prior.shape[-2:]) == (k, nb), "prior must have shape [..., num_streams, num_bits_per_symbol]"
        y = torch.broadcast_to(y, lead + (m,)).contiguous()
        s = torch.broadcast_to(s, lead + (m, m)).contiguous()
        prior = torch.broadcast_to(prior, lead + (k, nb)).contiguous()
        out = torch.empty(lead + (k, nb), dtype=torch.float32, device=y.device)
        h = h.contiguous()                      # (a named tensor: its storage must outlive the launch call)
        _ffi.check(_ffi.lib().samd_mmse_pic_f32(_ffi.ptr(y), _ffi.ptr(h), _ffi.ptr(s), _ffi.ptr(prior),
                                                _ffi.ptr(pts), y.numel() // m, m, k, nb, maxlog, num_iter, hard,
                                                _ffi.ptr(out), _ffi.stream()), "MMSEPICDetector")
        if self._output == "symbol":
            return wrap(self._llr_2_symbol_logits_output(out))
        return wrap(out)


def _pam_points_over_sqrt2(nbh, precision=None):
    """Constellation("pam", nbh, precision=precision)() / sqrt(2): the real dimensions of the unit-energy QAM constellation,
    formed in the block's precision (detection.py:1155-1161)."""
    from ..mapping import pam
    return np.asarray(pam(nbh, precision=precision), np.complex128).real / np.sqrt(2.0)


class EPDetector(Block):
    """``EPDetector(output, num_bits_per_symbol, hard_out=False, l=10, beta=0.9)(y, h, s)``: expectation
    propagation MIMO detector of [EP2014] for QAM (mimo/detection.py:1039-1312), bit output:
    y [...,M], h [...,M,K], s [...,M,M] -> LLRs [...,K,num_bits_per_symbol]."""

    def __init__(self, output, num_bits_per_symbol, hard_out=False, l=10, beta=0.9, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert output in ("bit", "symbol"), "Unknown output"
        assert l >= 1, "l must be a positive integer"
        assert 0.0 <= beta <= 1.0, "beta must be in [0,1]"
        assert num_bits_per_symbol % 2 == 0, "EPDetector works on QAM constellations"
        self._output, self._hard_out, self._l, self._beta = output, hard_out, int(l), float(beta)
        self._num_bits_per_symbol = int(num_bits_per_symbol)
        self._points = _pam_points_over_sqrt2(self._num_bits_per_symbol // 2, self.precision)
        self._es = float(np.var(self._points))
        self._prec = 1e-6 if self.precision == "single" else 1e-12           # detection.py:1129-1132
        if output == "symbol":
            self._pam2qam = PAM2QAM(self._num_bits_per_symbol, hard_out, precision=precision)

    def _kernel_params(self):
        """... and the output mode of ``samd_ep_f32``: 0 LLRs / 1 bits (output="bit"), 2 the logits of the two PAM
        constellations / 3 the QAM index of their argmax decisions (output="symbol")."""
        pam = _ffi.to_device(self._points.astype(self._np_rdtype), self.rdtype)
        mode = int(bool(self._hard_out)) + (2 if self._output == "symbol" else 0)
        return pam, self._num_bits_per_symbol, self._l, self._beta, self._es, self._prec, mode

    def _out_width(self):
        nb = self._num_bits_per_symbol
        return nb if self._output == "bit" else (1 if self._hard_out else 2 << (nb // 2))

    def _finish(self, out, lead_k):
        """kernel output [*lead_k, W] -> what the block returns (detection.py:1272-1312)"""
        if self._output == "bit":
            return out.reshape(lead_k + (self._num_bits_per_symbol,))
        if self._hard_out:
            return out.reshape(lead_k).to(torch.int32)
        P = 1 << (self._num_bits_per_symbol // 2)
        z = out.reshape(lead_k + (2, P))
        return self._pam2qam(z[..., 0, :].contiguous(), z[..., 1, :].contiguous())

    # The steps of one EP iteration as the reference exposes them (mimo/detection.py:1166-1227), for code that drives the
    # iteration itself (research variants of the detector): element-wise / small-matrix tensor algebra on whatever device the
    # operands live on.  ``call`` does NOT go through them - the whole detector is one kernel (csrc/mimo.hip ep_solve).
    def compute_sigma_mu(self, h_t_h, h_t_y, no, lam, gam):
        """Equations (28) and (29)"""
        sigma = torch.linalg.inv(h_t_h + no * torch.diag_embed(lam))
        mu = torch.matmul(sigma, h_t_y + no * gam.unsqueeze(-1)).squeeze(-1)
        return torch.diagonal(sigma * no, dim1=-2, dim2=-1), mu

    def compute_v_x_obs(self, sigma, mu, lam, gam):
        """Equations (31) and (32)"""
        v_obs = torch.clamp_min(1 / (1 / sigma - lam), self._prec)
        return v_obs, v_obs * (mu / sigma - gam)

    def compute_v_x(self, v_obs, x_obs):
        """Equation (33): mean and variance of the PAM symbols under the cavity distribution, and its logits"""
        pts = torch.as_tensor(self._points, dtype=x_obs.dtype, device=x_obs.device)
        logits = -torch.pow(x_obs.unsqueeze(-1) - pts, 2) / (2. * v_obs.unsqueeze(-1))
        pmf = torch.softmax(logits, dim=-1)
        x = torch.sum(pts * pmf, dim=-1, keepdim=True)
        v = torch.clamp_min(torch.sum((pts - x) ** 2 * pmf, dim=-1), self._prec)
        return v, x.squeeze(-1), logits

    def update_lam_gam(self, v, v_obs, x, x_obs, lam, gam):
        """Equations (35) - (38): new multipliers where they stay non-negative, damped by beta"""
        lam_new, gam_new = 1 / v - 1 / v_obs, x / v - x_obs / v_obs
        keep = lam_new < 0
        lam_new, gam_new = torch.where(keep, lam, lam_new), torch.where(keep, gam, gam_new)
        return (1 - self._beta) * lam_new + self._beta * lam, (1 - self._beta) * gam_new + self._beta * gam

    def _solve(self, y, h, s):
        """contiguous device tensors y [..., M], h [..., M, K], s [..., M, M] in the block's complex dtype -> the kernel's
        output [..., K, W] (one launch of samd_ep_f32 / samd_ep_f64)"""
        m, k = h.shape[-2], h.shape[-1]
        pam, nb, l, beta, es, prec, hard = self._kernel_params()
        out = torch.empty(tuple(h.shape[:-2]) + (k, self._out_width()), dtype=self.rdtype, device=y.device)
        fn = _ffi.lib().samd_ep_f64 if self.precision == "double" else _ffi.lib().samd_ep_f32
        _ffi.check(fn(_ffi.ptr(y), _ffi.ptr(h), _ffi.ptr(s), _ffi.ptr(pam), y.numel() // m, m, k, nb, l, beta, es, prec, hard,
                      _ffi.ptr(out), _ffi.stream()), "EPDetector")
        return out

    def call(self, y, h, s):
        y = _ffi.to_device(y, self.cdtype)
        h = _ffi.to_device(h, self.cdtype)
        s = _ffi.to_device(s, self.cdtype)
        m, k = h.shape[-2], h.shape[-1]
        lead = tuple(h.shape[:-2])
        y = torch.broadcast_to(y, lead + (m,)).contiguous()
        s = torch.broadcast_to(s, lead + (m, m)).contiguous()
        h = h.contiguous()                      # (a named tensor: its storage must outlive the launch call)
        return wrap(self._finish(self._solve(y, h, s), lead + (k,)))


class KBestDetector(Block):
    """``KBestDetector(output, num_streams, k, constellation_type=None, num_bits_per_symbol=None,
    constellation=None, hard_out=False, use_real_rep=False, list2llr=None)(y, h, s)``: breadth-first
    tree search keeping the k best partial paths (mimo/detection.py:539-1037) with the default
    ``List2LLRSimple`` (mimo/utils.py:420-578); complex representation or (``use_real_rep=True``, QAM) the real-valued
    equivalent of the channel with twice the streams over the PAM levels (``samd_kbest_real_f32``):
    y [...,M], h [...,M,K], s [...,M,M] -> [...,K,num_bits_per_symbol] (LLRs clipped to +-20, or bits)."""

    def __init__(self, output, num_streams, k, constellation_type=None, num_bits_per_symbol=None, constellation=None,
                 hard_out=False, use_real_rep=False, list2llr=None, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert output in ("bit", "symbol"), "Unknown output"
        if output == "symbol":
            assert hard_out is True, "Soft-symbols are not supported for this detector."
        self._output = output
        self._use_real_rep = bool(use_real_rep)
        if list2llr is not None:
            raise NotImplementedError("KBestDetector: custom list2llr callables have no HIP path (List2LLRSimple only)")
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        self._num_streams, self._hard_out = int(num_streams), bool(hard_out)
        num_symbols = 2 ** self._constellation.num_bits_per_symbol
        depth = self._num_streams
        if self._use_real_rep:
            # the real-valued equivalent of the channel: 2 num_streams real streams over the PAM levels of one axis
            # (mimo/detection.py:705-727): QAM only
            ct = constellation_type if constellation_type is not None else getattr(constellation, "_constellation_type", None)
            assert ct == "qam", "Only QAM can be used for the real-valued representation"
            nbh = self._constellation.num_bits_per_symbol // 2
            pam = np.asarray(Constellation("pam", nbh, normalize=False, precision=precision).points)
            self._pam_points = (pam / (np.std(pam) * np.sqrt(2))).astype(np.complex64)
            num_symbols, depth = 2 ** nbh, 2 * self._num_streams
        self._k = int(min(k, num_symbols ** depth))
        if self._k < k:
            import warnings
            warnings.warn(f"KBestDetector: The provided value of k={k} is larger than the possible maximum number of "
                          f"paths. It has been set to k={self._k}.")
        if self._k > 64:
            raise NotImplementedError("KBestDetector: k <= 64 on the HIP path")
        self._llr_clip_val = 20.0

    @property
    def list2llr(self):
        """The list-to-LLR function (mimo/detection.py:757-772).  ``None`` stands for the reference's default
        ``List2LLRSimple(num_bits_per_symbol)`` with llr_clip_val 20, which is what the kernel computes."""
        return None

    @list2llr.setter
    def list2llr(self, value):
        if value is not None:
            raise NotImplementedError("KBestDetector: custom list2llr callables have no HIP path (List2LLRSimple only)")

    def _kernel_params(self):
        pts = self._pam_points if self._use_real_rep else np.asarray(self._constellation.points, np.complex64)
        pts = _ffi.to_device(pts, torch.complex64)
        return pts, self._constellation.num_bits_per_symbol, self._k, self._llr_clip_val, int(self._hard_out)

    def _finish(self, out, lead_k):
        """kernel output [*lead_k, nb] (LLRs or the bits of the best path) -> what the block returns: for
        output="symbol" the index of the best path's symbols (detection.py:1001-1019) - their bit labels read as a number"""
        nb = self._constellation.num_bits_per_symbol
        out = out.reshape(lead_k + (nb,))
        if self._output == "bit":
            return out
        weights = (1 << torch.arange(nb - 1, -1, -1, device=out.device, dtype=torch.int32))
        return (out.to(torch.int32) * weights).sum(-1, dtype=torch.int32)

    def call(self, y, h, s):
        self._require_single()
        y = _ffi.to_device(y, torch.complex64)
        h = _ffi.to_device(h, torch.complex64)
        s = _ffi.to_device(s, torch.complex64)
        m, k = h.shape[-2], h.shape[-1]
        assert m >= k, "The number of receive antennas cannot be smaller than the number of streams"
        assert k == self._num_streams, "h must have num_streams columns"
        lead = tuple(h.shape[:-2])
        pts, nb, kk, clip, hard = self._kernel_params()
        y = torch.broadcast_to(y, lead + (m,)).contiguous()
        s = torch.broadcast_to(s, lead + (m, m)).contiguous()
        out = torch.empty(lead + (k, nb), dtype=torch.float32, device=y.device)
        h = h.contiguous()                      # (a named tensor: its storage must outlive the launch call)
        fn = _ffi.lib().samd_kbest_real_f32 if self._use_real_rep else _ffi.lib().samd_kbest_f32
        _ffi.check(fn(_ffi.ptr(y), _ffi.ptr(h), _ffi.ptr(s), _ffi.ptr(pts), y.numel() // m, m, k, nb, kk, clip, hard,
                      _ffi.ptr(out), _ffi.stream()), "KBestDetector")
        return wrap(self._finish(out, lead + (k,)))


class MaximumLikelihoodDetector(Block):
    """``MaximumLikelihoodDetector(output, demapping_method, num_streams, constellation_type=None, num_bits_per_symbol=None,
    constellation=None, hard_out=False)(y, h, s, prior=None)`` - exhaustive maximum-likelihood detection (reference
    mimo/detection.py:145-537): the channel is whitened with ``s``, the exponents ``-||y~ - H~ x||^2`` (+ prior) of ALL
    ``num_points ** num_streams`` candidate vectors are reduced per stream and constellation point with logsumexp ("app") or max
    ("maxlog") - one launch of ``samd_ml_detect_f32`` - and, for ``output="bit"``, turned into LLRs / hard bits by
    :class:`~sionna_amd.phy.mapping.SymbolLogits2LLRs` like in the reference (:531-536).

    y [..., M], h [..., M, num_streams], s [..., M, M]; prior: LLRs [..., num_streams, num_bits_per_symbol] ("bit") or logits
    [..., num_streams, num_points] ("symbol") -> LLRs / bits [..., num_streams, num_bits_per_symbol], or logits
    [..., num_streams, num_points] / symbol indices [..., num_streams] int32."""

    def __init__(self, output, demapping_method, num_streams, constellation_type=None, num_bits_per_symbol=None,
                 constellation=None, hard_out=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        assert output in ("bit", "symbol"), "Unknown output"
        assert demapping_method in ("app", "maxlog"), "Unknown demapping method"
        self._output, self._demapping_method, self._hard_out = output, demapping_method, bool(hard_out)
        self._num_streams = int(num_streams)
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        nb = self._constellation.num_bits_per_symbol
        if (1 << nb) ** self._num_streams > 65536 or self._num_streams * (1 << nb) > 200:
            raise NotImplementedError("MaximumLikelihoodDetector: num_points ** num_streams <= 65536 and "
                                      "num_streams * num_points <= 200 on the HIP path")
        if output == "bit":
            from ..mapping import SymbolLogits2LLRs, LLRs2SymbolLogits
            self._logits2llr = SymbolLogits2LLRs(demapping_method, nb, hard_out=hard_out, precision=precision)
            self._llrs2logits = LLRs2SymbolLogits(nb, hard_out=False, precision=precision)

    constellation = property(lambda self: self._constellation)

    def _kernel_params(self):
        pts = _ffi.to_device(np.asarray(self._constellation.points, np.complex64), torch.complex64)
        return pts, self._constellation.num_bits_per_symbol, int(self._demapping_method == "maxlog")

    def _finish(self, logits):
        """logits [..., num_streams, num_points] -> the block's output (mimo/detection.py:529-537)"""
        if self._output == "bit":
            return self._logits2llr(logits)
        if self._hard_out:
            return torch.argmax(logits.as_subclass(torch.Tensor), dim=-1).to(torch.int32)
        return logits

    def _logits(self, y, h, s, pr):
        """device tensors y [..., M], h [..., M, K], s [..., M, M] (contiguous, the block's complex dtype), pr logits on the points
        [..., K, num_points] or None -> logits [..., K, num_points]: one launch of samd_ml_detect_f32 / _f64"""
        m, k = h.shape[-2], h.shape[-1]
        dbl = self.precision == "double"
        nb = self._constellation.num_bits_per_symbol
        pts = _ffi.to_device(np.asarray(self._constellation.points, self._np_cdtype), self.cdtype)
        logits = torch.empty(tuple(h.shape[:-2]) + (k, 1 << nb), dtype=self.rdtype, device=y.device)
        fn = _ffi.lib().samd_ml_detect_f64 if dbl else _ffi.lib().samd_ml_detect_f32
        _ffi.check(fn(_ffi.ptr(y), _ffi.ptr(h), _ffi.ptr(s), _ffi.ptr(pr) if pr is not None else None, _ffi.ptr(pts), y.numel() // m, m, k,
                      nb, int(self._demapping_method == "maxlog"), _ffi.ptr(logits), _ffi.stream()), "MaximumLikelihoodDetector")
        return logits

    def call(self, y, h, s, prior=None):
        y = _ffi.to_device(y, self.cdtype)
        h = _ffi.to_device(h, self.cdtype)
        s = _ffi.to_device(s, self.cdtype)
        m, k = h.shape[-2], h.shape[-1]
        assert k == self._num_streams, "h must have num_streams columns"
        lead = tuple(h.shape[:-2])
        npts = 1 << self._constellation.num_bits_per_symbol
        y = torch.broadcast_to(y, lead + (m,)).contiguous()
        s = torch.broadcast_to(s, lead + (m, m)).contiguous()
        h = h.contiguous()                      # (a named tensor: its storage must outlive the launch call)
        pr = None
        if prior is not None:
            pr = _ffi.to_device(prior, self.rdtype)
            if self._output == "bit":           # LLRs on the bits -> logits on the points (:475-479)
                pr = self._llrs2logits(pr).as_subclass(torch.Tensor)
            pr = torch.broadcast_to(pr, lead + (k, npts)).contiguous()
        return wrap(self._finish(self._logits(y, h, s, pr)))
