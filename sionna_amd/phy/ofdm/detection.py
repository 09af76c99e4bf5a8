"""``ofdm.LinearDetector`` - mirror of reference src/sionna/phy/ofdm/detection.py:740-847
(-> ``OFDMDetector.call`` :289-317 -> ``mimo.LinearDetector``): fused LMMSE equaliser followed by
the LLR demapper with the per-symbol effective noise variance - and ``ofdm.MMSEPICDetector``
(:1062-1230 on ``OFDMDetectorWithPrior`` :320-560, bit output) in one fused launch."""
import torch

from ... import _ffi
from ..block import Block, wrap
from ..mapping import Demapper, SymbolDemapper, Constellation
from .equalization import OFDMEqualizer


class LinearDetector(Block):
    def __init__(self, equalizer, output, demapping_method, resource_grid, stream_management,
                 constellation_type=None, num_bits_per_symbol=None, constellation=None, hard_out=False,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if equalizer not in ("lmmse", "zf", "mf"):
            raise NotImplementedError(f"LinearDetector: equalizer '{equalizer}' has no HIP path (lmmse / zf / mf)")
        assert output in ("bit", "symbol"), "Unknown output"
        self._eq = OFDMEqualizer(equalizer, resource_grid, stream_management, precision=precision)
        self._constellation = Constellation.check_or_create(
            constellation_type=constellation_type, num_bits_per_symbol=num_bits_per_symbol,
            constellation=constellation, precision=precision)
        if output == "bit":
            self._demapper = Demapper(demapping_method, constellation=self._constellation, hard_out=hard_out,
                                      precision=precision)
        else:       # [batch, num_tx, num_streams, num_data_symbols, num_points] logits or [..., num_data_symbols] indices
            self._demapper = SymbolDemapper(constellation=self._constellation, hard_out=hard_out, precision=precision)

    def call(self, y, h_hat, err_var, no):
        dm = self._demapper
        if isinstance(dm, Demapper) and dm.precision == "single" and dm._separable:
            # a square QAM and a channel estimate still deferred by LSChannelEstimator("nn"): estimate, equalise and demap
            # in one launch (equalization.py _fused_lsnn); the LLRs are those of the three separate blocks, bit for bit
            lev = self._constellation.pam_levels(raw=True)
            if lev is not None:
                llr = self._eq._fused_lsnn(y, h_hat, err_var, no, demap=(self._constellation.num_bits_per_symbol,
                                                                        dm._method == "maxlog", bool(dm._hard_out), lev))
                if llr is not None:
                    return llr
        x_hat, no_eff = self._eq(y, h_hat, err_var, no)
        return self._demapper(x_hat, no_eff)          # bit: [batch, num_tx, num_streams, num_data_symbols*m]


class MMSEPICDetector(Block):
    """``MMSEPICDetector(output, demapping_method, resource_grid, stream_management, num_iter=1,
    constellation_type=None, num_bits_per_symbol=None, constellation=None, hard_out=False)``
    ``(y, h_hat, prior, err_var, no)``; prior / output [batch, num_tx, num_streams,
    num_data_symbols * num_bits_per_symbol] (LLRs; output = extrinsic LLRs)."""

    def __init__(self, output, demapping_method, resource_grid, stream_management, num_iter=1,
                 constellation_type=None, num_bits_per_symbol=None, constellation=None, hard_out=False,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        from ..mimo.detection import MMSEPICDetector as _MimoPIC
        self._det = _MimoPIC(output, demapping_method, num_iter, constellation_type, num_bits_per_symbol, constellation,
                             hard_out, precision=precision)
        self._pre = OFDMEqualizer("lmmse", resource_grid, stream_management, precision=precision)
        self._rg = resource_grid

    def call(self, y, h_hat, prior, err_var, no):
        self._require_single()
        rg = self._rg
        det = self._det
        pts, nb, maxlog, num_iter, hard = det._kernel_params()
        prior = _ffi.to_device(prior, torch.float32)
        keep, head, tabs, dims = self._pre._prepare(y, h_hat, err_var, no)
        b, nd = dims[0], rg.num_data_symbols
        lead = (b, rg.num_tx, rg.num_streams_per_tx)
        shape = lead + (nd * nb,)
        if det._output == "symbol":
            # logits on the points in, logits / indices out (ofdm/detection.py:531-560 around mimo/detection.py:1523-1524,
            # 1636-1637); the fused kernel works on the bit LLRs in between
            assert tuple(prior.shape) == lead + (nd, 1 << nb), "prior must have shape [batch, num_tx, num_streams, num_data_symbols, num_points]"
            prior = det._symbol_logits_2_llrs(prior).as_subclass(torch.Tensor).reshape(shape)
        assert tuple(prior.shape) == shape, "prior must have shape [batch, num_tx, num_streams, num_data_symbols*num_bits_per_symbol]"
        prior = prior.contiguous()
        out = torch.zeros(shape, dtype=torch.float32, device=prior.device)
        _ffi.check(_ffi.lib().samd_ofdm_mmse_pic_f32(*head, _ffi.ptr(prior), _ffi.ptr(pts), *tabs, *dims, nb, maxlog, num_iter,
                                                     hard, _ffi.ptr(out), _ffi.stream()), "ofdm.MMSEPICDetector")
        if det._output == "symbol":
            return wrap(det._llr_2_symbol_logits_output(out.reshape(lead + (nd, nb))))
        return wrap(out)


class EPDetector(Block):
    """``EPDetector(output, resource_grid, stream_management, num_bits_per_symbol, hard_out=False, l=10,
    beta=0.9)(y, h_hat, err_var, no)`` -> LLRs [batch, num_tx, num_streams, num_data_symbols *
    num_bits_per_symbol] (ofdm/detection.py EPDetector on OFDMDetector :198-317), one fused launch."""

    def __init__(self, output, resource_grid, stream_management, num_bits_per_symbol, hard_out=False, l=10, beta=0.9,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        from ..mimo.detection import EPDetector as _MimoEP
        self._det = _MimoEP(output, num_bits_per_symbol, hard_out, l, beta, precision=precision)
        self._pre = OFDMEqualizer("lmmse", resource_grid, stream_management, precision=precision)
        self._rg = resource_grid

    def call(self, y, h_hat, err_var, no):
        rg = self._rg
        if self.precision == "double":          # the reference's decomposition around the float64 kernel (ofdm/detection.py:229-317)
            y_dt, hd, s, extract, _ = self._pre._double_inputs(y, h_hat, err_var, no)
            out = extract(self._det._solve(y_dt, hd, s))                       # [batch, num_tx, num_streams, num_data, W]
            if self._det._output == "symbol":
                return wrap(self._det._finish(out, tuple(out.shape[:4])))
            return wrap(out.reshape(tuple(out.shape[:3]) + (-1,)))
        pam, nb, l, beta, es, prec, hard = self._det._kernel_params()
        keep, head, tabs, dims = self._pre._prepare(y, h_hat, err_var, no)
        lead = (dims[0], rg.num_tx, rg.num_streams_per_tx)
        out = torch.zeros(lead + (rg.num_data_symbols * self._det._out_width(),), dtype=torch.float32, device=keep[0].device)
        _ffi.check(_ffi.lib().samd_ofdm_ep_f32(*head, _ffi.ptr(pam), *tabs, *dims, nb, l, beta, es, prec, hard, _ffi.ptr(out),
                                               _ffi.stream()), "ofdm.EPDetector")
        if self._det._output == "symbol":       # [batch, num_tx, num_streams, num_data_symbols(, num_points)] (:289-317)
            return wrap(self._det._finish(out, lead + (rg.num_data_symbols,)))
        return wrap(out)


class KBestDetector(Block):
    """``KBestDetector(output, num_streams, k, resource_grid, stream_management, constellation_type=None,
    num_bits_per_symbol=None, constellation=None, hard_out=False, use_real_rep=False, list2llr=None)``
    ``(y, h_hat, err_var, no)`` -> [batch, num_tx, num_streams, num_data_symbols * num_bits_per_symbol]."""

    def __init__(self, output, num_streams, k, resource_grid, stream_management, constellation_type=None,
                 num_bits_per_symbol=None, constellation=None, hard_out=False, use_real_rep=False, list2llr=None,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        from ..mimo.detection import KBestDetector as _MimoKBest
        self._det = _MimoKBest(output, num_streams, k, constellation_type, num_bits_per_symbol, constellation, hard_out,
                               use_real_rep, None if list2llr in (None, "default") else list2llr, precision=precision)
        self._pre = OFDMEqualizer("lmmse", resource_grid, stream_management, precision=precision)
        self._rg = resource_grid

    def call(self, y, h_hat, err_var, no):
        self._require_single()
        rg = self._rg
        pts, nb, kk, clip, hard = self._det._kernel_params()
        keep, head, tabs, dims = self._pre._prepare(y, h_hat, err_var, no)
        out = torch.zeros((dims[0], rg.num_tx, rg.num_streams_per_tx, rg.num_data_symbols * nb), dtype=torch.float32,
                          device=keep[0].device)
        fn = _ffi.lib().samd_ofdm_kbest_real_f32 if self._det._use_real_rep else _ffi.lib().samd_ofdm_kbest_f32
        _ffi.check(fn(*head, _ffi.ptr(pts), *tabs, *dims, nb, kk, clip, hard, _ffi.ptr(out), _ffi.stream()), "ofdm.KBestDetector")
        if self._det._output == "symbol":       # indices of the best path's symbols [batch, num_tx, num_streams, num_data_symbols]
            return wrap(self._det._finish(out, (dims[0], rg.num_tx, rg.num_streams_per_tx, rg.num_data_symbols)))
        return wrap(out)


class MaximumLikelihoodDetector(Block):
    """``MaximumLikelihoodDetector(output, demapping_method, resource_grid, stream_management, constellation_type=None,
    num_bits_per_symbol=None, constellation=None, hard_out=False)(y, h_hat, err_var, no)`` (reference ofdm/detection.py:524-625 on
    OFDMDetector :21-317): exhaustive ML detection of every data-carrying resource element in one fused launch
    (``samd_ofdm_ml_f32``; covariance of noise + estimation error + undesired streams as for the other OFDM detectors), then
    ``SymbolLogits2LLRs`` for ``output="bit"``.  -> [batch, num_tx, num_streams, num_data_symbols * num_bits_per_symbol] (LLRs /
    bits), or [batch, num_tx, num_streams, num_data_symbols, num_points] logits / [..., num_data_symbols] int32 indices."""

    _with_prior = False

    def __init__(self, output, demapping_method, resource_grid, stream_management, constellation_type=None,
                 num_bits_per_symbol=None, constellation=None, hard_out=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        from ..mimo.detection import MaximumLikelihoodDetector as _MimoML
        self._det = _MimoML(output, demapping_method, stream_management.num_streams_per_rx, constellation_type, num_bits_per_symbol,
                            constellation, hard_out, precision=precision)
        self._pre = OFDMEqualizer("lmmse", resource_grid, stream_management, precision=precision)
        self._rg = resource_grid

    def _run_double(self, y, h_hat, prior, err_var, no):
        """precision="double": the reference's own decomposition (ofdm/detection.py:229-317, 454-510) - per-RE inputs built on the
        device, the float64 detector kernel on every resource element, the data symbols of the streams gathered"""
        rg, det = self._rg, self._det
        nb = det._constellation.num_bits_per_symbol
        y_dt, hd, s, extract, scatter = self._pre._double_inputs(y, h_hat, err_var, no)
        lead = (y_dt.shape[0], rg.num_tx, rg.num_streams_per_tx, rg.num_data_symbols)
        pr = None
        if prior is not None:
            pr = _ffi.to_device(prior, torch.float64)
            if det._output == "bit":
                assert tuple(pr.shape) == lead[:3] + (lead[3] * nb,), "prior must have shape [batch, num_tx, num_streams, num_data_symbols*num_bits_per_symbol]"
                pr = det._llrs2logits(pr.reshape(lead + (nb,))).as_subclass(torch.Tensor)
            assert tuple(pr.shape) == lead + (1 << nb,), "prior must have shape [batch, num_tx, num_streams, num_data_symbols, num_points]"
            pr = scatter(pr)                                                       # [B,rx,T,F,K,num_points], zeros off the data
        out = det._finish(extract(det._logits(y_dt, hd, s, pr)))
        if det._output == "bit":
            out = out.as_subclass(torch.Tensor).reshape(lead[:3] + (lead[3] * nb,))
        return wrap(out)

    def _run(self, y, h_hat, prior, err_var, no):
        if self.precision == "double":
            return self._run_double(y, h_hat, prior, err_var, no)
        rg, det = self._rg, self._det
        pts, nb, maxlog = det._kernel_params()
        npts = 1 << nb
        keep, head, tabs, dims = self._pre._prepare(y, h_hat, err_var, no)
        lead = (dims[0], rg.num_tx, rg.num_streams_per_tx, rg.num_data_symbols)
        pr = None
        if prior is not None:
            pr = _ffi.to_device(prior, torch.float32)
            if det._output == "bit":            # [batch, num_tx, num_streams, num_data_symbols * nb] LLRs -> logits on the points
                assert tuple(pr.shape) == lead[:3] + (lead[3] * nb,), "prior must have shape [batch, num_tx, num_streams, num_data_symbols*num_bits_per_symbol]"
                pr = det._llrs2logits(pr.reshape(lead + (nb,))).as_subclass(torch.Tensor)
            assert tuple(pr.shape) == lead + (npts,), "prior must have shape [batch, num_tx, num_streams, num_data_symbols, num_points]"
            pr = pr.contiguous()
        logits = torch.zeros(lead + (npts,), dtype=torch.float32, device=keep[0].device)
        _ffi.check(_ffi.lib().samd_ofdm_ml_f32(*head, _ffi.ptr(pr) if pr is not None else None, _ffi.ptr(pts), *tabs, *dims, nb, maxlog,
                                               _ffi.ptr(logits), _ffi.stream()), "ofdm.MaximumLikelihoodDetector")
        out = det._finish(logits)
        if det._output == "bit":                # [batch, num_tx, num_streams, num_data_symbols * nb] (ofdm/detection.py:289-317)
            out = out.as_subclass(torch.Tensor).reshape(lead[:3] + (lead[3] * nb,))
        return wrap(out)

    def call(self, y, h_hat, err_var, no):
        return self._run(y, h_hat, None, err_var, no)


class MaximumLikelihoodDetectorWithPrior(MaximumLikelihoodDetector):
    """``MaximumLikelihoodDetectorWithPrior(...)(y, h_hat, prior, err_var, no)`` (reference ofdm/detection.py:627-738 on
    OFDMDetectorWithPrior :320-510): ``prior`` = LLRs [batch, num_tx, num_streams, num_data_symbols * num_bits_per_symbol]
    (``output="bit"``) or logits [batch, num_tx, num_streams, num_data_symbols, num_points] (``"symbol"``)."""

    _with_prior = True

    def call(self, y, h_hat, prior, err_var, no):  # pylint: disable=arguments-differ
        return self._run(y, h_hat, prior, err_var, no)
