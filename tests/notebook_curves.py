"""BER/BLER curves the reference PUBLISHES (saved ``sim_ber`` tables of its tutorial notebooks), rebuilt from
``sionna_amd.phy`` blocks, and the statistics that decide whether two Monte-Carlo curves agree.

``tests/golden/notebook_ber.json`` (made by ``tools/extract_notebook_tables.py``) holds the reference's tables.
Each ``Curve`` below names one of them, carries the TRUE Eb/N0 grid of the notebook cell (the printed column is
rounded), and a ``build()`` that constructs the same link from this package the way the notebook cell does
(cell numbers and .ipynb line numbers cited per curve).  ``tests/test_gpu_ber_reference.py`` asserts agreement;
``tools/ber_vs_reference.py`` writes the overlay to ``profiles/``.

Agreement criteria (BASELINE.json: "BER curves overlapping the reference within 0.05 dB"):
  * per point: pooled two-proportion z-score of the block-error counts, |z| <= Z_POINT (4.0; with ~400 compared
    points a 3-sigma gate would fail by chance alone in about two runs out of three, 4 sigma in ~2 %), and the
    number of points beyond 3 sigma is reported;
  * per curve: sum z^2 over compared points against chi-square(npts), p >= 1e-4;
  * per curve: Eb/N0 at BLER 1e-1 and 1e-2 (log-linear interpolation between simulated points) within
    0.05 dB + 3 sigma_dB, where sigma_dB is the Monte-Carlo uncertainty of the crossing propagated from both
    curves' error counts through the local slope.  ``corr`` > 1 widens all variances for links where several
    codewords share one channel realisation (block errors inside one batch example are then not independent).
This module holds no GPU code; importing it needs only numpy."""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_JSON = os.path.join(HERE, "golden", "notebook_ber.json")

Z_POINT = 4.0
DB_TOL = 0.05
CHI2_P_MIN = 1e-4
MIN_POOLED_ERRORS = 30          # below this the normal approximation of the z-test is not used
MIN_ERRORS_EACH = 10            # ... and each side needs at least this many error events


def load_tables():
    with open(GOLDEN_JSON) as f:
        return json.load(f)["tables"]


# ----------------------------------------------------------------------------------------------------------------------
# statistics
# ----------------------------------------------------------------------------------------------------------------------
def z_score(e_ref, n_ref, e_our, n_our, corr=1.0):
    """Pooled two-proportion z (reference minus ours); None when there is too little data."""
    if n_ref == 0 or n_our == 0 or e_ref + e_our < MIN_POOLED_ERRORS or min(e_ref, e_our) < MIN_ERRORS_EACH:
        return None
    p = (e_ref + e_our) / (n_ref + n_our)
    if p >= 1.0:
        return None                  # every block in error on both sides: carries no information
    var = p * (1 - p) * (1 / n_ref + 1 / n_our) * corr
    return (e_ref / n_ref - e_our / n_our) / math.sqrt(var)


def chi2_sf(x, k):
    """Survival function of chi-square(k) (scipy when present, Wilson-Hilferty otherwise)."""
    try:
        from scipy.stats import chi2
        return float(chi2.sf(x, k))
    except Exception:  # pylint: disable=broad-except
        z = ((x / k) ** (1 / 3) - (1 - 2 / (9 * k))) / math.sqrt(2 / (9 * k))
        return 0.5 * math.erfc(z / math.sqrt(2))


def crossing(ebno, errs, blocks, level, corr=1.0):
    """Eb/N0 where the curve crosses ``level`` (first downward crossing, log10-linear interpolation) and the 1-sigma
    Monte-Carlo uncertainty of that abscissa.  Returns (None, None) when the curve does not bracket the level with
    points carrying >= 10 errors."""
    ebno = np.asarray(ebno, float)
    errs = np.asarray(errs, float)
    blocks = np.asarray(blocks, float)
    ok = (blocks > 0) & (errs >= 10)
    p = np.where(ok, errs / np.maximum(blocks, 1), np.nan)
    for i in range(len(ebno) - 1):
        if not (ok[i] and ok[i + 1]):
            continue
        if p[i] >= level > p[i + 1]:
            l0, l1 = math.log10(p[i]), math.log10(p[i + 1])
            t = (l0 - math.log10(level)) / (l0 - l1)
            x = ebno[i] + t * (ebno[i + 1] - ebno[i])
            slope = (l1 - l0) / (ebno[i + 1] - ebno[i])                  # decades per dB (negative)
            # var(log10 p) ~ (1-p)/(e ln10^2); the interpolated log-level mixes both ends with weights (1-t), t
            v0 = (1 - p[i]) / errs[i] * corr / math.log(10) ** 2
            v1 = (1 - p[i + 1]) / errs[i + 1] * corr / math.log(10) ** 2
            sig = math.sqrt((1 - t) ** 2 * v0 + t ** 2 * v1) / abs(slope)
            return x, sig
    return None, None


def compare(ref_rows, our_rows, ebno_true, corr=1.0, use_bits=False):
    """Compare two curves point by point and at the 1e-1 / 1e-2 crossings.  ``ref_rows``/``our_rows``: lists of dicts with
    block_errors/num_blocks (or bit_errors/num_bits when ``use_bits``; then only the dB offsets are meaningful, bit
    errors inside a block are correlated, and ``corr`` should carry the mean number of bit errors per block error)."""
    ek, nk = ("bit_errors", "num_bits") if use_bits else ("block_errors", "num_blocks")
    zs, pts = [], []
    for x, r, o in zip(ebno_true, ref_rows, our_rows):
        if o is None:
            continue
        z = z_score(r[ek], r[nk], o[ek], o[nk], corr)
        pts.append({"ebno_db": float(x), "ref": r[ek] / max(r[nk], 1), "ours": o[ek] / max(o[nk], 1),
                    "ref_errors": r[ek], "our_errors": o[ek], "our_n": o[nk], "z": z})
        if z is not None:
            zs.append(z)
    n = min(len(ref_rows), len(our_rows))
    xs = list(ebno_true[:n])
    out = {"points": pts, "n_z": len(zs), "max_abs_z": max((abs(z) for z in zs), default=0.0),
           "n_beyond_3sigma": sum(abs(z) > 3 for z in zs),
           "chi2": float(sum(z * z for z in zs)), "chi2_p": chi2_sf(sum(z * z for z in zs), len(zs)) if zs else 1.0,
           "crossings": {}}
    have = [o is not None for o in our_rows[:n]]
    for level in (1e-1, 1e-2, 1e-3):
        xr, sr = crossing(xs, [r[ek] for r in ref_rows[:n]], [r[nk] for r in ref_rows[:n]], level, corr)
        xo, so = crossing([x for x, h in zip(xs, have) if h], [o[ek] for o in our_rows[:n] if o is not None],
                          [o[nk] for o in our_rows[:n] if o is not None], level, corr)
        if xr is None or xo is None:
            continue
        sig = math.sqrt(sr * sr + so * so)
        out["crossings"]["%.0e" % level] = {"ref_db": xr, "our_db": xo, "delta_db": xo - xr, "sigma_db": sig,
                                            "tol_db": DB_TOL + 3 * sig, "ok": abs(xo - xr) <= DB_TOL + 3 * sig}
    out["ok_points"] = out["max_abs_z"] <= Z_POINT
    out["ok_chi2"] = out["chi2_p"] >= CHI2_P_MIN
    out["ok_crossings"] = all(c["ok"] for k, c in out["crossings"].items() if k in ("1e-01", "1e-02"))
    out["ok"] = out["ok_points"] and out["ok_chi2"] and out["ok_crossings"]
    return out


# ----------------------------------------------------------------------------------------------------------------------
# the notebooks' models, built from sionna_amd.phy
# ----------------------------------------------------------------------------------------------------------------------
def _phy():
    import sionna_amd.phy as phy
    return phy


class _AwgnFec:
    """``System_Model`` of 5G_Channel_Coding_Polar_vs_LDPC_Codes.ipynb cell 6 / Evolution_of_FEC.ipynb cell 5:
    source -> encoder -> QAM -> AWGN -> demapper -> decoder; ``encoder=None`` = uncoded with hard decisions."""

    def __init__(self, k, n, m, encoder, decoder, demapping_method="app", sim_esno=False):
        phy = _phy()
        self.k, self.n, self.m, self.sim_esno = k, n, m, sim_esno
        self.source = phy.mapping.BinarySource()
        self.constellation = phy.mapping.Constellation("qam", num_bits_per_symbol=m)
        self.mapper = phy.mapping.Mapper(constellation=self.constellation)
        self.demapper = phy.mapping.Demapper(demapping_method, constellation=self.constellation)
        self.channel = phy.channel.AWGN()
        self.encoder, self.decoder = encoder, decoder

    def __call__(self, batch_size, ebno_db):
        phy = _phy()
        u = self.source([batch_size, self.k])
        c = u if self.encoder is None else self.encoder(u)
        if self.sim_esno:
            no = phy.utils.ebnodb2no(ebno_db, num_bits_per_symbol=1, coderate=1)
        else:
            rate = 1 if self.encoder is None else self.k / self.n
            no = phy.utils.ebnodb2no(ebno_db, num_bits_per_symbol=self.m, coderate=rate)
        llr = self.demapper(self.channel(self.mapper(c), no), no)
        u_hat = phy.utils.hard_decisions(llr) if self.decoder is None else self.decoder(llr)
        return u, u_hat


def _ldpc(k, n, num_iter=20, m=2, **dec_kw):
    def build():
        phy = _phy()
        enc = phy.fec.ldpc.LDPC5GEncoder(k=k, n=n)
        return _AwgnFec(k, n, m, enc, phy.fec.ldpc.LDPC5GDecoder(enc, num_iter=num_iter, **dec_kw))
    return build


def _polar5g(k, n, dec_type, list_size=8, m=2):
    def build():
        phy = _phy()
        enc = phy.fec.polar.Polar5GEncoder(k=k, n=n)
        return _AwgnFec(k, n, m, enc, phy.fec.polar.Polar5GDecoder(enc, dec_type=dec_type, list_size=list_size))
    return build


def _rm_scl(r, mm, list_size=8):
    def build():
        phy = _phy()
        from sionna_amd.phy.fec.polar.utils import generate_rm_code
        f, _, n, k, _ = generate_rm_code(r, mm)
        return _AwgnFec(k, n, 2, phy.fec.polar.PolarEncoder(f, n), phy.fec.polar.PolarSCLDecoder(f, n, list_size=list_size))
    return build


def _uncoded(k, m=2):
    return lambda: _AwgnFec(k, k, m, None, None)


class _DiscoverE2E:
    """``e2e_model`` of Discover_Sionna.ipynb cell 31 with ``sys_params`` of cell 33: 1x1 OFDM (fft 256, 14 symbols,
    30 kHz, CP 16, Kronecker pilots at symbols 2 and 11), 16-QAM, 5G LDPC rate 1/2 BP-20 'boxplus', RowColumn
    interleaver, TDL-A 100 ns at 3.5 GHz / 3 m/s, normalised channel, LS + nearest-neighbour, LMMSE equaliser."""

    def __init__(self):
        phy = _phy()
        self.rg = phy.ofdm.ResourceGrid(num_ofdm_symbols=14, fft_size=256, subcarrier_spacing=30e3, num_tx=1,
                                        num_streams_per_tx=1, cyclic_prefix_length=16, pilot_pattern="kronecker",
                                        pilot_ofdm_symbol_indices=[2, 11])
        self.sm = phy.mimo.StreamManagement(rx_tx_association=np.array([[1]]), num_streams_per_tx=1)
        self.coderate, self.m = 0.5, 4
        self.n = int(self.rg.num_data_symbols * self.m)
        self.k = int(self.n * self.coderate)
        self.binary_source = phy.mapping.BinarySource()
        self.encoder = phy.fec.ldpc.LDPC5GEncoder(self.k, self.n)
        self.interleaver = phy.fec.interleaving.RowColumnInterleaver(row_depth=self.m)
        self.deinterleaver = phy.fec.interleaving.Deinterleaver(self.interleaver)
        self.mapper = phy.mapping.Mapper("qam", self.m)
        self.rg_mapper = phy.ofdm.ResourceGridMapper(self.rg)
        self.tdl = phy.channel.tr38901.TDL(model="A", delay_spread=100e-9, carrier_frequency=3.5e9, min_speed=3, max_speed=3)
        self.channel = phy.channel.OFDMChannel(self.tdl, self.rg, add_awgn=True, normalize_channel=True)
        self.ls_est = phy.ofdm.LSChannelEstimator(self.rg, interpolation_type="nn")
        self.lmmse_equ = phy.ofdm.LMMSEEqualizer(self.rg, self.sm)
        self.demapper = phy.mapping.Demapper("app", "qam", self.m)
        self.decoder = phy.fec.ldpc.LDPC5GDecoder(self.encoder, hard_out=True, cn_update="boxplus", num_iter=20)

    def __call__(self, batch_size, ebno_db):
        phy = _phy()
        b = self.binary_source([batch_size, 1, 1, self.k])
        x_rg = self.rg_mapper(self.mapper(self.interleaver(self.encoder(b))))
        no = phy.utils.ebnodb2no(ebno_db, self.m, self.coderate, self.rg)
        y = self.channel(x_rg, no)
        h_hat, err_var = self.ls_est(y, no)
        x_hat, no_eff = self.lmmse_equ(y, h_hat, err_var, no)
        b_hat = self.decoder(self.deinterleaver(self.demapper(x_hat, no_eff)))
        return b, b_hat


class _SimpleMimo:
    """``Model`` of Simple_MIMO_Simulation.ipynb cell 40 (uncorrelated): 4 tx x 16 rx i.i.d. Rayleigh flat fading with a
    fresh channel per symbol vector, 16-QAM, 5G LDPC (512, 1024) default decoder, ``lmmse_equalizer`` with S = no I."""

    def __init__(self, kronecker=False):
        phy = _phy()
        self.n, self.k, self.m, self.ntx, self.nrx = 1024, 512, 4, 4, 16
        self.binary_source = phy.mapping.BinarySource()
        self.encoder = phy.fec.ldpc.LDPC5GEncoder(self.k, self.n)
        self.mapper = phy.mapping.Mapper("qam", self.m)
        self.demapper = phy.mapping.Demapper("app", "qam", self.m)
        self.decoder = phy.fec.ldpc.LDPC5GDecoder(self.encoder, hard_out=True)
        # cell 44: KroneckerModel(exp_corr_mat(0.4, num_tx_ant), exp_corr_mat(0.7, num_rx_ant))
        corr = (phy.channel.KroneckerModel(phy.channel.exp_corr_mat(0.4, self.ntx), phy.channel.exp_corr_mat(0.7, self.nrx))
                if kronecker else None)
        self.channel = phy.channel.FlatFadingChannel(self.ntx, self.nrx, spatial_corr=corr, add_awgn=True, return_channel=True)

    def __call__(self, batch_size, ebno_db):
        import torch
        phy = _phy()
        b = self.binary_source([batch_size, self.ntx, self.k])
        x = self.mapper(self.encoder(b))
        shape = x.shape
        x = x.reshape(-1, self.ntx)
        no = float(phy.utils.ebnodb2no(ebno_db, self.m, self.k / self.n)) * math.sqrt(self.nrx)
        y, h = self.channel(x, no)
        s = (no * torch.eye(self.nrx, device=y.device)).to(torch.complex64)
        x_hat, no_eff = phy.mimo.lmmse_equalizer(y, h, s)
        llr = self.demapper(x_hat.reshape(shape), no_eff.reshape(shape))
        return b, self.decoder(llr)


class _BicmLdpc:
    """``LDPC_QAM_AWGN`` of Bit_Interleaved_Coded_Modulation.ipynb cell 23 (k=600, n=1200, default 20 iterations)."""

    def __init__(self, m, demapping_method="app", cn_update="boxplus", use_allzero=False, use_scrambler=False,
                 use_ldpc_output_interleaver=False, no_est_mismatch=1.0, k=600, n=1200, random_interleaver=False):
        phy = _phy()
        self.k, self.n, self.m = k, n, m
        # cell 17 ("Baseline (with encoder)"): RandomInterleaver / Deinterleaver around mapper ... demapper
        self.interleaver = phy.fec.interleaving.RandomInterleaver() if random_interleaver else None
        self.deinterleaver = phy.fec.interleaving.Deinterleaver(self.interleaver) if random_interleaver else None
        self.use_allzero, self.use_scrambler, self.mismatch = use_allzero, use_scrambler, no_est_mismatch
        self.source = phy.mapping.BinarySource()
        self.constellation = phy.mapping.Constellation("qam", num_bits_per_symbol=m)
        self.mapper = phy.mapping.Mapper(constellation=self.constellation)
        self.demapper = phy.mapping.Demapper(demapping_method, constellation=self.constellation)
        self.channel = phy.channel.AWGN()
        self.encoder = (phy.fec.ldpc.LDPC5GEncoder(k, n, m) if use_ldpc_output_interleaver
                        else phy.fec.ldpc.LDPC5GEncoder(k, n))
        self.decoder = phy.fec.ldpc.LDPC5GDecoder(self.encoder, cn_update=cn_update)
        self.scrambler = phy.fec.scrambling.Scrambler()
        self.descrambler = phy.fec.scrambling.Descrambler(self.scrambler, binary=False)

    def __call__(self, batch_size, ebno_db):
        import torch
        from sionna_amd import _ffi
        phy = _phy()
        no = phy.utils.ebnodb2no(ebno_db, num_bits_per_symbol=self.m, coderate=self.k / self.n)
        if self.use_allzero:
            u = torch.zeros([batch_size, self.k], device=_ffi.device())
            c = torch.zeros([batch_size, self.n], device=_ffi.device())
        else:
            u = self.source([batch_size, self.k])
            c = self.encoder(u)
        if self.use_scrambler:
            c = self.scrambler(c)
        if self.interleaver is not None:
            c = self.interleaver(c)
        y = self.channel(self.mapper(c), no)
        llr = self.demapper(y, no * self.mismatch)
        if self.interleaver is not None:
            llr = self.deinterleaver(llr)
        if self.use_scrambler:
            llr = self.descrambler(llr)
        return u, self.decoder(llr)


class _BicmGa:
    """``run_ber_ga`` of Bit_Interleaved_Coded_Modulation.ipynb cell 31: all-zero codeword, LLRs drawn by
    ``GaussianPriorSource`` (QPSK, k=600, n=1200), decoder of cell 17 (boxplus-phi, 20 iterations)."""

    def __init__(self, k=600, n=1200):
        phy = _phy()
        self.k, self.n = k, n
        enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
        self.decoder = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus-phi", num_iter=20)
        self.ga = phy.fec.utils.GaussianPriorSource()

    def __call__(self, batch_size, ebno_db):
        import torch
        from sionna_amd import _ffi
        phy = _phy()
        no = phy.utils.ebnodb2no(ebno_db, num_bits_per_symbol=2, coderate=self.k / self.n)
        u = torch.zeros([batch_size, self.k], device=_ffi.device())
        return u, self.decoder(self.ga([batch_size, self.n], no))


class _CdlModel:
    """``Model`` of MIMO_OFDM_Transmissions_over_CDL.ipynb cell 65 (uplink): 4-antenna UT -> 8-antenna BS over CDL, dual
    cross-polarised 38.901 arrays, fft 72 with guards [5, 6] and DC null, 14 symbols, Kronecker pilots, QPSK, 5G LDPC rate
    1/2 (default decoder), frequency- or time-domain channel, perfect CSI or LS + nearest neighbour, LMMSE equaliser.
    Four codewords (one per stream) share every channel realisation."""

    def __init__(self, domain, cdl_model, perfect_csi, speed, cyclic_prefix_length, pilot_ofdm_symbol_indices,
                 delay_spread=100e-9, subcarrier_spacing=15e3, l_min_override=None, cn_update=None):
        phy = _phy()
        t = phy.channel.tr38901
        self.domain, self.perfect_csi = domain, perfect_csi
        self.fc, self.cp = 2.6e9, cyclic_prefix_length
        self.n_ut, self.n_bs, self.m, self.coderate = 4, 8, 2, 0.5
        self.sm = phy.mimo.StreamManagement(np.array([[1]]), self.n_ut)
        self.rg = phy.ofdm.ResourceGrid(num_ofdm_symbols=14, fft_size=72, subcarrier_spacing=subcarrier_spacing, num_tx=1,
                                        num_streams_per_tx=self.n_ut, cyclic_prefix_length=cyclic_prefix_length,
                                        num_guard_carriers=[5, 6], dc_null=True, pilot_pattern="kronecker",
                                        pilot_ofdm_symbol_indices=pilot_ofdm_symbol_indices)
        self.n = int(self.rg.num_data_symbols * self.m)
        self.k = int(self.n * self.coderate)
        ut = t.AntennaArray(num_rows=1, num_cols=self.n_ut // 2, polarization="dual", polarization_type="cross",
                            antenna_pattern="38.901", carrier_frequency=self.fc)
        bs = t.AntennaArray(num_rows=1, num_cols=self.n_bs // 2, polarization="dual", polarization_type="cross",
                            antenna_pattern="38.901", carrier_frequency=self.fc)
        self.cdl = t.CDL(model=cdl_model, delay_spread=delay_spread, carrier_frequency=self.fc, ut_array=ut, bs_array=bs,
                         direction="uplink", min_speed=speed)
        self.freqs = phy.channel.subcarrier_frequencies(self.rg.fft_size, self.rg.subcarrier_spacing)
        if domain == "freq":
            self.channel_freq = phy.channel.ApplyOFDMChannel(add_awgn=True)
        else:
            self.l_min, self.l_max = phy.channel.time_lag_discrete_time_channel(self.rg.bandwidth)
            if l_min_override is not None:                             # probes of the ISI regime only (tools/probe_cp2.py)
                self.l_min = l_min_override
            self.l_tot = self.l_max - self.l_min + 1
            self.channel_time = phy.channel.ApplyTimeChannel(self.rg.num_time_samples, l_tot=self.l_tot, add_awgn=True)
            self.modulator = phy.ofdm.OFDMModulator(cyclic_prefix_length)
            self.demodulator = phy.ofdm.OFDMDemodulator(72, self.l_min, cyclic_prefix_length)
        self.source = phy.mapping.BinarySource()
        self.encoder = phy.fec.ldpc.LDPC5GEncoder(self.k, self.n)
        self.mapper = phy.mapping.Mapper("qam", self.m)
        self.rg_mapper = phy.ofdm.ResourceGridMapper(self.rg)
        self.ls_est = phy.ofdm.LSChannelEstimator(self.rg, interpolation_type="nn")
        self.lmmse = phy.ofdm.LMMSEEqualizer(self.rg, self.sm)
        self.demapper = phy.mapping.Demapper("app", "qam", self.m)
        self.decoder = (phy.fec.ldpc.LDPC5GDecoder(self.encoder, hard_out=True) if cn_update is None else
                        phy.fec.ldpc.LDPC5GDecoder(self.encoder, hard_out=True, cn_update=cn_update))   # (probes only)
        self.remove_nulled = phy.ofdm.RemoveNulledSubcarriers(self.rg)

    def __call__(self, batch_size, ebno_db):
        phy = _phy()
        rg = self.rg
        no = phy.utils.ebnodb2no(ebno_db, self.m, self.coderate, rg)
        b = self.source([batch_size, 1, self.n_ut, self.k])
        x_rg = self.rg_mapper(self.mapper(self.encoder(b)))
        if self.domain == "time":
            a, tau = self.cdl(batch_size, rg.num_time_samples + self.l_tot - 1, rg.bandwidth)
            h_time = phy.channel.cir_to_time_channel(rg.bandwidth, a, tau, l_min=self.l_min, l_max=self.l_max, normalize=True)
            a_freq = a[..., rg.cyclic_prefix_length:-1:(rg.fft_size + rg.cyclic_prefix_length)]
            a_freq = a_freq[..., :rg.num_ofdm_symbols]
            h_freq = phy.channel.cir_to_ofdm_channel(self.freqs, a_freq, tau, normalize=True)
            y = self.demodulator(self.channel_time(self.modulator(x_rg), h_time, no))
        else:
            a, tau = self.cdl(batch_size, rg.num_ofdm_symbols, 1 / rg.ofdm_symbol_duration)
            h_freq = phy.channel.cir_to_ofdm_channel(self.freqs, a, tau, normalize=True)
            y = self.channel_freq(x_rg, h_freq, no)
        if self.perfect_csi:
            h_hat, err_var = self.remove_nulled(h_freq), 0.0
        else:
            h_hat, err_var = self.ls_est(y, no)
        x_hat, no_eff = self.lmmse(y, h_hat, err_var, no)
        return b, self.decoder(self.demapper(x_hat, no_eff))


class _Part3Ofdm:
    """``OFDMSystem`` of Sionna_tutorial_part3.ipynb cell 40: single-antenna UT -> 4-antenna BS (dual cross-polarised)
    uplink over CDL-C (100 ns, 2.6 GHz, 10 m/s), fft 76, 30 kHz, CP 6, pilots on symbols 2 and 11, QPSK, 5G LDPC rate 1/2,
    ``OFDMChannel(normalize_channel=True)``, LS + nearest neighbour or perfect CSI, LMMSE equaliser."""

    def __init__(self, perfect_csi):
        phy = _phy()
        t = phy.channel.tr38901
        self.perfect_csi, self.m, self.coderate = perfect_csi, 2, 0.5
        self.sm = phy.mimo.StreamManagement(np.array([[1]]), 1)
        self.rg = phy.ofdm.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=30e3, num_tx=1, num_streams_per_tx=1,
                                        cyclic_prefix_length=6, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
        ut = t.Antenna(polarization="single", polarization_type="V", antenna_pattern="38.901", carrier_frequency=2.6e9)
        bs = t.AntennaArray(num_rows=1, num_cols=2, polarization="dual", polarization_type="cross", antenna_pattern="38.901",
                            carrier_frequency=2.6e9)
        cdl = t.CDL("C", 100e-9, 2.6e9, ut, bs, "uplink", min_speed=10.0)
        n = int(self.rg.num_data_symbols * self.m)
        self.k = int(n * self.coderate)
        self.source = phy.mapping.BinarySource()
        self.encoder = phy.fec.ldpc.LDPC5GEncoder(self.k, n)
        self.mapper = phy.mapping.Mapper("qam", self.m)
        self.rg_mapper = phy.ofdm.ResourceGridMapper(self.rg)
        self.channel = phy.channel.OFDMChannel(cdl, self.rg, add_awgn=True, normalize_channel=True, return_channel=True)
        self.ls_est = phy.ofdm.LSChannelEstimator(self.rg, interpolation_type="nn")
        self.lmmse = phy.ofdm.LMMSEEqualizer(self.rg, self.sm)
        self.demapper = phy.mapping.Demapper("app", "qam", self.m)
        self.decoder = phy.fec.ldpc.LDPC5GDecoder(self.encoder, hard_out=True)

    def __call__(self, batch_size, ebno_db):
        phy = _phy()
        no = phy.utils.ebnodb2no(ebno_db, num_bits_per_symbol=self.m, coderate=self.coderate, resource_grid=self.rg)
        bits = self.source([batch_size, 1, 1, self.k])
        y, h_freq = self.channel(self.rg_mapper(self.mapper(self.encoder(bits))), no)
        h_hat, err_var = (h_freq, 0.) if self.perfect_csi else self.ls_est(y, no)
        x_hat, no_eff = self.lmmse(y, h_hat, err_var, no)
        return bits, self.decoder(self.demapper(x_hat, no_eff))


class _IddRayleigh:
    """``NonIddModel`` / ``IddModel`` of Introduction_to_Iterative_Detection_and_Decoding.ipynb cells 11 / 13 with
    ``perfect_csi_rayleigh=True``: four single-antenna users -> 16 base-station antennas over i.i.d. Rayleigh block fading
    (perfect CSI), fft 48, 14 symbols, Kronecker pilots on symbols 2 and 11, 16-QAM, 5G LDPC rate 1/2 with the 5G output
    interleaver, min-sum decoding with 12 iterations.  ``detector``: 'lmmse' | 'k-best' (k = 64) | 'ep' (l = 10) - one
    detection pass - or 'idd2' / 'idd3': LMMSE detection, then (soft-output decoder with its state handed on ->
    MMSE-PIC detector with the decoder's LLRs as prior) once / twice, then the final hard-output decoder.
    Four codewords (users) share every channel realisation."""

    def __init__(self, detector):
        phy = _phy()
        self.m, self.n_ue, self.n_rx, self.R = 4, 4, 16, 0.5
        self.rg = phy.ofdm.ResourceGrid(num_ofdm_symbols=14, pilot_ofdm_symbol_indices=[2, 11], fft_size=48, num_tx=self.n_ue,
                                        pilot_pattern="kronecker", subcarrier_spacing=30e3)
        self.sm = phy.mimo.StreamManagement(np.ones([1, self.n_ue]), 1)
        self.N = int(48 * 12 * self.m)
        self.K = int(self.N * self.R)
        self.const = phy.mapping.Constellation("qam", num_bits_per_symbol=self.m)
        self.source = phy.mapping.BinarySource()
        self.encoder = phy.fec.ldpc.LDPC5GEncoder(self.K, self.N, num_bits_per_symbol=self.m)
        self.mapper = phy.mapping.Mapper(constellation=self.const)
        self.rg_mapper = phy.ofdm.ResourceGridMapper(self.rg)
        ch = phy.channel.RayleighBlockFading(num_rx=1, num_rx_ant=self.n_rx, num_tx=self.n_ue, num_tx_ant=1)
        self.channel = phy.channel.OFDMChannel(channel_model=ch, resource_grid=self.rg, add_awgn=True, normalize_channel=True,
                                               return_channel=True)
        self.remove_nulled = phy.ofdm.RemoveNulledSubcarriers(self.rg)
        od = phy.ofdm
        self.detector_kind = detector
        if detector == "k-best":
            self.detector = od.KBestDetector("bit", self.n_ue, 64, self.rg, self.sm, constellation_type="qam", num_bits_per_symbol=self.m, hard_out=False)
        elif detector == "ep":
            self.detector = od.EPDetector("bit", self.rg, self.sm, self.m, l=10, hard_out=False)
        else:
            self.detector = od.LinearDetector("lmmse", "bit", "maxlog", self.rg, self.sm, constellation_type="qam",
                                              num_bits_per_symbol=self.m, hard_out=False)
        ld = phy.fec.ldpc
        self.idd_iter = {"idd2": 2, "idd3": 3}.get(detector, 0)
        if self.idd_iter:
            self.siso_detector = od.MMSEPICDetector(output="bit", resource_grid=self.rg, stream_management=self.sm,
                                                    demapping_method="maxlog", constellation=self.const, num_iter=1, hard_out=False)
            self.siso_decoder = ld.LDPC5GDecoder(self.encoder, return_infobits=False, num_iter=12, return_state=True, hard_out=False,
                                                 cn_update="minsum")
            self.decoder = ld.LDPC5GDecoder(self.encoder, return_infobits=True, return_state=True, hard_out=True, num_iter=12,
                                            cn_update="minsum")
        else:
            self.decoder = ld.LDPC5GDecoder(self.encoder, return_infobits=True, hard_out=True, num_iter=12, cn_update="minsum")

    def __call__(self, batch_size, ebno_db):
        import torch
        from sionna_amd import _ffi
        phy = _phy()
        no = float(phy.utils.ebnodb2no(ebno_db, num_bits_per_symbol=self.m, coderate=self.R))
        no = torch.full([batch_size], no, dtype=torch.float32, device=_ffi.device())     # the notebook fills a [batch] vector
        b = self.source([batch_size, self.n_ue, 1, self.K])
        x_rg = self.rg_mapper(self.mapper(self.encoder(b)))
        y, h = self.channel(x_rg, no.reshape(-1, 1, 1, 1, 1))
        h_hat = self.remove_nulled(h)
        ev = torch.zeros(tuple(h_hat.shape), dtype=torch.float32, device=h_hat.device) if self.idd_iter else 0.0
        llr_ch = self.detector(y, h_hat, ev, no)
        if not self.idd_iter:
            return b, self.decoder(llr_ch)
        msg = None
        for _ in range(self.idd_iter - 1):
            llr_dec, msg = self.siso_decoder(llr_ch, msg_v2c=msg)
            llr_ch = self.siso_detector(y, h_hat, llr_dec, ev, no)
        b_hat, _ = self.decoder(llr_ch, msg_v2c=msg)
        return b, b_hat


def _cdl(**kw):
    return lambda: _CdlModel(**kw)


class _WeightedBPUntrained:
    """``WeightedBP`` / ``WeightedBP5G`` of Weighted_BP_Algorithm.ipynb cells 6 / 25 BEFORE training (all edge weights 1 =
    plain BP): ``num_iter`` calls of a ONE-iteration decoder that hands its v2c state to the next call (``return_state``,
    ``msg_v2c=``), tanh rule, all-zero codeword with ``GaussianPriorSource`` LLRs, soft outputs.  ``callback=True`` keeps
    the notebook's ``WeightedBPCallback`` registered (the decoder then runs on the device torch engine of custom.py)."""
    soft = True

    def __init__(self, pcm_id=None, k=None, n=None, num_iter=10, callback=False):
        phy = _phy()
        ld = phy.fec.ldpc
        if pcm_id is not None:
            pcm, self.k, self.n, self.rate = phy.fec.utils.load_parity_check_examples(pcm_id)
            kw = {"v2c_callbacks": [ld.WeightedBPCallback(num_edges=int(np.sum(pcm)))]} if callback else {}
            self.decoder = ld.LDPCBPDecoder(pcm, num_iter=1, return_state=True, hard_out=False, cn_update="boxplus", **kw)
            self.n_out = self.n
        else:
            enc = ld.LDPC5GEncoder(k, n)
            kw = {"v2c_callbacks": [ld.WeightedBPCallback(num_edges=int(np.sum(enc.pcm)))]} if callback else {}
            self.decoder = ld.LDPC5GDecoder(enc, num_iter=1, return_state=True, hard_out=False, prune_pcm=False, cn_update="boxplus", **kw)
            self.k, self.n, self.rate, self.n_out = k, n, k / n, k
        self.source = phy.fec.utils.GaussianPriorSource()
        self.num_iter = num_iter

    def __call__(self, batch_size, ebno_db):
        import torch
        from sionna_amd import _ffi
        phy = _phy()
        no = phy.utils.ebnodb2no(ebno_db, num_bits_per_symbol=2, coderate=self.rate)
        c = torch.zeros([batch_size, self.n_out], device=_ffi.device())
        llr = self.source([batch_size, self.n], no)
        msg = None
        for _ in range(self.num_iter):
            c_hat, msg = self.decoder(llr, msg_v2c=msg)
        return c, c_hat


class _Part1Uncoded:
    """``UncodedSystemAWGN`` of Sionna_tutorial_part1.ipynb cell 37: returns (bits, LLRs) - sim_ber(soft_estimates=True)."""
    soft = True

    def __init__(self, m=2, block_length=1024):
        phy = _phy()
        self.m, self.n = m, block_length
        self.mapper, self.demapper = phy.mapping.Mapper("qam", m), phy.mapping.Demapper("app", "qam", m)
        self.source, self.channel = phy.mapping.BinarySource(), phy.channel.AWGN()

    def __call__(self, batch_size, ebno_db):
        phy = _phy()
        no = phy.utils.ebnodb2no(ebno_db, num_bits_per_symbol=self.m, coderate=1.0)
        bits = self.source([batch_size, self.n])
        return bits, self.demapper(self.channel(self.mapper(bits), no), no)


class Curve:
    """One published table: ``key`` into notebook_ber.json, the cell's true Eb/N0 grid, and the model builder.
    ``use_bits``: the statistic is the BIT error count (uncoded links: independent bit errors, BLER is 1 everywhere);
    ``work``: relative cost of one block (coded bits x decoder iterations), used to bound the deep points."""

    def __init__(self, key, name, build, ebno, *, bits_per_block, corr=1.0, group="awgn", max_batch=None, cite="",
                 use_bits=False, work=None):
        self.key, self.name, self.build, self.ebno = key, name, build, np.asarray(ebno, float)
        self.bits_per_block, self.corr, self.group, self.max_batch, self.cite = bits_per_block, corr, group, max_batch, cite
        self.use_bits, self.work = use_bits, work or 40.0 * bits_per_block


PVL = "5G_Channel_Coding_Polar_vs_LDPC_Codes"
EVO = "Evolution_of_FEC"
BICM = "Bit_Interleaved_Coded_Modulation"

CURVES = [
    # --- 5G_Channel_Coding_Polar_vs_LDPC_Codes.ipynb cell 8/12 (k=64, n=128, QPSK, 1000 block errors per point)
    Curve(f"{PVL}/c12/t0", "5G LDPC BP-20 (64,128)", _ldpc(64, 128), np.arange(0, 5, 0.5), bits_per_block=64, cite="ipynb:417-430"),
    Curve(f"{PVL}/c12/t1", "5G Polar+CRC SC (64,128)", _polar5g(64, 128, "SC"), np.arange(0, 5, 0.5), bits_per_block=64, cite="ipynb:431-444"),
    Curve(f"{PVL}/c12/t2", "5G Polar+CRC SCL-8 (64,128)", _polar5g(64, 128, "SCL", 8), np.arange(0, 5, 0.5), bits_per_block=64, cite="ipynb:445-458"),
    Curve(f"{PVL}/c12/t3", "Reed Muller SCL-8 (64,128)", _rm_scl(3, 7), np.arange(0, 5, 0.5), bits_per_block=64, cite="ipynb:459-472"),
    # --- same notebook, cell 23/24: LDPC BP-20 at rate 1/2, n = 128 ... 16000 (500 block errors per point)
    *[Curve(f"{PVL}/c24/t{i}", f"5G LDPC BP-20 (n={n})", _ldpc(n // 2, n), np.arange(0, 5, 0.25), bits_per_block=n // 2,
            cite="ipynb:918-1100") for i, n in enumerate([128, 256, 512, 1000, 2000, 4000, 8000, 16000])],
    # --- same notebook, cell 46/48: Polar (128,256) hybrid SCL-8 and SC
    Curve(f"{PVL}/c48/t0", "5G Polar hybSCL-8 (128,256)", _polar5g(128, 256, "hybSCL", 8), np.arange(0, 5, 0.5), bits_per_block=128, cite="ipynb:1905-1920"),
    Curve(f"{PVL}/c48/t1", "5G Polar SC (128,256)", _polar5g(128, 256, "SC"), np.arange(0, 5, 0.5), bits_per_block=128, cite="ipynb:1921-1935"),
    # --- Evolution_of_FEC.ipynb cell 7/11 (k=512, n=1024) and cell 16/18 (k=2048, n=6156), 2000 block errors per point
    Curve(f"{EVO}/c11/t0", "Uncoded QPSK (512 bit blocks)", _uncoded(512), np.arange(0., 8, 0.2), bits_per_block=512, cite="cell 11", use_bits=True, work=512),
    Curve(f"{EVO}/c11/t3", "5G LDPC BP-40 (512,1024)", _ldpc(512, 1024, 40), np.arange(0., 8, 0.2), bits_per_block=512, work=1024 * 40, cite="ipynb:504-520"),
    Curve(f"{EVO}/c11/t4", "5G Polar hybSCL-32 (512,1024)", _polar5g(512, 1024, "hybSCL", 32), np.arange(0., 8, 0.2), bits_per_block=512,
          cite="ipynb:523-538", group="polar32", work=1024 * 32 * 10),
    Curve(f"{EVO}/c18/t0", "Uncoded QPSK (2048 bit blocks)", _uncoded(2048), np.arange(-1, 1.8, 0.1), bits_per_block=2048, cite="cell 18", use_bits=True, work=2048),
    Curve(f"{EVO}/c18/t2", "5G LDPC BP-40 (2048,6156)", _ldpc(2048, 6156, 40), np.arange(-1, 1.8, 0.1), bits_per_block=2048, work=6156 * 40, cite="cell 18"),
    # --- Bit_Interleaved_Coded_Modulation.ipynb (k=600, n=1200; these cells stop on 1000..2000 BIT errors, so few block errors)
    Curve(f"{BICM}/c19/t0", "BICM baseline with encoder + random interleaver, QPSK, boxplus-phi BP-20 (1000 bit errors per point)",
          lambda: _BicmLdpc(2, cn_update="boxplus-phi", random_interleaver=True), np.arange(0, 5, 0.25), bits_per_block=600, cite="cells 17/19"),
    Curve(f"{BICM}/c26/t0", "BICM all-zero QPSK, boxplus BP-20", lambda: _BicmLdpc(2, use_allzero=True), np.arange(0, 5, 0.25), bits_per_block=600, cite="cell 26"),
    Curve(f"{BICM}/c31/t0", "BICM Gaussian-approximated LLRs, boxplus-phi BP-20", _BicmGa, np.arange(0, 5, 0.25), bits_per_block=600, cite="ipynb:916-932"),
    *[Curve(f"{BICM}/c{c}/t0", name, (lambda kw=kw: _BicmLdpc(4, **kw)), np.arange(*grid), bits_per_block=600, cite=f"cell {c}")
      for c, name, kw, grid in (
          (35, "BICM baseline 16-QAM, boxplus BP-20", {}, (0, 5, 0.25)),
          (37, "BICM all-zero 16-QAM WITHOUT scrambler (the notebook's deliberately wrong curve)", {"use_allzero": True}, (0, 5, 0.25)),
          (39, "BICM all-zero 16-QAM with scrambler", {"use_allzero": True, "use_scrambler": True}, (0, 5, 0.25)),
          (41, "BICM 16-QAM with the 5G output interleaver", {"use_ldpc_output_interleaver": True}, (0, 5, 0.25)),
          (47, "BICM 16-QAM, demapper noise estimate x0.15, boxplus", {"no_est_mismatch": 0.15}, (0, 7, 0.5)),
          (48, "BICM 16-QAM, min-sum BP-20", {"cn_update": "minsum"}, (0, 7, 0.5)),
          (49, "BICM 16-QAM, min-sum, demapper noise estimate x0.15", {"cn_update": "minsum", "no_est_mismatch": 0.15}, (0, 7, 0.5)))],
    # --- Sionna_tutorial_part1.ipynb cells 37/41 (uncoded QPSK, LLR outputs) and 53/54/65 (LDPC (1024,2048), default decoder)
    Curve("Sionna_tutorial_part1/c41/t0", "Uncoded QPSK, soft outputs (1024 bit blocks)", _Part1Uncoded, np.linspace(-3, 5, 20),
          bits_per_block=1024, use_bits=True, work=1024, cite="cell 41"),
    Curve("Sionna_tutorial_part1/c54/t0", "5G LDPC BP-20 (1024,2048)", _ldpc(1024, 2048), np.linspace(-3, 5, 15), bits_per_block=1024, cite="cell 54"),
    Curve("Sionna_tutorial_part1/c65/t0", "5G LDPC BP-20 (1024,2048), second run", _ldpc(1024, 2048), np.linspace(-3, 5, 12), bits_per_block=1024, cite="cell 65"),
    # --- MIMO_OFDM_Transmissions_over_CDL.ipynb: cell 67 (uplink, CDL-A..E, perfect CSI), cell 73 (CDL-D, pilots on symbol 0,
    #     perfect / LS CSI x 0 / 20 m/s), cell 76 (CDL-C, LS CSI, cyclic prefix 20 / 2 x frequency / time domain); 1000 block errors
    *[Curve(f"MIMO_OFDM_Transmissions_over_CDL/c67/t{i}", f"8x4 uplink CDL-{mdl}, perfect CSI, LMMSE, QPSK LDPC r=1/2",
            _cdl(domain="freq", cdl_model=mdl, perfect_csi=True, speed=0.0, cyclic_prefix_length=6, pilot_ofdm_symbol_indices=[2, 11]),
            np.arange(-5, 20, 4.0), bits_per_block=768, corr=4.0, group="cdl", max_batch=4096, cite="ipynb:1835-1880")
      for i, mdl in enumerate("ABCDE")],
    *[Curve(f"MIMO_OFDM_Transmissions_over_CDL/c73/t{i}", f"8x4 uplink CDL-D, {'perfect' if pc else 'LS-NN'} CSI, {sp:.0f} m/s, pilots on symbol 0",
            _cdl(domain="freq", cdl_model="D", perfect_csi=pc, speed=sp, cyclic_prefix_length=6, pilot_ofdm_symbol_indices=[0]),
            np.arange(0, 32, 2.0), bits_per_block=832, corr=4.0, group="cdl", max_batch=4096, cite="ipynb:2183-2240")
      for i, (pc, sp) in enumerate(((True, 0.0), (True, 20.0), (False, 0.0), (False, 20.0)))],
    *[Curve(f"MIMO_OFDM_Transmissions_over_CDL/c76/t{i}", f"8x4 uplink CDL-C, LS-NN CSI, 3 m/s, CP {cp}, {dom} domain",
            _cdl(domain=dom, cdl_model="C", perfect_csi=False, speed=3.0, cyclic_prefix_length=cp, pilot_ofdm_symbol_indices=[2, 11]),
            np.arange(0, 17, 2.0), bits_per_block=768, corr=4.0, group="cdl_time" if dom == "time" else "cdl", max_batch=1024,
            cite="ipynb:2375-2425")
      for i, (cp, dom) in enumerate(((20, "freq"), (20, "time"), (2, "freq"), (2, "time")))],
    # --- Introduction_to_Iterative_Detection_and_Decoding.ipynb cells 11/13/15, perfect-CSI Rayleigh (16 x 4, 16-QAM):
    #     LMMSE, EP, K-Best (one-shot detection) and IDD with 2 / 3 detection-decoding iterations; 819 block errors per point
    *[Curve(f"Introduction_to_Iterative_Detection_and_Decoding/c15/t{i}", f"16x4 Rayleigh perfect CSI, {nm}, 16-QAM LDPC min-sum 12",
            (lambda d=d: _IddRayleigh(d)), np.linspace(-10, 0, 11), bits_per_block=1152, corr=4.0, group="idd", max_batch=2048,
            work=2304 * 12 * (8 if d == "k-best" else 3), cite="cell 15")
      for i, (d, nm) in enumerate((("lmmse", "LMMSE detector"), ("ep", "EP detector (l=10)"), ("k-best", "K-Best detector (k=64)"),
                                   ("idd2", "IDD, 2 iterations (MMSE-PIC)"), ("idd3", "IDD, 3 iterations (MMSE-PIC)")))],
    # --- Weighted_BP_Algorithm.ipynb cell 13 (BCH (63,45), untrained) and cell 26 (5G LDPC (400,800), prune_pcm=False, untrained):
    #     ten one-iteration decoder calls chained through the decoder state; 2000 bit errors per point
    Curve("Weighted_BP_Algorithm/c13/t0", "BCH(63,45) BP-10 tanh via state passing (generic HIP engine)", lambda: _WeightedBPUntrained(pcm_id=1),
          np.arange(1, 7, 0.5), bits_per_block=63, work=63 * 40, group="state", cite="ipynb:364-380"),
    Curve("Weighted_BP_Algorithm/c13/t0", "BCH(63,45) BP-10 with the WeightedBPCallback registered (device torch engine)",
          lambda: _WeightedBPUntrained(pcm_id=1, callback=True), np.arange(1, 7, 0.5), bits_per_block=63, work=63 * 400, group="state_cb",
          max_batch=20000, cite="ipynb:364-380"),
    Curve("Weighted_BP_Algorithm/c26/t0", "5G LDPC (400,800) unpruned, BP-10 tanh via state passing", lambda: _WeightedBPUntrained(k=400, n=800),
          np.arange(0, 4, 0.25), bits_per_block=400, work=800 * 40, group="state", cite="cell 26"),
    # --- Sionna_tutorial_part3.ipynb cells 40/41 (1x4 SIMO uplink CDL-C, 100 block errors per point)
    Curve("Sionna_tutorial_part3/c41/t0", "1x4 uplink CDL-C 10 m/s, LS-NN CSI, LMMSE, QPSK LDPC(912,1824)", lambda: _Part3Ofdm(False),
          np.linspace(-8, 3, 20), bits_per_block=912, group="cdl", max_batch=8192, cite="cell 41"),
    Curve("Sionna_tutorial_part3/c41/t1", "1x4 uplink CDL-C 10 m/s, perfect CSI, LMMSE, QPSK LDPC(912,1824)", lambda: _Part3Ofdm(True),
          np.linspace(-8, 3, 20), bits_per_block=912, group="cdl", max_batch=8192, cite="cell 41"),
    # --- Discover_Sionna.ipynb cells 31/33/42 (500 block errors per point, one codeword per channel realisation)
    Curve("Discover_Sionna/c42/t0", "OFDM 1x1 TDL-A LS-NN LMMSE 16-QAM LDPC(6144,12288) boxplus BP-20", _DiscoverE2E,
          np.arange(0, 15, 1.), bits_per_block=6144, group="ofdm", max_batch=2048, cite="ipynb:1118-1134"),
    # --- Simple_MIMO_Simulation.ipynb cells 40/43 (100 block errors per point; 4 codewords per example but a fresh channel per symbol)
    Curve("Simple_MIMO_Simulation/c43/t0", "4x16 i.i.d. flat fading, lmmse_equalizer, 16-QAM LDPC(512,1024)", _SimpleMimo,
          np.arange(-2.5, 0.25, 0.25), bits_per_block=512, group="mimo", max_batch=2048, cite="ipynb:821-832"),
    # --- same notebook, cell 44: Kronecker spatial correlation (exponential, 0.4 at the transmitter, 0.7 at the receiver; 200 block errors)
    Curve("Simple_MIMO_Simulation/c44/t0", "4x16 flat fading with Kronecker correlation, lmmse_equalizer, 16-QAM LDPC(512,1024)",
          lambda: _SimpleMimo(kronecker=True), np.arange(0, 2.6, 0.25), bits_per_block=512, group="mimo", max_batch=2048, cite="ipynb:866-880"),
]


def curve_by_key(key):
    for c in CURVES:
        if c.key == key:
            return c
    raise KeyError(key)


# ----------------------------------------------------------------------------------------------------------------------
# running one curve through the product's sim_ber
# ----------------------------------------------------------------------------------------------------------------------
def run_curve(curve, ref_rows, mult=4.0, max_work=2.5e11, max_blocks=8_000_000, max_bits_per_batch=1 << 27, min_errors=200,
              seed=1234, verbose=False):
    """Simulate every Eb/N0 point the reference simulated, through ``sionna_amd.phy.utils.sim_ber`` (one call per point so
    that each point gets its own error target = max(mult x the reference's error count, min_errors)), with the number of
    blocks per point bounded by ``max_blocks`` and by ``max_work / curve.work``.  Returns rows shaped like the
    reference's."""
    phy = _phy()
    phy.config.seed = seed
    model = curve.build()
    rows = []
    blocks_per_example = None
    ek = "bit_errors" if curve.use_bits else "block_errors"
    cap_blocks = int(min(max_blocks, max(2000, max_work / curve.work)))
    for x, r in zip(curve.ebno, ref_rows):
        if r["num_blocks"] == 0:
            rows.append(None)
            continue
        target = int(max(mult * r[ek], min_errors))
        ev_per_block = max(r[ek], 0.5) / r["num_blocks"]                 # error events per block in the reference
        want_blocks = min(cap_blocks, int(1.15 * target / ev_per_block) + 64)
        if blocks_per_example is None:
            u, _ = model(2, float(x))
            blocks_per_example = max(1, u.numel() // (2 * u.shape[-1]))
        cap = max(1, max_bits_per_batch // (curve.bits_per_block * blocks_per_example * 3))
        if curve.max_batch:
            cap = min(cap, curve.max_batch)
        batch = int(min(cap, max(256, -(-want_blocks // blocks_per_example))))
        iters = max(1, -(-want_blocks // (batch * blocks_per_example)))
        got = {}

        def cb(ii, i, ebno_dbs, bit_errors, block_errors, nb_bits, nb_blocks, got=got):
            got.update(bit_errors=int(bit_errors[i]), block_errors=int(block_errors[i]), num_bits=int(nb_bits[i]),
                       num_blocks=int(nb_blocks[i]))
            return None

        kw = {"num_target_bit_errors": target} if curve.use_bits else {"num_target_block_errors": target}
        phy.utils.sim_ber(model, [float(x)], batch_size=batch, max_mc_iter=iters, soft_estimates=bool(getattr(model, "soft", False)), early_stop=False,
                          verbose=False, callback=cb, **kw)
        got["ebno_db"] = float(x)
        got["ber"] = got["bit_errors"] / max(got["num_bits"], 1)
        got["bler"] = got["block_errors"] / max(got["num_blocks"], 1)
        rows.append(got)
        if verbose:
            print(f"  {x:6.2f} dB  ref BLER {r['bler']:.4e} BER {r['ber']:.4e} ({r['block_errors']}/{r['num_blocks']})  "
                  f"ours {got['bler']:.4e} {got['ber']:.4e} ({got['block_errors']}/{got['num_blocks']})", flush=True)
    return rows


def evaluate(curve, ref_rows, our_rows):
    """BLER comparison (the asserting statistic for coded links), plus BER crossings with the variance widened by the
    mean number of bit errors per erroneous block; for ``use_bits`` curves the bit counts carry the whole test."""
    n = min(len(ref_rows), len(our_rows))
    ref_rows, our_rows = ref_rows[:n], our_rows[:n]
    if curve.use_bits:
        res = compare(ref_rows, our_rows, curve.ebno, curve.corr, use_bits=True)
        res["statistic"] = "bit errors"
        return res
    res = compare(ref_rows, our_rows, curve.ebno, curve.corr)
    res["statistic"] = "block errors"
    be = sum(o["bit_errors"] for o in our_rows if o) / max(1, sum(o["block_errors"] for o in our_rows if o))
    ber = compare(ref_rows, our_rows, curve.ebno, curve.corr * max(be, 1.0), use_bits=True)
    res["ber_crossings"] = ber["crossings"]
    res["ok_ber_crossings"] = ber["ok_crossings"]
    res["ok"] = res["ok"] and ber["ok_crossings"]
    return res
