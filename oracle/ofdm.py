"""Oracle (NumPy, CPU): OFDM resource grid, pilots, stream management, frequency-domain channel
(TDL / Rayleigh), LS channel estimation with nearest-neighbour interpolation and the per-RE
LMMSE equaliser.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates (paths relative to
/root/reference/src/sionna/phy):

* StreamManagement                  mimo/stream_management.py:9-246
* ResourceGrid / build_type_grid    ofdm/resource_grid.py:15-311
* ResourceGridMapper / Demapper     ofdm/resource_grid.py:350-520, RemoveNulledSubcarriers :522-552
* PilotPattern / Kronecker pattern  ofdm/pilot_pattern.py:17-378
* subcarrier_frequencies            channel/utils.py:15-66
* cir_to_ofdm_channel               channel/utils.py:180-253
* TDL (sum of sinusoids)            channel/tr38901/tdl.py:372-498 (no spatial correlation)
* ApplyOFDMChannel                  channel/apply_ofdm_channel.py:70-80
* LSChannelEstimator + NN interp.   ofdm/channel_estimation.py:138-173, 257-285, 323-435
* OFDMEqualizer / LMMSEEqualizer    ofdm/equalization.py:109-275
* lmmse_equalizer / whiten_channel / lmmse_matrix / inv_cholesky
                                    mimo/equalization.py:11-233, mimo/utils.py:292-356, utils/linalg.py:8-32

Parity status: the reference's own tests pin LMMSE / TDL only statistically;
since round 4 this restatement is held to the reference's own code EXECUTED under tools/ref_exec:
resource grid / LS estimators / equalisers / detectors (tests/test_oracle_ref_exec_ofdm_rx.py),
modulator / demodulator / time channel (..._ofdm.py), the IDD chain (..._idd.py), the TDL generator's
parameters exactly and its statistics (..._tdl.py) - plus the invariants in tests/ (whitening gives
identity covariance, perfect-CSI noiseless recovery, PDP / unit energy of the TDL).
Random draws use the build's Philox stream (oracle/utils.py) with the element layout documented
in ``tdl_cir``; pilots of the Kronecker pattern are QPSK symbols drawn from that stream with the
pattern's seed (the reference draws them from tf.random.Generator.from_seed(0)).
"""
import numpy as np

from . import utils as outil
from .mapping import qam

PI = np.pi
SPEED_OF_LIGHT = 299792458.0


# ------------------------------------------------------------------ stream management
class StreamManagement:
    """mimo/stream_management.py:155-246 (literal)."""

    def __init__(self, rx_tx_association, num_streams_per_tx):
        self.num_streams_per_tx = int(num_streams_per_tx)
        a = np.array(rx_tx_association, np.int32)
        assert np.all((a == 0) | (a == 1))
        self.num_rx, self.num_tx = a.shape
        self.rx_tx_association = a
        self.num_tx_per_rx = int(np.sum(a, 1)[0])
        self.num_rx_per_tx = int(np.sum(a, 0)[0])
        self.num_streams_per_rx = int(self.num_tx * self.num_streams_per_tx / self.num_rx)
        sa = np.zeros([self.num_rx, self.num_tx, self.num_streams_per_tx], np.int32)
        n_streams = min(self.num_streams_per_rx, self.num_streams_per_tx)
        for j in range(self.num_tx):
            c = 0
            for i in range(self.num_rx):
                if a[i, j]:
                    sa[i, j, c:c + self.num_streams_per_rx] = np.ones([n_streams])
                    c += self.num_streams_per_rx
        self.stream_association = sa
        self.detection_desired_ind = np.where(sa.reshape(-1) == 1)[0]
        self.detection_undesired_ind = np.where(sa.reshape(-1) == 0)[0]
        rx_ids = np.zeros([self.num_rx, self.num_streams_per_rx], np.int32)
        for i in range(self.num_rx):
            c = []
            for j in range(self.num_tx):
                if a[i, j]:
                    c += list(np.where(sa[i, j])[0] + j * self.num_streams_per_tx)
            rx_ids[i, :] = c
        self.rx_stream_ids = rx_ids
        self.stream_ind = np.argsort(rx_ids.reshape(-1))


# ------------------------------------------------------------------ pilots / resource grid
class PilotPattern:
    def __init__(self, mask, pilots, normalize=False):
        self.mask = np.asarray(mask, bool)
        self._pilots = np.asarray(pilots, np.complex64)
        self.normalize = normalize

    @property
    def num_pilot_symbols(self):
        return self._pilots.shape[-1]

    @property
    def num_data_symbols(self):
        return self.mask.shape[-1] * self.mask.shape[-2] - self.num_pilot_symbols

    @property
    def pilots(self):
        if self.normalize and self._pilots.shape[-1] > 0:
            scale = 1 / np.sqrt(np.mean(np.abs(self._pilots) ** 2, axis=-1, keepdims=True))
            return (scale * self._pilots).astype(np.complex64)
        return self._pilots


def qpsk_pilots(seed, call, n):
    """n random QPSK symbols from the build's bit stream (2 bits per symbol)."""
    bits = outil.random_bits(seed, call, 2 * n)
    pts = qam(2)
    return pts[(bits[0::2] * 2 + bits[1::2]).astype(np.int64)]


def kronecker_pilot_pattern(rg, pilot_ofdm_symbol_indices, normalize=True, seed=0):
    """ofdm/pilot_pattern.py:330-378"""
    num_tx, ns = rg.num_tx, rg.num_streams_per_tx
    n_sym, n_sc = rg.num_ofdm_symbols, rg.num_effective_subcarriers
    num_pilot_symbols = len(pilot_ofdm_symbol_indices)
    num_seq = num_tx * ns
    num_pilots = num_pilot_symbols * n_sc / num_seq
    assert (num_pilots / num_pilot_symbols) % 1 == 0
    per_sym = int(num_pilots / num_pilot_symbols)
    mask = np.zeros([num_tx, ns, n_sym, n_sc], bool)
    pilots = np.zeros([num_tx, ns, num_pilot_symbols, n_sc], np.complex64)
    mask[..., pilot_ofdm_symbol_indices, :] = True
    call = 0
    for i in range(num_tx):
        for j in range(ns):
            p = qpsk_pilots(seed, call, num_pilot_symbols * per_sym).reshape(num_pilot_symbols, per_sym)
            call += 1
            pilots[i, j, :, i * ns + j::num_seq] = p
    return PilotPattern(mask, pilots.reshape(num_tx, ns, -1), normalize)


class ResourceGrid:
    """ofdm/resource_grid.py:61-311"""

    def __init__(self, num_ofdm_symbols, fft_size, subcarrier_spacing, num_tx=1, num_streams_per_tx=1,
                 cyclic_prefix_length=0, num_guard_carriers=(0, 0), dc_null=False, pilot_pattern=None,
                 pilot_ofdm_symbol_indices=None):
        self.num_ofdm_symbols, self.fft_size = num_ofdm_symbols, fft_size
        self.subcarrier_spacing = subcarrier_spacing
        self.num_tx, self.num_streams_per_tx = num_tx, num_streams_per_tx
        self.cyclic_prefix_length = int(cyclic_prefix_length)
        self.num_guard_carriers = np.array(num_guard_carriers)
        self.dc_null = dc_null
        if pilot_pattern is None or pilot_pattern == "empty":
            n_eff = self.num_effective_subcarriers
            self.pilot_pattern = PilotPattern(np.zeros([num_tx, num_streams_per_tx, num_ofdm_symbols, n_eff], bool),
                                              np.zeros([num_tx, num_streams_per_tx, 0], np.complex64))
        elif pilot_pattern == "kronecker":
            self.pilot_pattern = kronecker_pilot_pattern(self, pilot_ofdm_symbol_indices)
        else:
            self.pilot_pattern = pilot_pattern

    @property
    def num_effective_subcarriers(self):
        return int(self.fft_size - self.dc_null - np.sum(self.num_guard_carriers))

    @property
    def dc_ind(self):
        return int(self.fft_size / 2 - (self.fft_size % 2 == 1) / 2)

    @property
    def effective_subcarrier_ind(self):
        g = self.num_guard_carriers
        sc = np.arange(g[0], self.fft_size - g[1])
        if self.dc_null:
            sc = np.delete(sc, self.dc_ind - g[0])
        return sc

    @property
    def num_pilot_symbols(self):
        return self.pilot_pattern.num_pilot_symbols

    @property
    def num_data_symbols(self):
        return self.num_effective_subcarriers * self.num_ofdm_symbols - self.num_pilot_symbols

    @property
    def ofdm_symbol_duration(self):
        return (1. + self.cyclic_prefix_length / self.fft_size) / self.subcarrier_spacing

    def build_type_grid(self):
        """0 data, 1 pilot, 2 guard, 3 DC  (resource_grid.py:283-311)"""
        shape = [self.num_tx, self.num_streams_per_tx, self.num_ofdm_symbols]
        g = self.num_guard_carriers
        mask = self.pilot_pattern.mask.astype(np.int32)
        split = self.dc_ind - g[0]
        return np.concatenate([2 * np.ones(shape + [g[0]], np.int32), mask[..., :split],
                               3 * np.ones(shape + [int(self.dc_null)], np.int32), mask[..., split:],
                               2 * np.ones(shape + [g[1]], np.int32)], -1)


def rg_map(rg, x):
    """ResourceGridMapper.call (resource_grid.py:394-412): x [B,tx,s,num_data] -> [B,tx,s,T,fft]."""
    t = rg.build_type_grid()
    out = np.zeros((x.shape[0],) + t.shape, np.complex64)
    out[:, t == 1] = rg.pilot_pattern.pilots.reshape(-1)          # same (row-major) order as tf.where
    out[:, t == 0] = x.reshape(x.shape[0], -1)
    return out


def remove_nulled(rg, x):
    return x[..., rg.effective_subcarrier_ind]


def data_ind(pp):
    """argsort(mask) ascending, first num_data entries (resource_grid.py:455-459; stable)."""
    m = pp.mask.reshape(pp.mask.shape[:2] + (-1,))
    return np.argsort(m, axis=-1, kind="stable")[..., :pp.num_data_symbols]


# ------------------------------------------------------------------ channel
def subcarrier_frequencies(num_subcarriers, subcarrier_spacing):
    """channel/utils.py:15-66"""
    start = -(num_subcarriers // 2)
    limit = num_subcarriers // 2 if num_subcarriers % 2 == 0 else num_subcarriers // 2 + 1
    return (np.arange(start, limit, dtype=np.float32) * np.float32(subcarrier_spacing)).astype(np.float32)


def cir_to_ofdm_channel(frequencies, a, tau, normalize=False):
    """channel/utils.py:180-253.  a [B,rx,ra,tx,ta,P,T], tau [B,rx,tx,P] -> [B,rx,ra,tx,ta,T,F]."""
    tau = tau[:, :, None, :, None, :, None, None]                      # [B,rx,1,tx,1,P,1,1]
    e = np.exp(-2j * PI * frequencies.astype(np.float64) * tau.astype(np.float64))
    h_f = np.sum(a[..., None].astype(np.complex128) * e, axis=-3)      # sum over paths
    if normalize:
        c = np.mean(np.abs(h_f) ** 2, axis=(2, 4, 5, 6), keepdims=True)
        h_f = np.where(c > 0, h_f / np.sqrt(np.where(c > 0, c, 1)), 0)
    return h_f.astype(np.complex64)


def _u(seed, call, n, lo, hi):
    """n uniforms in (lo,hi): element i = word i%4 of Philox block i//4 of stream (seed, call)."""
    nb = (n + 3) // 4
    w = np.stack(outil.philox_block(seed, call, nb), axis=1).reshape(-1)[:n]
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * outil._u01(w)).astype(np.float32)


def tdl_cir(seed, call, batch, num_time_steps, sampling_frequency, delays_s, mean_powers, min_doppler,
            max_doppler, num_rx_ant=1, num_tx_ant=1, num_sinusoids=20, los_power=None, los_aoa=PI / 4):
    """tdl.py:372-470.  RNG layout (4 consecutive calls of the stream):
    call+0 doppler[b]; call+1 theta[b,p,n]; call+2 phi[b,ra,ta,p,n]; call+3 phi_0[b] (LoS only).
    Returns a [B,1,ra,1,ta,P,T] complex64, tau [B,1,1,P] float32."""
    P, N, T = len(mean_powers), num_sinusoids, num_time_steps
    f = np.float32
    t = (np.arange(T, dtype=np.float32) / f(sampling_frequency)).astype(np.float32)
    doppler = _u(seed, call, batch, min_doppler, max_doppler).reshape(batch, 1, 1, 1, 1, 1)
    theta = _u(seed, call + 1, batch * P * N, -PI / N, PI / N).reshape(batch, 1, 1, P, 1, N)
    phi = _u(seed, call + 2, batch * num_rx_ant * num_tx_ant * P * N, -PI, PI).reshape(batch, num_rx_ant, num_tx_ant, P, 1, N)
    alpha = (f(2 * PI / N) * np.arange(1, N + 1, dtype=np.float32)).reshape(1, 1, 1, 1, 1, N) + theta
    arg = (doppler * t.reshape(1, 1, 1, 1, T, 1) * np.cos(alpha) + phi).astype(np.float32)
    h = (np.cos(arg) + 1j * np.sin(arg)).sum(-1) * f(1 / np.sqrt(N))      # [B,ra,ta,P,T]
    h = np.sqrt(np.asarray(mean_powers, np.float32)).reshape(1, 1, 1, P, 1) * h
    if los_power is not None:
        phi0 = _u(seed, call + 3, batch, -PI, PI).reshape(batch, 1, 1, 1)
        arg0 = doppler.reshape(batch, 1, 1, 1) * t.reshape(1, 1, 1, T) * f(np.cos(los_aoa)) + phi0
        h[:, :, :, 0, :] += (np.cos(arg0) + 1j * np.sin(arg0)) * np.sqrt(f(los_power))
    a = h.astype(np.complex64)[:, None, :, None, :, :, :]                   # [B,1,ra,1,ta,P,T]
    tau = np.tile(np.asarray(delays_s, np.float32).reshape(1, 1, 1, P), [batch, 1, 1, 1])
    return a, tau


def apply_ofdm_channel(x, h_freq):
    """apply_ofdm_channel.py:70-80 without noise: x [B,tx,ta,T,F], h [B,rx,ra,tx,ta,T,F]."""
    return np.sum(h_freq * x[:, None, None], axis=(3, 4)).astype(np.complex64)


# ------------------------------------------------------------------ LS estimation + NN interpolation
def nn_gather_ind(pp):
    """channel_estimation.py:364-411"""
    mask = pp.mask
    shp = mask.shape
    m = mask.reshape([-1] + list(shp[-2:]))
    pil = pp.pilots.reshape(-1, pp.pilots.shape[-1])
    g = np.zeros_like(m, dtype=np.int32)
    for a in range(m.shape[0]):
        i_p, j_p = np.where(m[a])
        for i in range(shp[-2]):
            for j in range(shp[-1]):
                d = np.abs(i - i_p) + np.abs(j - j_p)
                d[np.abs(pil[a]) == 0] = np.sum(shp[-2:])
                g[a, i, j] = np.argmin(d)
    return g.reshape(shp)


def ls_estimate(rg, y, no, interpolation="nn"):
    """BaseChannelEstimator.call + LSChannelEstimator (channel_estimation.py:138-173, 257-285).
    y [B,rx,ra,T,fft]; no scalar or per example.  Returns h_hat [B,rx,ra,tx,s,T,Feff], err_var broadcastable."""
    pp = rg.pilot_pattern
    y_eff = remove_nulled(rg, y)
    y_flat = y_eff.reshape(y_eff.shape[:-2] + (-1,))
    m = pp.mask.reshape(pp.mask.shape[:2] + (-1,))
    pilot_ind = np.argsort(~m, axis=-1, kind="stable")[..., :pp.num_pilot_symbols]   # DESCENDING on mask
    y_p = y_flat[..., pilot_ind]                                        # [B,rx,ra,tx,s,Np]
    pil = pp.pilots
    with np.errstate(divide="ignore", invalid="ignore"):
        h_ls = np.where(pil != 0, y_p / np.where(pil != 0, pil, 1), 0).astype(np.complex64)
        # no: scalar or [B] / [B,rx] / [B,rx,ra], expanded at the end to the rank of h_ls (channel_estimation.py:270-276)
        no_ = np.asarray(no, np.float32)
        no_ = no_.reshape(no_.shape + (1,) * (h_ls.ndim - no_.ndim))
        ev = np.where(pil != 0, no_ / np.where(pil != 0, np.abs(pil) ** 2, 1), 0).astype(np.float32)
    if ev.ndim < h_ls.ndim:
        ev = ev[None, None, None]
    if interpolation is None:
        return h_ls, ev
    g = nn_gather_ind(pp)                                               # [tx,s,T,F]
    tx, s = np.indices(g.shape)[:2]
    h_hat = h_ls[:, :, :, tx, s, g]
    err = np.maximum(ev[:, :, :, tx, s, g], 0)
    return h_hat, err


# ------------------------------------------------------------------ LMMSE
def lmmse_equalizer(y, h, s, whiten_interference=True):
    """mimo/equalization.py:101-233 in complex128 (y [...,M], h [...,M,K], s [...,M,M])."""
    y, h, s = y.astype(np.complex128), h.astype(np.complex128), s.astype(np.complex128)
    hh = np.conj(np.swapaxes(h, -1, -2))
    if whiten_interference:
        l_inv = np.linalg.inv(np.linalg.cholesky(s))
        y = (l_inv @ y[..., None])[..., 0]
        h = l_inv @ h
        hh = np.conj(np.swapaxes(h, -1, -2))
        g = np.linalg.solve(hh @ h + np.eye(h.shape[-1]), hh)
    else:
        g = hh @ np.linalg.inv(h @ hh + s)
    gy = (g @ y[..., None])[..., 0]
    d = np.diagonal(g @ h, axis1=-2, axis2=-1)
    return gy / d, np.real(1 / d - 1)


def _ofdm_preprocess(rg, sm, y, h_hat, err_var, no):
    """OFDMEqualizer.call / OFDMDetector._preprocess_inputs (ofdm/equalization.py:109-230,
    ofdm/detection.py:229-287): per-RE y [B,rx,T,F,M], desired channels [B,rx,T,F,M,K] and the
    covariance of noise + estimation error + undesired streams [B,rx,T,F,M,M]."""
    y_eff = remove_nulled(rg, y)
    y_dt = np.transpose(y_eff, [0, 1, 3, 4, 2])
    ev = np.broadcast_to(err_var, h_hat.shape)
    ev = np.transpose(ev, [0, 1, 5, 6, 2, 3, 4])
    ev = ev.reshape(ev.shape[:5] + (-1,))
    h_dt = np.transpose(h_hat, [1, 3, 4, 0, 2, 5, 6])
    h_dt = h_dt.reshape((-1,) + h_dt.shape[3:])
    hd = h_dt[sm.detection_desired_ind].reshape((sm.num_rx, sm.num_streams_per_rx) + h_dt.shape[1:])
    hu = h_dt[sm.detection_undesired_ind].reshape((sm.num_rx, -1) + h_dt.shape[1:])
    perm = [2, 0, 4, 5, 3, 1]
    hd, hu = np.transpose(hd, perm), np.transpose(hu, perm)
    no = np.asarray(no, np.float32)
    no_dt = no.reshape(no.shape + (1,) * (3 - no.ndim))
    no_dt = np.broadcast_to(no_dt, y.shape[:3])[..., None, None]
    no_dt = np.broadcast_to(no_dt, y_eff.shape)
    no_dt = np.transpose(no_dt, [0, 1, 3, 4, 2])
    M = y_dt.shape[-1]
    s = hu.astype(np.complex128) @ np.conj(np.swapaxes(hu, -1, -2)).astype(np.complex128)
    eye = np.eye(M)
    s = s + no_dt[..., None] * eye + np.sum(ev, -1)[..., None] * eye
    return y_dt, hd, s


def _extract_data(rg, sm, z, B):
    """[B,rx,T,F,K,...] -> [B,tx,streams,num_data,...] (equalization.py:233-273)."""
    extra = z.shape[5:]
    z = z.reshape(z.shape[:5] + (-1,))
    z = np.transpose(z, [1, 4, 2, 3, 5, 0])                      # [rx,K,T,F,X,B]
    z = z.reshape((-1,) + z.shape[2:])[sm.stream_ind]
    z = z.reshape((sm.num_tx, sm.num_streams_per_tx, -1) + z.shape[3:])     # [tx,s,T*F,X,B]
    di = data_ind(rg.pilot_pattern)
    tx, st = np.indices(di.shape)[:2]
    z = z[tx, st, di]                                             # [tx,s,ND,X,B]
    z = np.transpose(z, [4, 0, 1, 2, 3])
    return z.reshape(z.shape[:4] + extra)


def ofdm_lmmse_equalize(rg, sm, y, h_hat, err_var, no, whiten_interference=True):
    """OFDMEqualizer.call with the LMMSE equaliser (ofdm/equalization.py:109-275).
    y [B,rx,ra,T,fft], h_hat [B,rx,ra,tx,s,T,Feff], err_var broadcastable, no scalar/[B]/[B,rx]/[B,rx,ra].
    Returns x_hat, no_eff [B,tx,s,num_data]."""
    y_dt, hd, s = _ofdm_preprocess(rg, sm, y, h_hat, err_var, no)
    x_hat, no_eff = lmmse_equalizer(y_dt, hd, s, whiten_interference)
    B = y.shape[0]
    return _extract_data(rg, sm, x_hat, B).astype(np.complex64), _extract_data(rg, sm, no_eff, B).astype(np.float32)


# ------------------------------------------------------------------ MMSE-PIC detector
def _log_sigmoid(x):
    return -np.logaddexp(0.0, -x)


def _bit_labels(nb):
    p = np.arange(2 ** nb)
    return ((p[:, None] >> (nb - 1 - np.arange(nb))[None, :]) & 1).astype(np.float64)       # [P, nb], MSB first


def mmse_pic(y, h, s, prior, points, method="maxlog", num_iter=1, hard_out=False, output="bit"):
    """MMSEPICDetector.call (mimo/detection.py:1496-1643) in float64.
    y [...,M], h [...,M,K], s [...,M,M], prior [...,K,nb] -> extrinsic LLRs [...,K,nb] (output="bit"); output="symbol":
    prior = logits [...,K,2^nb] turned into LLRs by SymbolLogits2LLRs(method) (:1523-1524), result = LLRs2SymbolLogits of
    the extrinsic LLRs, logits [...,K,2^nb] or with hard_out their argmax [...,K] (:1636-1637)."""
    y, h, s = y.astype(np.complex128), h.astype(np.complex128), s.astype(np.complex128)
    points = np.asarray(points, np.complex128)
    if output == "symbol":
        from .mapping import symbol_logits2llrs, llrs2symbol_logits
        nbs = int(np.log2(len(points)))
        llr_e = mmse_pic(y, h, s, symbol_logits2llrs(prior, nbs, method), points, method, num_iter, False).astype(np.float64)
        out = llrs2symbol_logits(llr_e, nbs, hard_out)
        return out if hard_out else out.astype(np.float32)
    nb = prior.shape[-1]
    K = h.shape[-1]
    lab = _bit_labels(nb)
    a = 2 * lab - 1
    l_inv = np.linalg.inv(np.linalg.cholesky(s))
    y = (l_inv @ y[..., None])[..., 0]
    h = l_inv @ h
    hh = np.conj(np.swapaxes(h, -1, -2))
    y_mf = (hh @ y[..., None])                                   # [...,K,1]
    g = hh @ h
    gr = np.concatenate([np.concatenate([g.real, -g.imag], -1), np.concatenate([g.imag, g.real], -1)], -2)
    red = (lambda v, axis: np.log(np.sum(np.exp(v - v.max(axis=axis, keepdims=True)), axis=axis)) + v.max(axis=axis)) \
        if method == "app" else (lambda v, axis: v.max(axis=axis))
    llr_d = prior.astype(np.float64)
    llr_a = np.zeros_like(llr_d)
    for _ in range(num_iter):
        llr_a = llr_d
        x_logits = np.sum(_log_sigmoid(a * llr_a[..., None, :]), axis=-1)              # [...,K,P]
        pr = np.exp(x_logits - x_logits.max(-1, keepdims=True))
        pr /= pr.sum(-1, keepdims=True)
        x_hat = np.sum(pr * points, -1)
        var_x = np.sum(pr * np.abs(points - x_hat[..., None]) ** 2, -1)                # [...,K]
        y_mf_pic = y_mf + g * x_hat[..., None, :] - g @ x_hat[..., None]
        var2 = np.concatenate([var_x, var_x], -1)
        am = gr * var2[..., None, :] + np.eye(2 * K)
        a_inv = np.linalg.inv(am)
        mu = np.sum(a_inv * np.swapaxes(gr, -1, -2), -1)
        ypt = np.swapaxes(y_mf_pic, -1, -2)
        ypt = np.concatenate([ypt.real, ypt.imag], -1)
        ypt = np.concatenate([ypt, ypt], -2)
        xr = np.sum(a_inv * ypt, -1) / mu
        x_hat = xr[..., :K] + 1j * xr[..., K:]
        vx = (mu / np.maximum(1 - var2 * mu, 1e-4))[..., :K]
        no_eff = np.maximum(1.0 / vx, np.finfo(np.float32).tiny)
        expo = -np.abs(x_hat[..., None] - points) ** 2 / no_eff[..., None]             # [...,K,P]
        t = expo + x_logits
        one = lab.T.astype(bool)                                                       # [nb,P]
        t1 = np.stack([red(t[..., one[b]], -1) for b in range(nb)], -1)
        t0 = np.stack([red(t[..., ~one[b]], -1) for b in range(nb)], -1)
        llr_d = t1 - t0
    llr_e = llr_d - llr_a
    return (llr_e > 0).astype(np.float32) if hard_out else llr_e.astype(np.float32)


def ofdm_mmse_pic(rg, sm, y, h_hat, prior, err_var, no, points, method="maxlog", num_iter=1, hard_out=False, output="bit"):
    """ofdm.MMSEPICDetector.call (ofdm/detection.py:448-560, 1062-1230).
    output="bit": prior [B,tx,streams,num_data*nb] -> LLRs of the same shape; output="symbol": prior
    [B,tx,streams,num_data,num_points] logits (zero off the data REs, :531-541) -> logits of the same shape or, with
    hard_out, indices [B,tx,streams,num_data]."""
    nb = int(np.log2(len(points)))
    B = y.shape[0]
    y_dt, hd, s = _ofdm_preprocess(rg, sm, y, h_hat, err_var, no)
    T, F = rg.num_ofdm_symbols, rg.num_effective_subcarriers
    if output == "symbol":
        nb = len(points)                                   # the prior's last dimension: one logit per point
    pr = prior.reshape(B, sm.num_tx, rg.num_streams_per_tx, -1, nb)
    grid = np.zeros((B, sm.num_tx, rg.num_streams_per_tx, T * F, nb), np.float32)         # zero prior off the data REs
    di = data_ind(rg.pilot_pattern)
    tx, st = np.indices(di.shape)[:2]
    grid[:, tx, st, di] = pr
    grid = grid.reshape(B, -1, T, F, nb)                                                  # [B,S,T,F,nb]
    # detection_desired_ind indexes the [rx, tx*streams] flattening of h; the stream id is its remainder
    desired = np.asarray(sm.detection_desired_ind).reshape(sm.num_rx, sm.num_streams_per_rx) % grid.shape[1]
    pri = np.stack([grid[:, desired[r]] for r in range(sm.num_rx)], axis=1)               # [B,rx,K,T,F,nb]
    pri = np.transpose(pri, [0, 1, 3, 4, 2, 5])
    llr = mmse_pic(y_dt, hd, s, pri, points, method, num_iter, hard_out, output=output)   # [B,rx,T,F,K,nb]
    out = _extract_data(rg, sm, llr, B)                                                   # [B,tx,s,ND,nb]
    return out if output == "symbol" else out.reshape(out.shape[:3] + (-1,))


# ------------------------------------------------------------------ time-domain variant
def _cp_vector(cyclic_prefix_length, num_ofdm_symbols):
    cp = np.asarray(cyclic_prefix_length, np.int64)
    return np.full(num_ofdm_symbols, int(cp), np.int64) if cp.ndim == 0 else cp


def ofdm_modulate(x, cyclic_prefix_length):
    """ofdm/modulator.py:97-124 with ifft = sqrt(N) * numpy ifft (signal/utils.py:205-262).
    x [..., num_ofdm_symbols, fft_size] -> [..., sum_s (fft_size + cp_s)]."""
    n = x.shape[-1]
    xt = np.fft.ifft(np.fft.ifftshift(x.astype(np.complex128), axes=-1), axis=-1) * np.sqrt(n)
    cp = _cp_vector(cyclic_prefix_length, x.shape[-2])
    parts = [np.concatenate([xt[..., s, n - int(cp[s]):], xt[..., s, :]], axis=-1) for s in range(x.shape[-2])]
    return np.concatenate(parts, axis=-1).astype(np.complex64)


def ofdm_demodulate(y, fft_size, l_min, cyclic_prefix_length, num_ofdm_symbols=None):
    """ofdm/demodulator.py:143-203.  y [..., num_time_samples] -> [..., num_ofdm_symbols, fft_size]."""
    n = fft_size
    cp0 = np.asarray(cyclic_prefix_length, np.int64)
    if cp0.ndim == 0:
        num_ofdm_symbols = y.shape[-1] // (n + int(cp0))
    cp = _cp_vector(cyclic_prefix_length, num_ofdm_symbols)
    off = np.concatenate([[0], np.cumsum(cp + n)[:-1]])
    rows = np.stack([y[..., int(off[s] + cp[s]):int(off[s] + cp[s]) + n] for s in range(len(cp))], axis=-2)
    xf = np.fft.fft(rows.astype(np.complex128), axis=-1) / np.sqrt(n)
    tmp = (np.float32(-2 * PI * l_min) / np.float32(n) * np.arange(n, dtype=np.float32)).astype(np.float64)
    xf = xf * np.exp(1j * tmp)
    return np.fft.fftshift(xf, axes=-1).astype(np.complex64)


def time_lag_discrete_time_channel(bandwidth, maximum_delay_spread=3e-6):
    """channel/utils.py:121-178"""
    return -6, int(np.ceil(maximum_delay_spread * bandwidth)) + 6


def cir_to_time_channel(bandwidth, a, tau, l_min, l_max, normalize=False):
    """channel/utils.py:256-349.  a [B,rx,ra,tx,ta,P,T], tau [B,rx,tx,P] -> [B,rx,ra,tx,ta,T,L]."""
    tau = tau[:, :, None, :, None, :, None, None].astype(np.float64)   # [B,rx,1,tx,1,P,1,1]
    l = np.arange(l_min, l_max + 1, dtype=np.float64)
    g = np.sinc(l - tau * bandwidth)                                   # [B,rx,1,tx,1,P,1,L]
    hm = np.sum(a[..., None].astype(np.complex128) * g, axis=-3)       # [B,rx,ra,tx,ta,T,L]
    if normalize:
        c = np.mean(np.sum(np.abs(hm) ** 2, axis=6, keepdims=True), axis=(2, 4, 5), keepdims=True)
        hm = np.where(c > 0, hm / np.sqrt(np.where(c > 0, c, 1)), 0)
    return hm.astype(np.complex64)


def apply_time_channel(x, h_time):
    """channel/apply_time_channel.py:85-137 without noise: x [B,tx,ta,Tn],
    h_time [B,rx,ra,tx,ta,Tn+L-1,L] -> y [B,rx,ra,Tn+L-1]."""
    B, rx, ra, tx, ta, Tout, L = h_time.shape
    Tn = x.shape[-1]
    assert Tout == Tn + L - 1
    xp = np.concatenate([x.astype(np.complex128), np.zeros(x.shape[:-1] + (L,), np.complex128)], axis=-1)
    y = np.zeros((B, rx, ra, Tout), np.complex128)
    t = np.arange(Tout)
    for l in range(L):
        idx = t - l
        valid = (idx >= 0) & (idx < Tn)
        xs = np.where(valid, xp[..., np.clip(idx, 0, Tn - 1)], 0)      # [B,tx,ta,Tout]
        y += np.sum(h_time[..., l].astype(np.complex128) * xs[:, None, None], axis=(3, 4))
    return y.astype(np.complex64)


# ------------------------------------------------------------------ linear interpolation of LS estimates
class LinearInterpolator:
    """ofdm/channel_estimation.py:437-733: index tables (:529-640) and the two 1-D interpolations
    (:642-733).  pp: oracle PilotPattern (mask [tx, streams, T, F], pilots [tx, streams, P])."""

    def __init__(self, pp, time_avg=False):
        assert pp.num_pilot_symbols > 0, "The pilot pattern cannot be empty"
        self.time_avg = time_avg
        mask = np.asarray(pp.mask)
        self.mask_shape = mask.shape
        mask = mask.reshape((-1,) + mask.shape[-2:])
        pilots = np.asarray(pp.pilots).reshape(-1, pp.pilots.shape[-1])
        assert np.max(np.sum(np.abs(pilots) == 0, -1)) < pilots.shape[-1], \
            "Each pilot sequence must have at least one nonzero entry"
        S, T, F = mask.shape
        z = np.zeros(mask.shape, pilots.dtype)
        for a in range(S):
            z[a][np.where(mask[a])] = pilots[a]
        x0 = np.zeros(mask.shape, np.int32)
        x1 = np.zeros(mask.shape, np.int32)
        empty = np.sum(np.abs(z), axis=-1) == 0
        x0[empty] = -1
        x1[empty] = -1
        y0, y1 = x0.copy(), x1.copy()
        for a in range(S):
            count = 0
            pilot_ind = np.where(np.abs(pilots[a]))[0]
            for i in range(T):
                po = np.where(np.abs(z[a][i]))[0]
                if len(po) == 1:
                    x0[a, i] = x1[a, i] = po[0]
                    y0[a, i] = y1[a, i] = pilot_ind[count]
                elif len(po) >= 2:
                    k0, k1 = 0, 1
                    for j in range(F):
                        x0[a, i, j], x1[a, i, j] = po[k0], po[k1]
                        y0[a, i, j], y1[a, i, j] = pilot_ind[count + k0], pilot_ind[count + k1]
                        if j == po[k1] and k1 < len(po) - 1:
                            k0, k1 = k1, k1 + 1
                count += len(po)
        self.x0f, self.x1f, self.y0f, self.y1f = x0, x1, y0 + 1, y1 + 1       # +1: index 0 = zero pad
        t0 = np.zeros((S, T), np.int32)
        t1 = np.zeros((S, T), np.int32)
        for a in range(S):
            sym = np.where(np.sum(np.abs(z[a]), axis=-1))[0]
            if len(sym) == 1:
                t0[a] = t1[a] = sym[0]
            elif len(sym) >= 2:
                k0, k1 = 0, 1
                for i in range(T):
                    t0[a, i], t1[a, i] = sym[k0], sym[k1]
                    if i == sym[k1] and k1 < len(sym) - 1:
                        k0, k1 = k1, k1 + 1
        self.t0, self.t1 = t0, t1
        self.npil = np.sum(np.sum(np.abs(z), axis=-1) > 0, axis=-1)             # [S]

    def _interp(self, x):
        """x [..., tx, streams, P] -> [..., tx, streams, T, F] (complex128 arithmetic)."""
        S, T, F = self.x0f.shape
        lead = x.shape[:-3]
        xp = np.concatenate([np.zeros(x.shape[:-1] + (1,), np.complex128), x.astype(np.complex128)], axis=-1)
        xp = xp.reshape(lead + (S, -1))
        sidx = np.arange(S)[:, None, None]
        y0 = xp[..., sidx, self.y0f]
        y1 = xp[..., sidx, self.y1f]
        dx = (self.x1f - self.x0f).astype(np.float64)
        slope = np.where(dx != 0, (y1 - y0) / np.where(dx != 0, dx, 1), 0)
        hf = (np.arange(F) - self.x0f) * slope + y0                                # [..., S, T, F]
        if self.time_avg:
            hf = np.sum(hf, axis=-2, keepdims=True) / self.npil[:, None, None]
            hf = np.repeat(hf, T, axis=-2)
        a = hf[..., sidx[:, :, 0], self.t0, :]
        b = hf[..., sidx[:, :, 0], self.t1, :]
        dt = (self.t1 - self.t0).astype(np.float64)[..., None]
        slope = np.where(dt != 0, (b - a) / np.where(dt != 0, dt, 1), 0)
        out = (np.arange(T)[None, :, None] - self.t0[..., None]) * slope + a
        return out.reshape(lead + self.mask_shape)

    def __call__(self, h_hat, err_var):
        h = self._interp(h_hat).astype(np.complex64)
        ev = np.real(self._interp(np.asarray(err_var, np.float64).astype(np.complex128))).astype(np.float32)
        return h, ev


def ls_estimate_lin(rg, y, no, time_avg=False):
    """LSChannelEstimator(interpolation_type="lin" / "lin_time_avg") (channel_estimation.py:138-173, 257-285)."""
    h_p, ev_p = ls_estimate(rg, y, no, interpolation=None)
    h, ev = LinearInterpolator(rg.pilot_pattern, time_avg)(h_p, np.broadcast_to(ev_p, h_p.shape))
    return h, np.maximum(ev, 0)


# ------------------------------------------------------------------ EP detector
def _pam_points_over_sqrt2(nbh):
    """Unit-energy Gray PAM points in label order / sqrt(2) (mimo/detection.py:1157-1163, mapping.py:15-42, 120-192)."""
    pts = np.zeros(2 ** nbh)
    for i in range(2 ** nbh):
        b = [(i >> (nbh - 1 - j)) & 1 for j in range(nbh)]
        def gray(b):                                    # mapping.py:15-42
            return 1 - 2 * b[0] if len(b) == 1 else (1 - 2 * b[0]) * (2 ** (len(b) - 1) - gray(b[1:]))
        pts[i] = gray(b)
    pts = pts / np.sqrt(np.mean(pts ** 2))
    return pts / np.sqrt(2.0)


def ep_detector(y, h, s, num_bits_per_symbol, l=10, beta=0.9, hard_out=False, prec=1e-6, output="bit", out_dtype=np.float32):
    """EPDetector.call (mimo/detection.py:1166-1312) in float64.
    y [...,M], h [...,M,K], s [...,M,M] -> max-log LLRs [...,K,num_bits_per_symbol] (output="bit"), or QAM logits
    [...,K,2^num_bits_per_symbol] / QAM indices [...,K] through PAM2QAM (output="symbol", :1276-1295)."""
    nbh = num_bits_per_symbol // 2
    pts = _pam_points_over_sqrt2(nbh)
    es = np.var(pts)
    y, h, s = y.astype(np.complex128), h.astype(np.complex128), s.astype(np.complex128)
    l_inv = np.linalg.inv(np.linalg.cholesky(s))
    y = (l_inv @ y[..., None])[..., 0]
    h = l_inv @ h
    K = h.shape[-1]
    yr = np.concatenate([y.real, y.imag], -1)
    hr = np.concatenate([np.concatenate([h.real, -h.imag], -1), np.concatenate([h.imag, h.real], -1)], -2)
    no = 0.5
    hth = np.swapaxes(hr, -1, -2) @ hr
    hty = (np.swapaxes(hr, -1, -2) @ yr[..., None])[..., 0]
    lam = np.ones(hty.shape) / es
    gam = np.zeros(hty.shape)
    eye = np.eye(2 * K)
    for _ in range(l):
        sig_full = np.linalg.inv(hth + no * lam[..., None] * eye)
        mu = (sig_full @ (hty + no * gam)[..., None])[..., 0]
        sigma = no * np.diagonal(sig_full, axis1=-2, axis2=-1)
        v_obs = np.maximum(1 / (1 / sigma - lam), prec)
        x_obs = v_obs * (mu / sigma - gam)
        logits = -(x_obs[..., None] - pts) ** 2 / (2 * v_obs[..., None])
        pmf = np.exp(logits - logits.max(-1, keepdims=True))
        pmf /= pmf.sum(-1, keepdims=True)
        x = np.sum(pts * pmf, -1)
        v = np.maximum(np.sum((pts - x[..., None]) ** 2 * pmf, -1), prec)
        lam_n, gam_n = 1 / v - 1 / v_obs, x / v - x_obs / v_obs
        keep = lam_n < 0
        lam_n, gam_n = np.where(keep, lam, lam_n), np.where(keep, gam, gam_n)
        lam, gam = (1 - beta) * lam_n + beta * lam, (1 - beta) * gam_n + beta * gam
    if output == "symbol":
        from .mapping import pam2qam
        if hard_out:
            return pam2qam(np.argmax(logits[..., :K, :], -1), np.argmax(logits[..., K:, :], -1), num_bits_per_symbol, True)
        return pam2qam(logits[..., :K, :], logits[..., K:, :], num_bits_per_symbol, False).astype(out_dtype)
    lab = _bit_labels(nbh).T.astype(bool)                                          # [nbh, P]
    llr = np.stack([logits[..., lab[b]].max(-1) - logits[..., ~lab[b]].max(-1) for b in range(nbh)], -1)   # [..., 2K, nbh]
    llr = np.stack([llr[..., :K, :], llr[..., K:, :]], -1).reshape(llr.shape[:-2] + (K, 2 * nbh))
    return (llr > 0).astype(out_dtype) if hard_out else llr.astype(out_dtype)


def ofdm_ep_detector(rg, sm, y, h_hat, err_var, no, num_bits_per_symbol, l=10, beta=0.9, hard_out=False, output="bit", prec=1e-6,
                     out_dtype=np.float32):
    """ofdm.EPDetector.call -> [B,tx,streams,num_data*num_bits_per_symbol] (output="bit"), or the logits
    [B,tx,streams,num_data,num_points] / indices [B,tx,streams,num_data] of output="symbol" (ofdm/detection.py:289-317)."""
    y_dt, hd, s = _ofdm_preprocess(rg, sm, y, h_hat, err_var, no)
    llr = ep_detector(y_dt, hd, s, num_bits_per_symbol, l, beta, hard_out, prec, output=output, out_dtype=out_dtype)
    out = _extract_data(rg, sm, llr, y.shape[0])
    return out if output == "symbol" else out.reshape(out.shape[:3] + (-1,))


# ------------------------------------------------------------------ K-Best detector
def pam_levels_real_rep(nbh):
    """the PAM constellation of the real-valued K-Best (mimo/detection.py:719-724): Gray-labelled, unnormalised levels of
    mapping.pam_gray divided by std * sqrt(2)"""
    from . import mapping as omap
    lab = _bit_labels(nbh)
    pts = np.array([omap.pam_gray(lab[i]) for i in range(1 << nbh)], np.float64)
    return pts / (np.std(pts) * np.sqrt(2))


def kbest_detector_real(y, h, s, num_bits_per_symbol, k, hard_out=False, llr_clip=20.0, output="bit"):
    """KBestDetector(use_real_rep=True).call (mimo/detection.py:705-727, 815-823, 1011-1030): complex2real_channel
    (mimo/utils.py:13-190), the tree search over the PAM levels with 2K real streams, distances halved in List2LLRSimple
    (mimo/utils.py:544-547), LLRs / bits of the in-phase and quadrature streams interleaved into the QAM's bit order."""
    y, h, s = np.asarray(y, np.complex128), np.asarray(h, np.complex128), np.asarray(s, np.complex128)
    K = h.shape[-1]
    yr = np.concatenate([y.real, y.imag], -1)
    hr = np.concatenate([np.concatenate([h.real, -h.imag], -1), np.concatenate([h.imag, h.real], -1)], -2)
    sr = 0.5 * np.concatenate([np.concatenate([s.real, -s.imag], -1), np.concatenate([s.imag, s.real], -1)], -2)
    nbh = num_bits_per_symbol // 2
    out = kbest_detector(yr, hr, sr, pam_levels_real_rep(nbh), k, hard_out, llr_clip, "bit", dist_scale=0.5)     # [n, 2K, nbh]
    out = np.stack([out[:, :K], out[:, K:]], -1).reshape(out.shape[0], K, 2 * nbh)
    if output == "symbol":
        assert hard_out, "Soft-symbols are not supported for this detector."
        return np.sum(out.astype(np.int64) << (2 * nbh - 1 - np.arange(2 * nbh)), -1).astype(np.int32)
    return out


def kbest_detector(y, h, s, points, k, hard_out=False, llr_clip=20.0, output="bit", dist_scale=1.0):
    """KBestDetector.call (complex representation, mimo/detection.py:815-1037) + List2LLRSimple
    (mimo/utils.py:539-578) in float64: y [n,M], h [n,M,K], s [n,M,M] -> LLRs [n,K,nb]."""
    y, h, s = y.astype(np.complex128), h.astype(np.complex128), s.astype(np.complex128)
    points = np.asarray(points, np.complex128)
    P, nb = len(points), int(np.log2(len(points)))
    n, M, K = h.shape
    l_inv = np.linalg.inv(np.linalg.cholesky(s))
    y = (l_inv @ y[..., None])[..., 0]
    h = l_inv @ h
    order = np.argsort(-np.sum(np.abs(h) ** 2, axis=1), axis=-1, kind="stable")              # :820-824
    h = np.take_along_axis(h, order[:, None, :], axis=2)
    q, r = np.linalg.qr(h)
    y = (np.conj(np.swapaxes(q, -1, -2)) @ y[..., None])[..., 0]
    k = min(k, P ** K)
    dists = np.zeros((n, 1))
    inds = np.zeros((n, 1, 0), np.int64)
    for stream in range(K):
        col = K - 1 - stream
        npth = dists.shape[1]
        d = np.repeat(dists, P, axis=1)                                                        # path-major candidates
        pi = np.concatenate([np.repeat(inds, P, axis=1), np.tile(np.arange(P), npth)[None, :, None].repeat(n, 0)], axis=-1)
        syms = points[pi]                                                                      # [n, cand, stream+1]
        rr = r[:, col, col:][:, ::-1]                                                          # reversed like :915
        d = d + np.abs(y[:, col, None] - np.sum(rr[:, None, :] * syms, axis=-1)) ** 2
        sel = np.argsort(d, axis=1, kind="stable")[:, :min(k, d.shape[1])]                     # top_k: ties -> lower index
        dists = np.take_along_axis(d, sel, axis=1)
        inds = np.take_along_axis(pi, sel[:, :, None], axis=1)
    inds = inds[:, :, ::-1]                                                                    # sorted-column order
    unsort = np.argsort(order, axis=-1, kind="stable")
    inds = np.take_along_axis(inds, unsort[:, None, :], axis=2)                                # original stream order
    if output == "symbol":                                                                     # :1001-1019 (hard decisions only, :799-801)
        assert hard_out, "Soft-symbols are not supported for this detector."
        return inds[:, 0].astype(np.int32)
    bits = (inds[..., None] >> (nb - 1 - np.arange(nb))) & 1                                   # [n, paths, K, nb]
    if hard_out:
        return bits[:, 0].astype(np.float32)
    dd = dist_scale * dists[:, :, None, None]
    l0 = np.min(np.where(bits == 0, dd, np.inf), axis=1)
    l1 = np.min(np.where(bits == 1, dd, np.inf), axis=1)
    with np.errstate(invalid="ignore"):
        return np.clip(l0 - l1, -llr_clip, llr_clip).astype(np.float32)


def ofdm_kbest_detector(rg, sm, y, h_hat, err_var, no, points, k, hard_out=False, output="bit", use_real_rep=False):
    """ofdm.KBestDetector.call -> [B,tx,streams,num_data*nb] (output="bit") or indices [B,tx,streams,num_data]
    (output="symbol", hard decisions)."""
    y_dt, hd, s = _ofdm_preprocess(rg, sm, y, h_hat, err_var, no)
    shp = hd.shape[:-2]
    if use_real_rep:
        llr = kbest_detector_real(y_dt.reshape((-1,) + y_dt.shape[-1:]), hd.reshape((-1,) + hd.shape[-2:]),
                                  s.reshape((-1,) + s.shape[-2:]), int(np.log2(len(points))), k, hard_out, output=output)
    else:
        llr = kbest_detector(y_dt.reshape((-1,) + y_dt.shape[-1:]), hd.reshape((-1,) + hd.shape[-2:]),
                             s.reshape((-1,) + s.shape[-2:]), points, k, hard_out, output=output)
    out = _extract_data(rg, sm, llr.reshape(shp + llr.shape[1:]), y.shape[0])
    return out if output == "symbol" else out.reshape(out.shape[:3] + (-1,))


# ------------------------------------------------------------------ maximum-likelihood detector
def ml_detector(y, h, s, points, method="app", prior=None, output="bit", hard_out=False):
    """MaximumLikelihoodDetector.call (mimo/detection.py:145-537) in float64: y [n,M], h [n,M,K], s [n,M,M]; prior = LLRs
    [n,K,nb] (output "bit") or logits [n,K,P] ("symbol").  Candidate vectors as _build_vecs (:414-470): stream 0 is the slowest
    index.  -> LLRs / hard bits [n,K,nb], or logits [n,K,P] / indices [n,K]."""
    y, h, s = np.asarray(y, np.complex128), np.asarray(h, np.complex128), np.asarray(s, np.complex128)
    points = np.asarray(points, np.complex128)
    n, M, K = h.shape
    P = len(points)
    nb = int(np.log2(P))
    lab = _bit_labels(nb)                                                        # [P, nb], 0/1
    if prior is not None:
        prior = np.asarray(prior, np.float64)
        if output == "bit":                                                     # LLRs2SymbolLogits (mapping.py:969-1058)
            prior = np.sum(_log_sigmoid((2. * lab[None, None] - 1.) * prior[:, :, None, :]), -1)
    li = np.linalg.inv(np.linalg.cholesky(s))                                   # whiten_channel (mimo/utils.py:292-356)
    yw = np.einsum("nij,nj->ni", li, y)
    hw = li @ h
    idx = np.stack(np.meshgrid(*[np.arange(P)] * K, indexing="ij"), -1).reshape(-1, K)          # [num_vecs, K]
    vecs = points[idx]                                                                           # [num_vecs, K]
    diff = yw[:, None, :] - np.einsum("nmk,vk->nvm", hw, vecs)
    expo = -np.sum(np.abs(diff) ** 2, -1)                                                        # [n, num_vecs]
    if prior is not None:
        expo = expo + np.sum(prior[:, np.arange(K)[None, :], idx], -1)
    logits = np.empty((n, K, P))
    for k in range(K):
        for p_ in range(P):
            e = expo[:, idx[:, k] == p_]
            mx = e.max(-1)
            logits[:, k, p_] = mx if method == "maxlog" else mx + np.log(np.sum(np.exp(e - mx[:, None]), -1))
    if output == "symbol":
        return np.argmax(logits, -1).astype(np.int32) if hard_out else logits
    red = (lambda a: a.max(-1)) if method == "maxlog" else (lambda a: a.max(-1) + np.log(np.sum(np.exp(a - a.max(-1, keepdims=True)), -1)))
    llr = np.stack([red(logits[..., lab[:, b] == 1]) - red(logits[..., lab[:, b] == 0]) for b in range(nb)], -1)   # :927-967
    return (llr > 0).astype(np.float64) if hard_out else llr


def ofdm_ml_detector(rg, sm, y, h_hat, err_var, no, points, method="app", prior=None, output="bit", hard_out=False):
    """ofdm.MaximumLikelihoodDetector(.WithPrior).call (ofdm/detection.py:524-738): prior / result in the block's layouts
    ([B,tx,streams,num_data*nb] for "bit", [B,tx,streams,num_data,P] for "symbol")."""
    y_dt, hd, s = _ofdm_preprocess(rg, sm, y, h_hat, err_var, no)
    shp = hd.shape[:-2]                                                          # [B,rx,T,F]
    K, P = hd.shape[-1], len(points)
    nb = int(np.log2(P))
    pr = None
    if prior is not None:
        # [B,tx,streams,num_data,(nb|P)] -> per RE of the receivers' streams [B,rx,T,F,K,(nb|P)]: zero prior on REs without data
        last = nb if output == "bit" else P
        pri = np.asarray(prior, np.float64).reshape(prior.shape[:3] + (rg.num_data_symbols, last))
        full = np.zeros((y.shape[0], rg.num_tx * rg.num_streams_per_tx, rg.num_ofdm_symbols * rg.num_effective_subcarriers, last))
        di = data_ind(rg.pilot_pattern).reshape(rg.num_tx * rg.num_streams_per_tx, -1)
        flat = pri.reshape(y.shape[0], rg.num_tx * rg.num_streams_per_tx, rg.num_data_symbols, last)
        for st in range(full.shape[1]):
            full[:, st, di[st]] = flat[:, st]
        full = full.reshape(y.shape[0], -1, rg.num_ofdm_symbols, rg.num_effective_subcarriers, last)
        # stream (tx-major) detected at position k of receiver r: the inverse of stream_ind (ofdm/detection.py:289-317).  The reference
        # itself tiles the priors of ALL streams over the receivers (:497-498) and is only defined for one receiver detecting every
        # stream, where the two coincide.
        sel = np.argsort(np.asarray(sm.stream_ind)).reshape(sm.num_rx, sm.num_streams_per_rx)
        pr = np.stack([full[:, sel[r]] for r in range(sm.num_rx)], 1)           # [B,rx,K,T,F,last]
        pr = np.transpose(pr, [0, 1, 3, 4, 2, 5]).reshape(-1, K, last)
    out = ml_detector(y_dt.reshape((-1,) + y_dt.shape[-1:]), hd.reshape((-1,) + hd.shape[-2:]), s.reshape((-1,) + s.shape[-2:]),
                      points, method, pr, output, hard_out)
    out = _extract_data(rg, sm, out.reshape(shp + out.shape[1:]), y.shape[0])
    return out.reshape(out.shape[:3] + (-1,)) if output == "bit" else out


# ------------------------------------------------------------------ ZF / MF equalisers
def zf_equalizer(y, h, s):
    """mimo/equalization.py:235-298 (complex128)."""
    y, h, s = y.astype(np.complex128), h.astype(np.complex128), s.astype(np.complex128)
    hh = np.conj(np.swapaxes(h, -1, -2))
    g = np.linalg.solve(hh @ h, hh)
    x = (g @ y[..., None])[..., 0]
    return x, np.real(np.diagonal(g @ s @ np.conj(np.swapaxes(g, -1, -2)), axis1=-2, axis2=-1))


def mf_equalizer(y, h, s):
    """mimo/equalization.py:300-463 (complex128)."""
    y, h, s = y.astype(np.complex128), h.astype(np.complex128), s.astype(np.complex128)
    hh = np.conj(np.swapaxes(h, -1, -2))
    d = 1 / np.diagonal(hh @ h, axis1=-2, axis2=-1)
    g = d[..., None] * hh
    x = (g @ y[..., None])[..., 0]
    gh = g @ h
    e = np.eye(h.shape[-1]) - gh
    cov = e @ np.conj(np.swapaxes(e, -1, -2)) + g @ s @ np.conj(np.swapaxes(g, -1, -2))
    return x, np.abs(np.diagonal(cov, axis1=-2, axis2=-1))


def ofdm_linear_equalize(rg, sm, y, h_hat, err_var, no, kind):
    """OFDMEqualizer.call with the ZF / MF equaliser."""
    y_dt, hd, s = _ofdm_preprocess(rg, sm, y, h_hat, err_var, no)
    x_hat, no_eff = {"zf": zf_equalizer, "mf": mf_equalizer}[kind](y_dt, hd, s)
    B = y.shape[0]
    return _extract_data(rg, sm, x_hat, B).astype(np.complex64), _extract_data(rg, sm, no_eff, B).astype(np.float32)
