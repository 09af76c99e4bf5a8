"""TEST INFRASTRUCTURE (not part of the product): float64 / complex128 restatement of the OFDM link blocks of config C4 -
the specification of ``precision="double"`` (reference src/sionna/phy/block.py:25-52) for the kernels of
sionna_amd/csrc/f64_ofdm.hip and the double-precision OFDM (de)modulator of csrc/ofdm_time.hip.

The formulas are those of oracle/ofdm.py and oracle/utils.py (which cite the reference lines and are pinned to the
executed reference in float32: tests/test_oracle_ref_exec_ofdm_rx.py) with every intermediate kept in double; the random
draws use the SAME Philox words and the same 24-bit uniforms as the float32 stream (exact in double), so a double run is
the float32 run's realisation up to float32 rounding - checked in tests/test_oracle_pins.py.  Parity status: the
float64 path has no executed-reference fixture of its own (the reference's float64 differs from its float32 only in
rounding); it is pinned through its float32 twin.
"""
import numpy as np

from . import ofdm as o32
from . import utils as outil

PI = np.pi


def _u01(w):
    return (w >> np.uint32(8)).astype(np.float64) * 2.0 ** -24 + 2.0 ** -25


def _u(seed, call, n, lo, hi):
    nb = (n + 3) // 4
    w = np.stack(outil.philox_block(seed, call, nb), axis=1).reshape(-1)[:n]
    return lo + (hi - lo) * _u01(w)


def complex_normal(seed, call, n, var=1.0):
    """utils/misc.py:19-54 on the stream layout of oracle/utils.py::complex_normal"""
    nb = (n + 1) // 2
    w0, w1, w2, w3 = outil.philox_block(seed, call, nb)
    ua = np.stack([_u01(w0), _u01(w2)], axis=1).reshape(-1)[:n]
    ub = np.stack([_u01(w1), _u01(w3)], axis=1).reshape(-1)[:n]
    r = np.sqrt(-2.0 * np.log(ua))
    t = 6.283185307179586 * ub
    s = np.sqrt(0.5) * np.sqrt(var)
    return (r * np.cos(t)) * s + 1j * ((r * np.sin(t)) * s)


def awgn(x, no, seed, call):
    """channel/awgn.py:63-78"""
    x = np.asarray(x, np.complex128)
    no = np.broadcast_to(np.asarray(no, np.float64), x.shape).reshape(-1)
    w = complex_normal(seed, call, x.size, 1.0)
    return (x.reshape(-1) + w * np.sqrt(no)).reshape(x.shape)


def rg_map(rg, x):
    """ofdm/resource_grid.py:394-412"""
    t = rg.build_type_grid()
    out = np.zeros((x.shape[0],) + t.shape, np.complex128)
    out[:, t == 1] = np.asarray(rg.pilot_pattern.pilots, np.complex128).reshape(-1)
    out[:, t == 0] = x.reshape(x.shape[0], -1)
    return out


def subcarrier_frequencies(num_subcarriers, subcarrier_spacing):
    start = -(num_subcarriers // 2)
    limit = num_subcarriers // 2 if num_subcarriers % 2 == 0 else num_subcarriers // 2 + 1
    return np.arange(start, limit, dtype=np.float64) * float(subcarrier_spacing)


def tdl_cir(seed, call, batch, num_time_steps, sampling_frequency, delays_s, mean_powers, min_doppler, max_doppler,
            num_rx_ant=1, num_tx_ant=1, num_sinusoids=20, los_power=None, los_aoa=PI / 4):
    """channel/tr38901/tdl.py:372-470, stream layout of oracle/ofdm.py::tdl_cir"""
    P, N, T = len(mean_powers), num_sinusoids, num_time_steps
    t = np.arange(T, dtype=np.float64) / float(sampling_frequency)
    doppler = _u(seed, call, batch, min_doppler, max_doppler).reshape(batch, 1, 1, 1, 1, 1)
    theta = _u(seed, call + 1, batch * P * N, -PI / N, PI / N).reshape(batch, 1, 1, P, 1, N)
    phi = _u(seed, call + 2, batch * num_rx_ant * num_tx_ant * P * N, -PI, PI).reshape(batch, num_rx_ant, num_tx_ant, P, 1, N)
    alpha = (2 * PI / N) * np.arange(1, N + 1, dtype=np.float64).reshape(1, 1, 1, 1, 1, N) + theta
    arg = doppler * t.reshape(1, 1, 1, 1, T, 1) * np.cos(alpha) + phi
    h = (np.cos(arg) + 1j * np.sin(arg)).sum(-1) * (1 / np.sqrt(N))
    h = np.sqrt(np.asarray(mean_powers, np.float64)).reshape(1, 1, 1, P, 1) * h
    if los_power is not None:
        phi0 = _u(seed, call + 3, batch, -PI, PI).reshape(batch, 1, 1, 1)
        arg0 = doppler.reshape(batch, 1, 1, 1) * t.reshape(1, 1, 1, T) * np.cos(los_aoa) + phi0
        h[:, :, :, 0, :] += (np.cos(arg0) + 1j * np.sin(arg0)) * np.sqrt(los_power)
    a = h[:, None, :, None, :, :, :]
    tau = np.tile(np.asarray(delays_s, np.float64).reshape(1, 1, 1, P), [batch, 1, 1, 1])
    return a, tau


def cir_to_ofdm_channel(frequencies, a, tau, normalize=False):
    """channel/utils.py:180-253"""
    tau = tau[:, :, None, :, None, :, None, None]
    e = np.exp(-2j * PI * np.asarray(frequencies, np.float64) * tau)
    h_f = np.sum(a[..., None] * e, axis=-3)
    if normalize:
        c = np.mean(np.abs(h_f) ** 2, axis=(2, 4, 5, 6), keepdims=True)
        h_f = np.where(c > 0, h_f / np.sqrt(np.where(c > 0, c, 1)), 0)
    return h_f


def apply_ofdm_channel(x, h_freq):
    """channel/apply_ofdm_channel.py:70-80"""
    return np.sum(h_freq * x[:, None, None], axis=(3, 4))


def ls_estimate(rg, y, no, interpolation="nn"):
    """ofdm/channel_estimation.py:138-173, 257-285, 364-435"""
    pp = rg.pilot_pattern
    y_eff = o32.remove_nulled(rg, y)
    y_flat = y_eff.reshape(y_eff.shape[:-2] + (-1,))
    m = pp.mask.reshape(pp.mask.shape[:2] + (-1,))
    pilot_ind = np.argsort(~m, axis=-1, kind="stable")[..., :pp.num_pilot_symbols]
    y_p = y_flat[..., pilot_ind]
    pil = np.asarray(pp.pilots, np.complex128)
    with np.errstate(divide="ignore", invalid="ignore"):
        h_ls = np.where(pil != 0, y_p / np.where(pil != 0, pil, 1), 0)
        no_ = np.asarray(no, np.float64)
        no_ = no_.reshape(no_.shape + (1,) * (h_ls.ndim - no_.ndim))
        ev = np.where(pil != 0, no_ / np.where(pil != 0, np.abs(pil) ** 2, 1), 0)
    if ev.ndim < h_ls.ndim:
        ev = ev[None, None, None]
    if interpolation is None:
        return h_ls, ev
    g = o32.nn_gather_ind(pp)
    tx, s = np.indices(g.shape)[:2]
    return h_ls[:, :, :, tx, s, g], np.maximum(ev[:, :, :, tx, s, g], 0)


def ls_estimate_lin(rg, y, no, time_avg=False):
    """LSChannelEstimator(interpolation_type="lin" / "lin_time_avg") (ofdm/channel_estimation.py:138-173, 437-733): the float64
    interior of oracle/ofdm.py::LinearInterpolator without its casts to single precision"""
    h_p, ev_p = ls_estimate(rg, y, no, interpolation=None)
    li = o32.LinearInterpolator(rg.pilot_pattern, time_avg)
    h = li._interp(h_p)
    ev = np.real(li._interp(np.broadcast_to(ev_p, h_p.shape).astype(np.complex128)))
    return h, np.maximum(ev, 0)


def ofdm_modulate(x, cyclic_prefix_length):
    """ofdm/modulator.py:97-124"""
    n = x.shape[-1]
    xt = np.fft.ifft(np.fft.ifftshift(np.asarray(x, np.complex128), axes=-1), axis=-1) * np.sqrt(n)
    cp = o32._cp_vector(cyclic_prefix_length, x.shape[-2])
    parts = [np.concatenate([xt[..., s, n - int(cp[s]):], xt[..., s, :]], axis=-1) for s in range(x.shape[-2])]
    return np.concatenate(parts, axis=-1)


def ofdm_demodulate(y, fft_size, l_min, cyclic_prefix_length, num_ofdm_symbols=None):
    """ofdm/demodulator.py:143-203"""
    n = fft_size
    cp0 = np.asarray(cyclic_prefix_length, np.int64)
    if cp0.ndim == 0:
        num_ofdm_symbols = y.shape[-1] // (n + int(cp0))
    cp = o32._cp_vector(cyclic_prefix_length, num_ofdm_symbols)
    off = np.concatenate([[0], np.cumsum(cp + n)[:-1]])
    rows = np.stack([y[..., int(off[s] + cp[s]):int(off[s] + cp[s]) + n] for s in range(len(cp))], axis=-2)
    xf = np.fft.fft(np.asarray(rows, np.complex128), axis=-1) / np.sqrt(n)
    xf = xf * np.exp(1j * ((-2 * PI * l_min) / n * np.arange(n, dtype=np.float64)))
    return np.fft.fftshift(xf, axes=-1)


def cir_to_time_channel(bandwidth, a, tau, l_min, l_max, normalize=False):
    """channel/utils.py:256-349: a [B,rx,ra,tx,ta,P,T], tau [B,rx,tx,P] -> h [B,rx,ra,tx,ta,T,L]"""
    lags = np.arange(l_min, l_max + 1, dtype=np.float64)
    g = np.sinc(lags - np.asarray(tau, np.float64)[..., None] * float(bandwidth))                 # [B,rx,tx,P,L]
    h = np.einsum("brxtypn,brtpl->brxtynl", np.asarray(a, np.complex128), g)
    if normalize:
        c = np.mean(np.sum(np.abs(h) ** 2, axis=-1), axis=(2, 4, 5), keepdims=True)[..., None]
        h = np.where(c > 0, h / np.sqrt(np.where(c > 0, c, 1)), 0)
    return h


def apply_time_channel(x, h_time):
    """channel/apply_time_channel.py:85-137: x [B,tx,ta,Tn], h [B,rx,ra,tx,ta,Tn+L-1,L] -> y [B,rx,ra,Tn+L-1]"""
    x, h = np.asarray(x, np.complex128), np.asarray(h_time, np.complex128)
    tn, l_tot = x.shape[-1], h.shape[-1]
    tout = tn + l_tot - 1
    y = np.zeros(h.shape[:3] + (tout,), np.complex128)
    for l in range(l_tot):
        xs = np.zeros(x.shape[:-1] + (tout,), np.complex128)
        xs[..., l:l + tn] = x                                                                     # x[t - l]
        y += np.sum(h[..., l] * xs[:, None, None], axis=(3, 4))
    return y
