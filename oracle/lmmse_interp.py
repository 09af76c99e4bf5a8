"""CPU oracle of the LMMSE channel interpolator (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``LMMSEInterpolator`` of the reference (src/sionna/phy/ofdm/channel_estimation.py:1367-1853) with its two building
blocks ``LMMSEInterpolator1D`` (:736-1155) and ``SpatialChannelFilter`` (:1157-1365) and the covariance helpers
``tdl_freq_cov_mat`` / ``tdl_time_cov_mat`` (:1856-2070) in NumPy float64 / complex128.

The reference pads every row of the resource grid to the largest pilot count, adds the error variances to the padded pilot
covariance with a scatter and solves with ``tf.linalg.lstsq(fast=False)`` (minimum-norm: padded rows / columns come out
zero).  Here rows with the SAME pilot positions are grouped and solved without padding - the same numbers whenever the
pilot covariance plus error variance is non-singular (it is as soon as the noise variance is positive).

Pinned against the reference's own classes executed under the NumPy stand-in: tests/golden/lmmse_interp_ref_golden.npz
(tools/gen_lmmse_interp_ref_golden.py), tests/test_oracle_ref_exec_lmmse_interp.py."""
import json
import os

import numpy as np
from scipy.special import jv

C128 = np.complex128


def build_pilot_mask(mask, pilots):
    """_build_pilot_mask (:1704-1734): 0 data / unused, 1 pilot with energy, 2 zero-power pilot."""
    mask, pilots = np.asarray(mask), np.asarray(pilots)
    ntx, ns, T, F = mask.shape
    pm = np.zeros([ntx, ns, T, F], int)
    for tx in range(ntx):
        for st in range(ns):
            idx = np.flatnonzero(mask[tx, st].reshape(-1))
            vals = np.where(np.abs(pilots[tx, st, :len(idx)]) > 0.0, 1, 2)
            pm[tx, st].reshape(-1)[idx] = vals
    return pm


def inputs_to_rg(pilot_mask, h_pil):
    """scatter of the per-pilot inputs onto the resource grid (:1736-1761, 1793-1812): [..., tx, st, num_pilots] ->
    [..., tx, st, T, F]; zero-power pilots consume an input slot but are not written."""
    ntx, ns, T, F = pilot_mask.shape
    out = np.zeros(h_pil.shape[:-3] + (ntx, ns, T * F), h_pil.dtype)
    for tx in range(ntx):
        for st in range(ns):
            pos = np.flatnonzero(pilot_mask[tx, st].reshape(-1) != 0)
            keep = pilot_mask[tx, st].reshape(-1)[pos] == 1
            out[..., tx, st, pos[keep]] = h_pil[..., tx, st, :len(pos)][..., keep]
    return out.reshape(out.shape[:-1] + (T, F))


def update_pilot_mask(pm):
    """_update_pilot_mask_interp (:1763-1772): a row with at least one pilot is interpolated completely."""
    return np.where(np.any(pm == 1, axis=-1, keepdims=True), 1, pm)


def _rescale(h, err, hv, h_var):
    """the re-scaling of an intermediate step (:1129-1153, 1347-1363): makes the estimate conditionally unbiased for the next
    step; complex arithmetic like the reference, divide_no_nan."""
    den = hv + h_var - err
    s = np.where(den == 0, 0, 2. * h_var / np.where(den == 0, 1, den))
    h = s * h
    err = np.real(s * (s - 1.) * hv + (1. - s) * h_var + s * err)
    return h, np.maximum(err, 0.)


def interp_1d(h, err, pilot_mask, cov, last_step):
    """LMMSEInterpolator1D.__call__ (:972-1155): h, err [..., tx, st, O, I], pilot_mask [tx, st, O, I], cov [I, I]."""
    cov = np.asarray(cov, C128)
    ntx, ns, O, I = pilot_mask.shape
    h = np.asarray(h, C128)
    err = np.asarray(err, np.float64)
    h_var = np.diagonal(cov)
    out_h = np.zeros(h.shape, C128)
    out_e = np.broadcast_to(np.maximum(np.real(h_var), 0.), err.shape).copy()
    out_hv = np.zeros(h.shape, C128)                                     # variance of the estimate (rows without pilots: 0)
    groups = {}
    for tx in range(ntx):
        for st in range(ns):
            for o in range(O):
                groups.setdefault(tuple(np.flatnonzero(pilot_mask[tx, st, o] == 1)), []).append((tx, st, o))
    for pil, rows in groups.items():
        if not pil:
            continue
        p = np.asarray(pil)
        tx, st, o = (np.asarray(v) for v in zip(*rows))
        hp = h[..., tx, st, o, :][..., p]                                 # [..., G, np]
        ep = err[..., tx, st, o, :][..., p]
        a = cov[np.ix_(p, p)] + ep[..., :, None] * np.eye(len(p))        # (:1015-1033)
        b = cov[p, :]                                                     # [np, I]
        x = np.linalg.solve(a, np.broadcast_to(b, a.shape[:-2] + b.shape))
        ext = np.conj(np.swapaxes(x, -1, -2))                            # [..., G, I, np] (:1051-1055)
        hn = (ext @ hp[..., None])[..., 0]
        en = np.maximum(np.real(h_var - np.sum(ext * b.T, axis=-1)), 0.)  # (:1094-1106)
        hv = np.sum(ext * (np.conj(ext) @ cov[np.ix_(p, p)].T), axis=-1) + np.sum(ext * np.conj(ext) * ep[..., None, :], axis=-1)
        out_h[..., tx, st, o, :] = hn
        out_e[..., tx, st, o, :] = en
        out_hv[..., tx, st, o, :] = hv
    if not last_step:
        out_h, out_e = _rescale(out_h, out_e.astype(C128), out_hv, h_var)
    return out_h, out_e


def spatial_filter(h, err, cov, last_step):
    """SpatialChannelFilter.__call__ (:1252-1365): h, err [..., num_rx_ant] (the antenna dimension last), cov [ra, ra]."""
    cov = np.asarray(cov, C128)
    h = np.asarray(h, C128)
    err = np.asarray(err, np.float64)
    ra = cov.shape[0]
    a = cov + err[..., :, None] * np.eye(ra)
    w = np.conj(np.swapaxes(np.linalg.solve(a, np.broadcast_to(cov, a.shape)), -1, -2))     # (A^-1 C)^H (:1301-1306)
    hn = (w @ h[..., None])[..., 0]
    h_var = np.diagonal(cov)
    en = np.maximum(np.real(h_var - np.sum(cov.T * w, axis=-1)), 0.)
    if not last_step:
        hv = np.sum(w * (np.conj(w) @ cov.T), axis=-1) + np.sum(w * np.conj(w) * err[..., None, :], axis=-1)
        hn, en = _rescale(hn, en.astype(C128), hv, h_var)
    return hn, en


def lmmse_interpolate(mask, pilots, h_pil, err_pil, cov_time, cov_freq, cov_space=None, order="t-f"):
    """LMMSEInterpolator(...)(h_hat, err_var) (:1623-1853): inputs at the pilots [B, rx, ra, tx, st, num_pilots] ->
    (h_hat, err_var) [B, rx, ra, tx, st, T, F]."""
    order = order.split("-")
    pm = build_pilot_mask(mask, pilots)
    h = inputs_to_rg(pm, np.asarray(h_pil, C128))
    err = inputs_to_rg(pm, np.broadcast_to(np.asarray(err_pil, np.float64), np.asarray(h_pil).shape))
    for i, o in enumerate(order):
        last = i == len(order) - 1
        if o == "f":
            h, err = interp_1d(h, err, pm, cov_freq, last)
            pm = update_pilot_mask(pm)
        elif o == "t":
            pmt = np.swapaxes(pm, -1, -2)
            ht, et = interp_1d(np.swapaxes(h, -1, -2), np.swapaxes(err, -1, -2), pmt, cov_time, last)
            h, err = np.swapaxes(ht, -1, -2), np.swapaxes(et, -1, -2)
            pm = np.swapaxes(update_pilot_mask(pmt), -1, -2)
        else:
            hs, es = spatial_filter(np.moveaxis(h, 2, -1), np.moveaxis(err, 2, -1), cov_space, last)
            h, err = np.moveaxis(hs, -1, 2), np.moveaxis(es, -1, 2)
        err = err * (pm == 1)
    return h, err


_TDL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sionna_amd", "phy", "channel", "tr38901", "tdl_models.json")


def _tdl_params(model):
    """the power delay profile tables of TR 38.901 (the package's copy of the reference's TDL-*.json values; checked against
    the reference objects in tests/test_oracle_ref_exec_tdl.py)"""
    with open(_TDL) as f:
        return json.load(f)[model]


def tdl_freq_cov_mat(model, subcarrier_spacing, fft_size, delay_spread):
    """:1856-1953: R_f[u, v] = sum_l P_l exp(-j 2 pi tau_l df (u - v)); the LoS models merge their first two taps."""
    p = _tdl_params(model)
    delays = np.asarray(p["delays"], float) * delay_spread
    pw = np.power(10.0, np.asarray(p["powers"], float) / 10.0)
    if p["los"]:
        pw = np.concatenate([[pw[0] + pw[1]], pw[2:]])
        delays = delays[1:]
    pw = pw / pw.sum()
    ph = np.exp(1j * (-2. * np.pi * subcarrier_spacing * np.arange(fft_size))[None, :] * delays[:, None])     # [L, M]
    return np.einsum("l,lu,lv->uv", pw, ph, np.conj(ph))


def tdl_time_cov_mat(model, speed, carrier_frequency, ofdm_symbol_duration, num_ofdm_symbols, los_angle_of_arrival=np.pi / 4.):
    """:1956-2070: Jakes' J0 of the Doppler spread (+ the specular term of the LoS models)."""
    p = _tdl_params(model)
    pw = np.power(10.0, np.asarray(p["powers"], float) / 10.0)
    pw = pw / pw.sum()
    nu = 2. * np.pi * speed / 299792458. * carrier_frequency
    i = np.arange(num_ofdm_symbols)
    e = nu * ofdm_symbol_duration * (i[:, None] - i[None, :])
    if p["los"]:
        return jv(0.0, e) * pw[1:].sum() + np.exp(1j * e * np.cos(los_angle_of_arrival)) * pw[0]
    return (jv(0.0, e) * pw.sum()).astype(C128)
