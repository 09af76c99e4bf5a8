"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement (NumPy, float64) of the clustered-delay-line channel model of 3GPP TR 38.901
Sec. 7.7.1 as implemented by the reference:
  CDL                          /root/reference/src/sionna/phy/channel/tr38901/cdl.py:22-695
  ChannelCoefficientsGenerator /root/reference/src/sionna/phy/channel/tr38901/channel_coefficients.py:15-1031
                               (TR 38.901 Sec. 7.5 steps 10 and 11, no sub-clustering)
  AntennaElement / AntennaPanel / PanelArray / Antenna / AntennaArray
                               /root/reference/src/sionna/phy/channel/tr38901/antenna.py:14-743

The reference draws the random ray coupling, the initial phases and the velocity vector from
TensorFlow's RNG; this build defines its own counter-based streams instead ("parity unpinned" for
the realisations; the deterministic part - tables, fields, array responses, Doppler, LoS / K-factor
combination - is pinned by the physical checks of tests/test_oracle_cdl.py, and the generator as a
whole STATISTICALLY against the reference's own cdl.py / rays.py / channel_coefficients.py / antenna.py
executed under tools/ref_exec: tests/test_oracle_ref_exec_cdl.py):
  (seed, call+0..2)  speed v_r, azimuth v_phi, zenith v_theta     one uniform per batch example
  (seed, call+3..6)  sort keys of the AoA, AoD, ZoA, ZoD shuffles  u32 per (b, cluster, ray)
  (seed, call+7)     initial phases Phi in (-pi, pi)               per (b, cluster, ray, 4)
"""
import json
import os

import numpy as np

from . import utils as outil
from .ofdm import _u

PI = np.pi
SPEED_OF_LIGHT = 299792458.0
NUM_RAYS = 20
_RAY_OFFSETS = np.array([0.0447, -0.0447, 0.1413, -0.1413, 0.2492, -0.2492, 0.3715, -0.3715, 0.5129, -0.5129, 0.6797,
                         -0.6797, 0.8844, -0.8844, 1.1481, -1.1481, 1.5195, -1.5195, 2.1551, -2.1551])   # TR 38.901 Tab. 7.5-3
_MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sionna_amd", "phy", "channel",
                       "tr38901", "cdl_models.json")


# ------------------------------------------------------------------ antennas (antenna.py)
def radiation_pattern(pattern, theta, phi):
    """antenna.py:115-147 (TR 38.901 Table 7.3-1), linear power gain."""
    if pattern == "omni":
        return np.ones_like(theta)
    th3 = ph3 = 65 / 180 * PI
    a_v = -np.minimum(12 * ((theta - PI / 2) / th3) ** 2, 30)
    a_h = -np.minimum(12 * (phi / ph3) ** 2, 30)
    a_db = -np.minimum(-(a_v + a_h), 30) + 8
    return 10 ** (a_db / 10)


def element_field(pattern, slant, theta, phi):
    """antenna.py:53-68 (7.3-4/5): (F_theta, F_phi) of an element with polarisation slant angle."""
    a = np.sqrt(radiation_pattern(pattern, theta, phi))
    return a * np.cos(slant), a * np.sin(slant)


class PanelArray:
    """antenna.py:281-655: geometry (metres, LCS: panel in the y-z plane) and polarisation of every
    element.  ``pol`` [num_ant] = 0 / 1 selects the slant angle ``slants[pol]``."""

    def __init__(self, num_rows_per_panel, num_cols_per_panel, polarization, polarization_type, antenna_pattern,
                 carrier_frequency, num_rows=1, num_cols=1, panel_vertical_spacing=None, panel_horizontal_spacing=None,
                 element_vertical_spacing=None, element_horizontal_spacing=None):
        assert polarization in ("single", "dual")
        ev = 0.5 if element_vertical_spacing is None else element_vertical_spacing
        eh = 0.5 if element_horizontal_spacing is None else element_horizontal_spacing
        pv = (num_rows_per_panel - 1) * ev + 0.5 if panel_vertical_spacing is None else panel_vertical_spacing
        ph = (num_cols_per_panel - 1) * eh + 0.5 if panel_horizontal_spacing is None else panel_horizontal_spacing
        p = 1 if polarization == "single" else 2
        if polarization == "single":
            assert polarization_type in ("V", "H")
            self.slants = [0.0 if polarization_type == "V" else PI / 2]
        else:
            assert polarization_type in ("VH", "cross")
            s0 = 0.0 if polarization_type == "VH" else -PI / 4
            self.slants = [s0, s0 + PI / 2]
        self.polarization, self.pattern = polarization, antenna_pattern
        ne = num_rows_per_panel * num_cols_per_panel
        panel = np.zeros((ne * p, 3))
        for i in range(num_rows_per_panel):
            for j in range(num_cols_per_panel):
                panel[i + j * num_rows_per_panel] = [0, j * eh, -i * ev]
        panel[:ne] += [0, -(num_cols_per_panel - 1) * eh / 2, (num_rows_per_panel - 1) * ev / 2]
        if p == 2:
            panel[ne:] = panel[:ne]
        pos, pol = [], []
        for j in range(num_cols):
            for i in range(num_rows):
                pos.append(panel + [0, j * ph, -i * pv])
                pol.append(np.repeat(np.arange(p), ne))
        pos = np.concatenate(pos) + [0, -(num_cols - 1) * ph / 2, (num_rows - 1) * pv / 2]
        self.ant_pos = pos * (SPEED_OF_LIGHT / carrier_frequency)
        self.pol = np.concatenate(pol)
        self.num_ant = len(self.pol)


class AntennaArray(PanelArray):
    """antenna.py:690-743"""

    def __init__(self, num_rows, num_cols, polarization, polarization_type, antenna_pattern, carrier_frequency,
                 vertical_spacing=None, horizontal_spacing=None):
        super().__init__(num_rows, num_cols, polarization, polarization_type, antenna_pattern, carrier_frequency,
                         element_vertical_spacing=vertical_spacing, element_horizontal_spacing=horizontal_spacing)


class Antenna(PanelArray):
    """antenna.py:657-688"""

    def __init__(self, polarization, polarization_type, antenna_pattern, carrier_frequency):
        super().__init__(1, 1, polarization, polarization_type, antenna_pattern, carrier_frequency)


# ------------------------------------------------------------------ geometry (channel_coefficients.py:196-393)
def unit_vector(theta, phi):
    return np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=-1)


def rotation_matrix(o):
    """:219-248 (7.1-4): orientation (bearing alpha, downtilt beta, slant gamma), LCS -> GCS."""
    a, b, c = o
    return np.array([[np.cos(a) * np.cos(b), np.cos(a) * np.sin(b) * np.sin(c) - np.sin(a) * np.cos(c),
                      np.cos(a) * np.sin(b) * np.cos(c) + np.sin(a) * np.sin(c)],
                     [np.sin(a) * np.cos(b), np.sin(a) * np.sin(b) * np.sin(c) + np.cos(a) * np.cos(c),
                      np.sin(a) * np.sin(b) * np.cos(c) - np.cos(a) * np.sin(c)],
                     [-np.sin(b), np.cos(b) * np.sin(c), np.cos(b) * np.cos(c)]])


def gcs_to_lcs(o, theta, phi):
    """:288-331 (7.1-7/8)"""
    rho = unit_vector(theta, phi) @ rotation_matrix(o)            # R^T rho, row-vector form
    return np.arccos(np.clip(rho[..., 2], -1, 1)), np.arctan2(rho[..., 1], rho[..., 0])


def psi_angle(o, theta, phi):
    """:333-362 (7.1-15)"""
    a, b, c = o
    re = np.sin(c) * np.cos(theta) * np.sin(phi - a) + np.cos(c) * (np.cos(b) * np.sin(theta) - np.sin(b) * np.cos(theta) * np.cos(phi - a))
    im = np.sin(c) * np.cos(phi - a) + np.sin(b) * np.cos(c) * np.sin(phi - a)
    return np.arctan2(im, re)


def field_gcs(arr, o, theta, phi):
    """(F_theta, F_phi) in the GCS of both polarisations of ``arr`` for rays (theta, phi) in the GCS
    (:659-700 + :364-393, 7.1-11): -> [..., num_pol, 2]."""
    tp, pp = gcs_to_lcs(o, theta, phi)
    psi = psi_angle(o, theta, phi)
    out = []
    for slant in arr.slants:
        ft, fp = element_field(arr.pattern, slant, tp, pp)
        out.append(np.stack([np.cos(psi) * ft - np.sin(psi) * fp, np.sin(psi) * ft + np.cos(psi) * fp], axis=-1))
    return np.stack(out, axis=-2)


# ------------------------------------------------------------------ CDL model (cdl.py)
class CDL:
    """cdl.py:187-555 (parameters) and :258-333 + channel_coefficients.py:173-194, 459-1031 (sampling)."""

    def __init__(self, model, delay_spread, carrier_frequency, ut_array, bs_array, direction, ut_orientation=None,
                 bs_orientation=None, min_speed=0., max_speed=None):
        assert direction in ("uplink", "downlink") and model in "ABCDE"
        with open(_MODELS) as f:
            p = json.load(f)[model]
        ut_o = np.array([PI, 0., 0.]) if ut_orientation is None else np.asarray(ut_orientation, float)
        bs_o = np.zeros(3) if bs_orientation is None else np.asarray(bs_orientation, float)
        self.lambda_0 = SPEED_OF_LIGHT / carrier_frequency
        self.delay_spread, self.min_speed = delay_spread, min_speed
        self.max_speed = min_speed if max_speed is None else max_speed
        self.los = bool(p["los"])
        powers = 10 ** (np.array(p["powers"]) / 10)
        powers = powers / powers.sum()
        delays = np.array(p["delays"])
        ang = {k: np.array(p[k]) for k in ("aod", "aoa", "zod", "zoa")}
        if self.los:
            los_power, powers, delays = powers[0], powers[1:], delays[1:]
            los_ang = {k: np.deg2rad(v[0]) for k, v in ang.items()}
            ang = {k: v[1:] for k, v in ang.items()}
            norm = powers.sum()
            powers = powers / norm
            self.k_factor = los_power / norm
        else:
            self.k_factor, los_ang = 1.0, dict(aod=0., aoa=0., zod=0., zoa=0.)
        c = dict(aod=p["cASD"], aoa=p["cASA"], zod=p["cZSD"], zoa=p["cZSA"])
        rays = {k: np.deg2rad(ang[k][:, None] + c[k] * _RAY_OFFSETS[None, :]) for k in ang}      # [N, 20]
        if direction == "downlink":                    # BS transmits: departure angles at the BS
            self.tx_array, self.rx_array, self.tx_o, self.rx_o = bs_array, ut_array, bs_o, ut_o
            self.moving_end = "rx"
        else:                                          # uplink: the tables' departure side is the receiver
            self.tx_array, self.rx_array, self.tx_o, self.rx_o = ut_array, bs_array, ut_o, bs_o
            self.moving_end = "tx"
            rays = dict(aod=rays["aoa"], aoa=rays["aod"], zod=rays["zoa"], zoa=rays["zod"])
            los_ang = dict(aod=los_ang["aoa"], aoa=los_ang["aod"], zod=los_ang["zoa"], zoa=los_ang["zod"])
        self.rays, self.los_ang = rays, los_ang
        self.powers, self.delays = powers, delays
        self.num_clusters = len(powers)
        self.xpr = 10 ** (p["xpr"] / 10)
        self.order = np.argsort(delays, kind="stable")                 # channel_coefficients.py:905-913

    def draw(self, seed, call, batch, precision="single"):
        """The random quantities of one call (stream layout: module docstring).  precision="double": the same 24-bit uniforms
        mapped to their intervals in float64 (oracle/f64_ofdm.py::_u) instead of float32."""
        N, M = self.num_clusters, NUM_RAYS
        if precision == "double":
            from .f64_ofdm import _u as _ud
        else:
            _ud = _u
        v_r = _ud(seed, call, batch, self.min_speed, self.max_speed)
        v_phi = _ud(seed, call + 1, batch, 0.0, 2 * PI)
        v_theta = _ud(seed, call + 2, batch, 0.0, PI)
        vel = np.stack([v_r * np.cos(v_phi) * np.sin(v_theta), v_r * np.sin(v_phi) * np.sin(v_theta), v_r * np.cos(v_theta)],
                       axis=-1).astype(np.float64)
        perms = []
        for i in range(4):                               # aoa, aod, zoa, zod
            nb = (batch * N * M + 3) // 4
            keys = np.stack(outil.philox_block(seed, call + 3 + i, nb), axis=1).reshape(-1)[:batch * N * M]
            perms.append(np.argsort(keys.reshape(batch, N, M), axis=-1, kind="stable"))
        phases = _ud(seed, call + 7, batch * N * M * 4, -PI, PI).reshape(batch, N, M, 4).astype(np.float64)
        return vel, dict(aoa=perms[0], aod=perms[1], zoa=perms[2], zod=perms[3]), phases

    def _link(self, aoa, aod, zoa, zod, pm, vel, t):
        """sum over the last ray axis of field * array * doppler (channel_coefficients.py:786-828):
        angles [..., R], pm [..., R, 2, 2], vel [..., 3] -> [..., U, S, T]."""
        frx = field_gcs(self.rx_array, self.rx_o, zoa, aoa)[..., self.rx_array.pol, :]          # [..., R, U, 2]
        ftx = field_gcs(self.tx_array, self.tx_o, zod, aod)[..., self.tx_array.pol, :]          # [..., R, S, 2]
        field = np.einsum("...ua,...ab,...sb->...us", frx, pm, ftx)                              # [..., R, U, S]
        r_rx, r_tx = unit_vector(zoa, aoa), unit_vector(zod, aod)
        d_rx = self.rx_array.ant_pos @ rotation_matrix(self.rx_o).T                              # GCS positions
        d_tx = self.tx_array.ant_pos @ rotation_matrix(self.tx_o).T
        a_rx = np.exp(2j * PI / self.lambda_0 * (r_rx @ d_rx.T))                                 # [..., R, U]
        a_tx = np.exp(2j * PI / self.lambda_0 * (r_tx @ d_tx.T))
        w = 2 * PI / self.lambda_0 * np.sum(r_rx * vel[..., None, :], axis=-1)                    # [..., R] (arrival side, :517-573)
        dop = np.exp(1j * w[..., None] * t)                                                       # [..., R, T]
        return np.einsum("...rus,...ru,...rs,...rt->...ust", field, a_rx, a_tx, dop)

    def __call__(self, seed, call, batch, num_time_steps, sampling_frequency, precision="single"):
        """-> a [B,1,U,1,S,N,T] complex64, tau [B,1,1,N] float32 (cdl.py:258-333); precision="double": complex128 / float64
        without the final casts and with the draws mapped in float64."""
        N = self.num_clusters
        vel, perm, phi = self.draw(seed, call, batch, precision)
        t = np.arange(num_time_steps) / sampling_frequency
        ang = {k: np.take_along_axis(np.broadcast_to(self.rays[k], (batch, N, NUM_RAYS)), perm[k], axis=-1) for k in perm}
        k = np.sqrt(1 / self.xpr)
        e = np.exp(1j * phi)
        pm = np.stack([np.stack([e[..., 0], k * e[..., 1]], -1), np.stack([k * e[..., 2], e[..., 3]], -1)], -2)   # :482-515
        h = self._link(ang["aoa"], ang["aod"], ang["zoa"], ang["zod"], pm, vel[:, None, :], t)    # [B,N,U,S,T]
        h = h * np.sqrt(self.powers / NUM_RAYS)[None, :, None, None, None]
        h = h[:, self.order]
        delays = (self.delays * self.delay_spread)[self.order]
        if self.los:                                                                               # :919-1031
            la = {k2: np.full((batch, 1), v) for k2, v in self.los_ang.items()}
            pm_los = np.broadcast_to(np.array([[1., 0.], [0., -1.]], complex), (batch, 1, 2, 2))
            h_los = self._link(la["aoa"], la["aod"], la["zoa"], la["zod"], pm_los, vel, t)         # [B,U,S,T]
            kf = self.k_factor
            h = h * np.sqrt(1 / (kf + 1))
            h[:, 0] += h_los * np.sqrt(kf / (kf + 1))
        a = np.transpose(h, [0, 2, 3, 1, 4])[:, None, :, None]                                     # [B,1,U,1,S,N,T]
        tau = np.broadcast_to(delays[None, None, None, :], (batch, 1, 1, N))
        if precision == "double":
            return np.ascontiguousarray(a), np.ascontiguousarray(tau, dtype=np.float64)
        return a.astype(np.complex64), tau.astype(np.float32)
