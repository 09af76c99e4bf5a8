"""Clustered delay line (CDL) channel model of 3GPP TR 38.901 Sec. 7.7.1 - mirror of
``sionna.phy.channel.tr38901.CDL`` (reference src/sionna/phy/channel/tr38901/cdl.py:22-695, with the
coefficient generation of channel_coefficients.py, TR 38.901 Sec. 7.5 steps 10-11).
``CDL(model, delay_spread, carrier_frequency, ut_array, bs_array, direction, ...)(batch_size,
num_time_steps, sampling_frequency) -> (a, tau)``.

Host side (NumPy, once per instance): cluster tables, ray angles, and - for all 20 x 20 (zenith ray,
azimuth ray) pairs of every cluster and both link ends - the GCS field patterns of both polarisations,
the array responses of every antenna and the arrival unit vectors.  The random ray coupling, initial
phases, Doppler rotation and the sum over the rays run in ``samd_cdl_cir_c64``."""
import json
import os

import numpy as np
import torch

from .... import _ffi
from ...block import Object, wrap
from ...config import config
from .antenna import PanelArray

PI = np.pi
SPEED_OF_LIGHT = 299792458.0
_MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cdl_models.json")
# ray offset angles within a cluster for 1 degree RMS angle spread, TR 38.901 Table 7.5-3
_RAY_OFFSETS = np.array([0.0447, -0.0447, 0.1413, -0.1413, 0.2492, -0.2492, 0.3715, -0.3715, 0.5129, -0.5129, 0.6797,
                         -0.6797, 0.8844, -0.8844, 1.1481, -1.1481, 1.5195, -1.5195, 2.1551, -2.1551])


def _rot(o):
    """LCS -> GCS rotation for (bearing, downtilt, slant), TR 38.901 (7.1-4)."""
    ca, sa, cb, sb, cc, sc = np.cos(o[0]), np.sin(o[0]), np.cos(o[1]), np.sin(o[1]), np.cos(o[2]), np.sin(o[2])
    return np.array([[ca * cb, ca * sb * sc - sa * cc, ca * sb * cc + sa * sc],
                     [sa * cb, sa * sb * sc + ca * cc, sa * sb * cc - ca * sc],
                     [-sb, cb * sc, cb * cc]])


def _side_tables(arr, o, zen, az, lam):
    """Per-ray-pair tables of one link end: zen / az [N, R] ray angles [rad] in the GCS ->
    fields [N,R,R,2,2], array responses [N,R,R,num_ant] complex, unit vectors [N,R,R,3]."""
    th = np.broadcast_to(zen[:, :, None], zen.shape + (az.shape[1],))
    ph = np.broadcast_to(az[:, None, :], th.shape)
    rho = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], axis=-1)
    r = _rot(o)
    loc = rho @ r                                              # R^T rho: direction in the LCS
    th_l, ph_l = np.arccos(np.clip(loc[..., 2], -1, 1)), np.arctan2(loc[..., 1], loc[..., 0])
    a, b, c = o
    psi = np.arctan2(np.sin(c) * np.cos(ph - a) + np.sin(b) * np.cos(c) * np.sin(ph - a),
                     np.sin(c) * np.cos(th) * np.sin(ph - a) + np.cos(c) * (np.cos(b) * np.sin(th) - np.sin(b) * np.cos(th) * np.cos(ph - a)))
    fields = np.zeros(th.shape + (2, 2))
    for k, el in enumerate([arr.ant_pol1] + ([arr.ant_pol2] if arr.polarization == "dual" else [])):
        ft, fp = el.field(th_l, ph_l)
        fields[..., k, 0] = np.cos(psi) * ft - np.sin(psi) * fp          # (7.1-11)
        fields[..., k, 1] = np.sin(psi) * ft + np.cos(psi) * fp
    pos = arr.ant_pos @ r.T                                     # GCS antenna positions
    resp = np.exp(2j * PI / lam * (rho @ pos.T))
    return fields, resp, rho


class CDL(Object):
    def __init__(self, model, delay_spread, carrier_frequency, ut_array, bs_array, direction, ut_orientation=None,
                 bs_orientation=None, min_speed=0., max_speed=None, precision=None):
        super().__init__(precision=precision)
        assert direction in ("uplink", "downlink"), "Invalid link direction"
        assert model in ("A", "B", "C", "D", "E"), "Invalid CDL model"
        assert isinstance(ut_array, PanelArray) and isinstance(bs_array, PanelArray), "arrays must be PanelArray instances"
        self._direction = direction
        ut_o = np.array([PI, 0.0, 0.0]) if ut_orientation is None else np.asarray(ut_orientation, np.float64).reshape(3)
        bs_o = np.zeros(3) if bs_orientation is None else np.asarray(bs_orientation, np.float64).reshape(3)
        if direction == "downlink":
            self._moving_end, self._tx_array, self._rx_array, self._tx_o, self._rx_o = "rx", bs_array, ut_array, bs_o, ut_o
        else:
            self._moving_end, self._tx_array, self._rx_array, self._tx_o, self._rx_o = "tx", ut_array, bs_array, ut_o, bs_o
        self._carrier_frequency = float(carrier_frequency)
        self._delay_spread = float(delay_spread)
        self._min_speed = float(min_speed)
        if max_speed is None:
            self._max_speed = self._min_speed
        else:
            assert max_speed >= min_speed, "min_speed cannot be larger than max_speed"
            self._max_speed = float(max_speed)
        self._load_parameters(model)
        self._dev = None
        self._ws = _ffi.Workspace()

    NUM_RAYS = 20

    def _load_parameters(self, model):
        with open(_MODELS) as f:
            p = json.load(f)[model]
        self._los = bool(p["los"])
        powers = np.power(10.0, np.asarray(p["powers"]) / 10.0)
        powers = powers / powers.sum()
        delays = np.asarray(p["delays"], np.float64)
        centre = {"aod": np.asarray(p["aod"]), "aoa": np.asarray(p["aoa"]), "zod": np.asarray(p["zod"]), "zoa": np.asarray(p["zoa"])}
        spread = {"aod": p["cASD"], "aoa": p["cASA"], "zod": p["cZSD"], "zoa": p["cZSA"]}
        los_ang = dict.fromkeys(centre, 0.0)
        self._k_factor = 1.0
        if self._los:                                           # first row = specular path (cdl.py:422-445)
            los_power, powers, delays = powers[0], powers[1:], delays[1:]
            los_ang = {k: np.deg2rad(v[0]) for k, v in centre.items()}
            centre = {k: v[1:] for k, v in centre.items()}
            norm = powers.sum()
            powers = powers / norm
            self._k_factor = float(los_power / norm)
            self._los_power = float(los_power)
        rays = {k: np.deg2rad(centre[k][:, None] + spread[k] * _RAY_OFFSETS[None, :]) for k in centre}
        if self._direction == "uplink":                         # the tables' departure side is the receiver
            rays = {"aod": rays["aoa"], "aoa": rays["aod"], "zod": rays["zoa"], "zoa": rays["zod"]}
            los_ang = {"aod": los_ang["aoa"], "aoa": los_ang["aod"], "zod": los_ang["zoa"], "zoa": los_ang["zod"]}
        self._rays, self._los_ang = rays, los_ang
        self._powers, self._delays_norm = powers, delays
        self._num_clusters = len(powers)
        self._xpr = float(np.power(10.0, p["xpr"] / 10.0))
        self._order = np.argsort(delays, kind="stable")

    num_clusters = property(lambda self: self._num_clusters)
    los = property(lambda self: self._los)
    powers = property(lambda self: self._powers)

    @property
    def k_factor(self):
        assert self._los, "This property is only defined for LoS models"
        return self._k_factor / self._powers[0]                 # cdl.py:345-351

    @property
    def delays(self):
        return self._delays_norm * self._delay_spread

    @property
    def delay_spread(self):
        return self._delay_spread

    @delay_spread.setter
    def delay_spread(self, value):
        self._delay_spread = float(value)

    def _tables(self):
        if self._dev is None:
            lam = SPEED_OF_LIGHT / self._carrier_frequency
            f_rx, a_rx, r_rx = _side_tables(self._rx_array, self._rx_o, self._rays["zoa"], self._rays["aoa"], lam)
            f_tx, a_tx, _ = _side_tables(self._tx_array, self._tx_o, self._rays["zod"], self._rays["aod"], lam)
            amp = np.sqrt(self._powers / self.NUM_RAYS)
            los = None
            if self._los:
                one = lambda v: np.array([[v]])
                lf_rx, la_rx, lr_rx = _side_tables(self._rx_array, self._rx_o, one(self._los_ang["zoa"]), one(self._los_ang["aoa"]), lam)
                lf_tx, la_tx, _ = _side_tables(self._tx_array, self._tx_o, one(self._los_ang["zod"]), one(self._los_ang["aod"]), lam)
                kf = self._k_factor
                amp = amp * np.sqrt(1 / (kf + 1))
                c2f = lambda z: np.stack([z.real, z.imag], -1).reshape(-1)
                los = np.concatenate([lf_rx.reshape(-1), lf_tx.reshape(-1), c2f(la_rx.reshape(-1)), c2f(la_tx.reshape(-1)),
                                      lr_rx.reshape(-1), [np.sqrt(kf / (kf + 1))]])
            f32 = lambda x: _ffi.to_device(np.ascontiguousarray(x, self._np_rdtype), self.rdtype)
            c64 = lambda x: _ffi.to_device(np.ascontiguousarray(x, self._np_cdtype), self.cdtype)
            i32 = lambda x: _ffi.to_device(np.ascontiguousarray(x, np.int32), torch.int32)
            self._dev = (f32(f_rx), f32(f_tx), c64(a_rx), c64(a_tx), f32(r_rx), i32(self._rx_array.pol_index()),
                         i32(self._tx_array.pol_index()), i32(self._order), f32(amp), f32(los) if los is not None else None,
                         float(np.sqrt(1 / self._xpr)), float(2 * PI / lam))
        return self._dev

    def __call__(self, batch_size, num_time_steps, sampling_frequency):
        f_rx, f_tx, a_rx, a_tx, r_rx, pol_rx, pol_tx, order, amp, los, xpr_scale, k0 = self._tables()
        b, n, t = int(batch_size), self._num_clusters, int(num_time_steps)
        u, s = self._rx_array.num_ant, self._tx_array.num_ant
        rng = config.rng
        call = rng.next_call()
        for _ in range(7):                                      # the kernel consumes calls call .. call+7
            rng.next_call()
        a = torch.empty((b, 1, u, 1, s, n, t), dtype=self.cdtype, device=_ffi.device())
        ws, ws_bytes = self._ws.get(_ffi.lib().samd_cdl_workspace_bytes(b, n))
        _ffi.check((_ffi.lib().samd_cdl_cir_c128 if self.precision == "double" else _ffi.lib().samd_cdl_cir_c64)(rng.seed, call, b, n, u, s, t, float(sampling_frequency), _ffi.ptr(f_rx),
                                               _ffi.ptr(f_tx), _ffi.ptr(a_rx), _ffi.ptr(a_tx), _ffi.ptr(r_rx), _ffi.ptr(pol_rx),
                                               _ffi.ptr(pol_tx), _ffi.ptr(order), _ffi.ptr(amp), _ffi.ptr(los), xpr_scale, k0,
                                               self._min_speed, self._max_speed, _ffi.ptr(ws), ws_bytes, _ffi.ptr(a),
                                               _ffi.stream()), "CDL")
        tau = torch.from_numpy((self._delays_norm * self._delay_spread)[self._order].astype(self._np_rdtype)).to(a.device)
        tau = tau.reshape(1, 1, 1, n).expand(b, 1, 1, n).contiguous()
        return wrap(a), wrap(tau)
