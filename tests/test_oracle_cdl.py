"""CPU tests pinning the deterministic part of oracle/cdl.py by physics (TR 38.901 Sec. 7.1, 7.3, 7.5
step 11, 7.7.1): cluster tables, antenna pattern, array response, Doppler, polarisation, rotation
invariance, LoS / K-factor combination, and the statistics of the random part (power per cluster)."""
import numpy as np
import pytest

from oracle import cdl as oc

FC = 3.5e9
LAM = oc.SPEED_OF_LIGHT / FC


def _ant(pol="single", ptype="V", pattern="omni"):
    return oc.Antenna(pol, ptype, pattern, FC)


@pytest.mark.parametrize("model,n,los", [("A", 23, False), ("B", 23, False), ("C", 24, False), ("D", 13, True), ("E", 14, True)])
def test_tables(model, n, los):
    c = oc.CDL(model, 300e-9, FC, _ant(), _ant(), "downlink")
    assert c.num_clusters == n and c.los == los
    assert np.isclose(c.powers.sum(), 1.0)
    for k in ("aoa", "aod", "zoa", "zod"):
        assert c.rays[k].shape == (n, 20)
    if los:
        import json
        tab = json.load(open(oc._MODELS))[model]
        pw = 10 ** (np.array(tab["powers"]) / 10)
        assert np.isclose(c.k_factor, pw[0] / pw[1:].sum())            # LoS power over the total NLoS power (cdl.py:429-437)
        assert np.isclose(10 * np.log10(pw[0] / pw[1]), {"D": 13.3, "E": 22.03}[model], atol=0.05)   # K of TR 38.901 Tab. 7.7.1-4/5
    up = oc.CDL(model, 300e-9, FC, _ant(), _ant(), "uplink")
    assert np.array_equal(up.rays["aoa"], c.rays["aod"]) and np.array_equal(up.rays["zod"], c.rays["zoa"])
    assert up.moving_end == "tx" and c.moving_end == "rx"


def test_pattern_38901():
    assert np.isclose(10 * np.log10(oc.radiation_pattern("38.901", np.array(np.pi / 2), np.array(0.0))), 8.0)
    assert np.isclose(10 * np.log10(oc.radiation_pattern("38.901", np.array(np.pi / 2), np.array(np.pi))), -22.0)
    # 3 dB beamwidth of 65 degrees in azimuth
    assert np.isclose(10 * np.log10(oc.radiation_pattern("38.901", np.array(np.pi / 2), np.array(np.deg2rad(32.5)))), 5.0)
    ft, fp = oc.element_field("omni", -np.pi / 4, np.array(1.0), np.array(0.3))
    assert np.isclose(ft, np.cos(np.pi / 4)) and np.isclose(fp, -np.sin(np.pi / 4))


def test_panel_geometry():
    arr = oc.AntennaArray(2, 4, "dual", "cross", "38.901", FC)
    assert arr.num_ant == 16 and arr.ant_pos.shape == (16, 3)
    assert np.allclose(arr.ant_pos[:8], arr.ant_pos[8:]) and np.array_equal(arr.pol, [0] * 8 + [1] * 8)
    assert np.allclose(arr.ant_pos.mean(0), 0) and np.allclose(arr.ant_pos[:, 0], 0)
    # column-major element order, half-wavelength spacing, first element top-left
    assert np.allclose(arr.ant_pos[1] - arr.ant_pos[0], [0, 0, -0.5 * LAM]) and np.allclose(arr.ant_pos[2] - arr.ant_pos[0], [0, 0.5 * LAM, 0])
    pa = oc.PanelArray(1, 2, "dual", "VH", "omni", FC, num_rows=2, num_cols=2)
    assert pa.num_ant == 16 and np.array_equal(pa.pol[:4], [0, 0, 1, 1])


def test_rotation_and_lcs():
    o = (0.3, -0.2, 0.5)
    r = oc.rotation_matrix(o)
    assert np.allclose(r @ r.T, np.eye(3)) and np.isclose(np.linalg.det(r), 1)
    # a ray along the rotated local x axis has theta' = 90 deg, phi' = 0 in the LCS
    x_g = r @ np.array([1., 0., 0.])
    th, ph = np.arccos(x_g[2]), np.arctan2(x_g[1], x_g[0])
    tp, pp = oc.gcs_to_lcs(o, np.array(th), np.array(ph))
    assert np.isclose(tp, np.pi / 2) and np.isclose(pp, 0, atol=1e-12)
    # pure bearing rotation: phi' = phi - alpha, psi = 0, field unchanged
    tp, pp = oc.gcs_to_lcs((0.4, 0, 0), np.array(1.1), np.array(0.9))
    assert np.isclose(tp, 1.1) and np.isclose(pp, 0.5) and np.isclose(oc.psi_angle((0.4, 0, 0), np.array(1.1), np.array(0.9)), 0)
    # field magnitude is invariant under any rotation of the array (the polarisation basis rotates by psi)
    arr = _ant("dual", "cross", "omni")
    f = oc.field_gcs(arr, o, np.array(1.2), np.array(-0.4))
    assert np.allclose(np.sum(f ** 2, axis=-1), 1.0)


def _one_ray(c, aoa, aod, zoa, zod, pm, vel, t):
    f = lambda v: np.array([[v]])
    return c._link(f(aoa), f(aod), f(zoa), f(zod), np.asarray(pm, complex)[None, None], np.asarray(vel, float)[None], np.asarray(t, float))[0]


def test_array_response_and_doppler():
    rx = oc.AntennaArray(1, 2, "single", "V", "omni", FC)                 # two elements along y, half a wavelength apart
    c = oc.CDL("A", 100e-9, FC, rx, _ant(), "downlink", ut_orientation=[0., 0., 0.])
    pm = np.eye(2)
    h = _one_ray(c, np.deg2rad(30), 0.2, np.pi / 2, np.pi / 2, pm, [0, 0, 0], [0.0])
    # elements at y = -lambda/4 (index 0) and +lambda/4: phase(u1) - phase(u0) = 2 pi (d / lambda) sin(aoa) = pi / 2
    assert np.isclose(np.angle(h[1, 0, 0] / h[0, 0, 0]), np.pi * 0.5)
    assert np.allclose(np.abs(h), 1.0)
    # receiver moving along +x at 30 m/s, ray arriving from phi = 0: Doppler shift v / lambda
    t = np.arange(5) * 1e-4
    h = _one_ray(c, 0.0, 0.2, np.pi / 2, np.pi / 2, pm, [30., 0, 0], t)
    assert np.allclose(h[0, 0] / h[0, 0, 0], np.exp(2j * np.pi * 30 / LAM * t))
    # moving orthogonally to the ray: no Doppler
    h = _one_ray(c, 0.0, 0.2, np.pi / 2, np.pi / 2, pm, [0, 30., 0], t)
    assert np.allclose(h[0, 0], h[0, 0, 0])


def test_polarisation():
    t, z = [0.0], np.zeros(3)
    pm = np.array([[np.exp(0.3j), 0.1 * np.exp(1j)], [0.2 * np.exp(2j), np.exp(-0.7j)]])
    mk = lambda rxp, txp: oc.CDL("A", 1e-7, FC, _ant("single", rxp), _ant("single", txp), "downlink", ut_orientation=[0., 0., 0.])
    args = (0.3, -0.4, np.pi / 2, np.pi / 2, pm, z, t)                    # horizontal rays: psi = 0
    assert np.isclose(_one_ray(mk("V", "V"), *args)[0, 0, 0], pm[0, 0])
    assert np.isclose(_one_ray(mk("H", "H"), *args)[0, 0, 0], pm[1, 1])
    assert np.isclose(_one_ray(mk("H", "V"), *args)[0, 0, 0], pm[1, 0])   # F_rx^T PM F_tx
    assert np.isclose(_one_ray(mk("V", "H"), *args)[0, 0, 0], pm[0, 1])
    dual = oc.CDL("A", 1e-7, FC, _ant("dual", "VH"), _ant("dual", "VH"), "downlink", ut_orientation=[0., 0., 0.])
    assert np.allclose(_one_ray(dual, *args)[:, :, 0], pm)


@pytest.mark.parametrize("model", ["A", "D"])
def test_cluster_powers_and_los(model):
    c = oc.CDL(model, 300e-9, FC, _ant(), _ant(), "downlink", min_speed=3.0)
    a, tau = c(7, 0, 3000, 2, 1e3)
    assert a.shape == (3000, 1, 1, 1, 1, c.num_clusters, 2) and tau.shape == (3000, 1, 1, c.num_clusters)
    assert np.all(np.diff(tau[0, 0, 0]) >= 0) and np.isclose(tau[0, 0, 0, -1], np.max(c.delays) * 300e-9)
    p = np.mean(np.abs(a[:, 0, 0, 0, 0, :, 0]) ** 2, axis=0)
    ref = c.powers[c.order].copy()
    if c.los:
        ref = ref / (c.k_factor + 1)
        ref[0] += c.k_factor / (c.k_factor + 1)
    assert np.allclose(p, ref, rtol=0.15, atol=2e-4)
    assert np.isclose(p.sum(), 1.0, rtol=0.05)
    if c.los:                                   # the LoS tap is (almost) deterministic in magnitude
        assert np.std(np.abs(a[:, 0, 0, 0, 0, 0, 0])) < 0.2
    # reproducible, and different calls differ
    a2, _ = c(7, 0, 8, 2, 1e3)
    assert np.array_equal(a2, a[:8]) and not np.array_equal(c(7, 8, 8, 2, 1e3)[0], a2)


def test_random_coupling_is_a_permutation():
    c = oc.CDL("B", 1e-7, FC, _ant(), _ant(), "uplink")
    vel, perm, phi = c.draw(3, 0, 50)
    for k in perm:
        assert np.array_equal(np.sort(perm[k], axis=-1), np.broadcast_to(np.arange(20), perm[k].shape))
    assert not np.array_equal(perm["aoa"], perm["aod"]) and np.all(np.abs(phi) <= np.pi)
    assert np.allclose(np.linalg.norm(vel, axis=-1), 0.0)
