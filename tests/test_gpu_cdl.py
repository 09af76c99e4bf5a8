"""GPU parity of the CDL channel model (samd_cdl_cir_c64 + host tables) against oracle/cdl.py on the
same Philox streams (complex64 vs float64 reference: 1e-3 relative to the tap scale), plus the use of
the model inside the OFDM channel blocks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cdl as oc

FC = 3.5e9


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


def _arrays(mod, kind):
    if kind == "siso":
        return mod.Antenna("single", "V", "omni", FC), mod.Antenna("single", "V", "omni", FC)
    if kind == "mimo":
        return mod.Antenna("dual", "cross", "omni", FC), mod.AntennaArray(2, 2, "dual", "cross", "38.901", FC)
    return mod.AntennaArray(1, 2, "single", "H", "38.901", FC), mod.PanelArray(1, 2, "dual", "VH", "38.901", FC, num_rows=1, num_cols=2)


@pytest.mark.parametrize("model,direction,kind,T,orient,speeds", [
    ("A", "downlink", "siso", 14, None, (0.0, None)),
    ("A", "uplink", "mimo", 14, None, (3.0, 30.0)),
    ("B", "downlink", "mimo", 40, ([0.4, 0.1, -0.2], [0.1, 0.2, 0.05]), (10.0, 10.0)),
    ("C", "uplink", "panel", 5, ([3.0, 0.0, 0.3], None), (0.0, 20.0)),
    ("D", "downlink", "mimo", 14, None, (3.0, 30.0)),
    ("E", "uplink", "panel", 33, (None, [0.2, 0.3, 0.0]), (5.0, None)),
])
def test_cdl_vs_oracle(phy, model, direction, kind, T, orient, speeds):
    t38 = phy.channel.tr38901
    ut, bs = _arrays(t38, kind)
    out, obs = _arrays(oc, kind)
    uo, bo = orient if orient is not None else (None, None)
    kw = dict(ut_orientation=uo, bs_orientation=bo, min_speed=speeds[0], max_speed=speeds[1])
    cdl = t38.CDL(model, 300e-9, FC, ut, bs, direction, **kw)
    ref = oc.CDL(model, 300e-9, FC, out, obs, direction, **kw)
    phy.config.seed = 42
    B, fs = 33, 15e3 * 14
    a, tau = cdl(B, T, fs)
    a_ref, tau_ref = ref(42, 0, B, T, fs)
    assert tuple(a.shape) == a_ref.shape and tuple(tau.shape) == tau_ref.shape
    assert np.allclose(_np(tau), tau_ref, rtol=1e-6, atol=0)
    scale = np.sqrt(np.mean(np.abs(a_ref) ** 2))
    assert np.allclose(_np(a), a_ref, rtol=1e-3, atol=1e-3 * scale), np.max(np.abs(_np(a) - a_ref)) / scale
    # the second call continues on the stream (call index 8)
    a2, _ = cdl(4, T, fs)
    assert np.allclose(_np(a2), ref(42, 8, 4, T, fs)[0], rtol=1e-3, atol=1e-3 * scale)
    assert cdl.num_clusters == ref.num_clusters and cdl.los == ref.los


def test_cdl_in_ofdm_channel(phy):
    t38 = phy.channel.tr38901
    ut, bs = t38.Antenna("single", "V", "omni", FC), t38.AntennaArray(1, 4, "dual", "cross", "38.901", FC)
    cdl = t38.CDL("B", 300e-9, FC, ut, bs, "uplink", min_speed=10.)
    rg = phy.ofdm.ResourceGrid(14, 76, 15e3, num_tx=1, num_streams_per_tx=1, cyclic_prefix_length=6,
                               num_guard_carriers=[5, 6], dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    ch = phy.channel.OFDMChannel(cdl, rg, normalize_channel=True, return_channel=True)
    phy.config.seed = 1
    x = _np(phy.ofdm.ResourceGridMapper(rg)(phy.mapping.QAMSource(2)([64, 1, 1, rg.num_data_symbols])))
    y, h = ch(x, 0.01)
    assert tuple(h.shape) == (64, 1, 8, 1, 1, 14, 76) and tuple(y.shape) == (64, 1, 8, 14, 76)
    e = np.mean(np.abs(_np(h)) ** 2, axis=(2, 4, 5, 6))
    assert np.allclose(e, 1.0, atol=1e-3)
    # frequency selectivity and time variation are present, and the two polarisations are weakly correlated
    hn = _np(h)[:, 0, :, 0, 0]
    assert np.std(np.abs(hn[:, 0, 0, :])) > 0.05 and np.mean(np.abs(hn[:, 0, 0, 0] - hn[:, 0, 13, 0])) > 1e-3
