"""Pins the OFDM receiver front end with ESTIMATED channel state to the reference's OWN code executed here:
tests/golden/ofdm_rx_ref_golden.npz comes from tools/gen_ofdm_rx_ref_golden.py, which runs ResourceGrid /
ResourceGridMapper / RemoveNulledSubcarriers, LSChannelEstimator ("nn", "lin", "lin_time_avg"), LMMSE / ZF / MF
OFDM equalizers and the LinearDetector / KBestDetector / EPDetector / MMSEPICDetector from the reference's source files
under the NumPy stand-in for TensorFlow - guard carriers, DC null, error variance > 0, per-example noise variance, one
transmitter with four streams and two transmitters with one.  The oracle on the same received grid must agree to 1e-5 of
each quantity's scale (list / fixed-point detectors: see the bars below)."""
import os

import numpy as np
import pytest

from oracle import ofdm as o, mapping as om

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ofdm_rx_ref_golden.npz")
# "c4b" (tools/gen_ofdm_rx_ref_golden.py --baseline, fixture ofdm_rx_ref_golden_c4.npz): BASELINE config C4's grid ITSELF -
# one transmitter with two streams, guards [5, 6], DC null, pilots at symbols 2 and 11, four receive antennas, QPSK
GOLD_C4 = os.path.join(os.path.dirname(__file__), "golden", "ofdm_rx_ref_golden_c4.npz")
LINKS = {"c4": dict(fft=76, guards=(3, 4), num_tx=2, spt=1, m=4, kbest=16), "cdl": dict(fft=72, guards=(5, 6), num_tx=1, spt=4, m=2, kbest=32),
         "c4b": dict(fft=76, guards=(5, 6), num_tx=1, spt=2, m=2, kbest=16)}


@pytest.fixture(scope="module")
def gold():
    merged = dict(np.load(GOLD))
    merged.update(np.load(GOLD_C4))
    return merged


def link(gold, name):
    L = LINKS[name]
    g = {k.split("/", 1)[1]: gold[k] for k in gold if k.startswith(name + "/")}
    rg = o.ResourceGrid(14, L["fft"], 15e3, num_tx=L["num_tx"], num_streams_per_tx=L["spt"], cyclic_prefix_length=6,
                        num_guard_carriers=L["guards"], dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    assert np.array_equal(rg.pilot_pattern.mask, g["mask"].astype(bool))
    rg.pilot_pattern._pilots = g["pilots"]                      # the generator's pilot symbols (before normalisation)
    return L, g, rg, o.StreamManagement(np.ones([1, L["num_tx"]]), L["spt"])


def close(a, b, tol=1e-5):
    a = np.asarray(a)
    a, b = np.broadcast_arrays(a, b) if a.ndim == b.ndim else (a.reshape(b.shape), b)   # (error variances: broadcastable)
    return np.abs(a - b).max() <= 4 * tol * np.abs(b).max()


@pytest.mark.parametrize("name", list(LINKS))
def test_grid_mapping_and_ls_estimators(gold, name):
    L, g, rg, sm = link(gold, name)
    pts = om.qam(L["m"])
    x = om.mapper(g["b"].astype(np.float32), pts)
    assert np.array_equal(o.rg_map(rg, x), g["x_rg"])
    assert np.array_equal(o.remove_nulled(rg, g["y"]), g["removed"])
    h, ev = o.ls_estimate(rg, g["y"], g["no"])
    assert np.array_equal(h, g["h_hat_nn"]) and close(ev, g["err_var_nn"], 1e-6)
    for key, avg in (("lin", False), ("lin_time_avg", True)):
        h, ev = o.ls_estimate_lin(rg, g["y"], g["no"], time_avg=avg)
        assert close(h, g[f"h_hat_{key}"]) and close(ev, g[f"err_var_{key}"]), key


@pytest.mark.parametrize("name", list(LINKS))
def test_equalizers_and_linear_detectors(gold, name):
    L, g, rg, sm = link(gold, name)
    pts = om.qam(L["m"])
    y, no, hh, ev = g["y"], g["no"], g["h_hat_lin"], g["err_var_lin"]
    xo, neo = o.ofdm_lmmse_equalize(rg, sm, y, hh, ev, no)
    assert close(xo, g["x_hat_lmmse"]) and close(neo, g["no_eff_lmmse"])
    if name == "c4b":                                            # the bench's chain: LS (nearest neighbour) -> LMMSE -> app demapper
        xn, nn = o.ofdm_lmmse_equalize(rg, sm, y, g["h_hat_nn"], g["err_var_nn"], no)
        assert close(xn, g["x_hat_lmmse_nn"]) and close(nn, g["no_eff_lmmse_nn"])
        assert close(om.demapper(xn.astype(np.complex64), nn.astype(np.float32), pts, "app"), g["llr_lmmse_nn_app"], 2e-5)
    for meth in ("app", "maxlog"):
        assert close(om.demapper(xo.astype(np.complex64), neo.astype(np.float32), pts, meth), g[f"llr_lmmse_{meth}"], 2e-5)
    x, ne = o.ofdm_linear_equalize(rg, sm, y, hh, ev, no, "mf")
    assert close(x, g["x_hat_mf"]) and close(ne, g["no_eff_mf"])
    # ZF inverts the Gram matrix of the ESTIMATE in complex64 on the reference side (complex128 here): the square 4 x 4 link
    # has resource elements with a noise enhancement of 4.7e3, so the bar is per element and follows the conditioning -
    # every element within 2e-3 of its own magnitude, 99 % within 1e-4
    x, ne = o.ofdm_linear_equalize(rg, sm, y, hh, ev, no, "zf")
    for a, b in ((x, g["x_hat_zf"]), (ne, g["no_eff_zf"])):
        rel = np.abs(a.reshape(b.shape) - b) / np.maximum(np.abs(b), 1e-3 * np.abs(b).max())
        assert rel.max() < 2e-3 and np.quantile(rel, 0.99) < 1e-4, (rel.max(), np.quantile(rel, 0.99))
    llr, ref = om.demapper(x.astype(np.complex64), ne.astype(np.float32), pts, "maxlog"), g["llr_zf_maxlog"]
    rel = np.abs(llr.reshape(ref.shape) - ref) / np.maximum(np.abs(ref), 1e-2 * np.abs(ref).max())
    assert rel.max() < 5e-3 and np.quantile(rel, 0.99) < 2e-4, (rel.max(), np.quantile(rel, 0.99))


@pytest.mark.parametrize("name", list(LINKS))
def test_nonlinear_detectors(gold, name):
    L, g, rg, sm = link(gold, name)
    pts = om.qam(L["m"])
    y, no, hh, ev = g["y"], g["no"], g["h_hat_lin"], g["err_var_lin"]
    for meth in ("app", "maxlog"):
        assert close(o.ofdm_mmse_pic(rg, sm, y, hh, g["prior"], ev, no, pts, meth, 2), g[f"llr_pic_{meth}"], 1e-4), meth
    kb = o.ofdm_kbest_detector(rg, sm, y, hh, ev, no, pts, L["kbest"])
    ref = g["llr_kbest"]
    # list detector: a near-tie among the survivors flips a counter-hypothesis in or out (tests/test_oracle_ref_exec_idd.py)
    assert np.mean(np.isclose(kb.reshape(ref.shape), ref, rtol=1e-4, atol=1e-3)) > 0.99
    # EP: six fixed-point iterations with a matrix inverse each, float32 on the reference side; LLRs up to +-200 here
    ep = o.ofdm_ep_detector(rg, sm, y, hh, ev, no, L["m"], l=6)
    ref = g["llr_ep"]
    rel = np.abs(ep.reshape(ref.shape) - ref) / np.maximum(np.abs(ref), 1.0)
    assert rel.max() < 1e-2 and np.quantile(rel, 0.5) < 1e-3, (rel.max(), np.quantile(rel, 0.5))
