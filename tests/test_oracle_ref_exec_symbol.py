"""Pins the SYMBOL-domain rows (logits / indices of constellation points instead of bit LLRs) to the reference's OWN code
executed here: tests/golden/symbol_ref_golden.npz comes from tools/gen_symbol_ref_golden.py, which runs
LLRs2SymbolLogits, SymbolLogits2Moments, SymbolInds2Bits, QAM2PAM, PAM2QAM (mapping.py:969-1314), the mimo detectors with
``output="symbol"`` (EPDetector soft / hard, KBestDetector hard, MMSEPICDetector with logits as priors, LinearDetector;
mimo/detection.py) and their OFDM wrappers (ofdm/detection.py) from the reference's source files under the NumPy stand-in
for TensorFlow.  Index tables and decisions: exact; float pipelines: 1e-5 of scale (fixed-point detectors: bars below)."""
import os

import numpy as np
import pytest

from oracle import mapping as om, ofdm as o, mimo_f32 as mf

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "symbol_ref_golden.npz"))
MIMO = [tuple(int(v) for v in r) for r in GOLD["mimo_cases"]]


def close(a, b, tol=1e-5):
    return np.abs(np.asarray(a, np.float64) - b).max() <= 4 * tol * max(np.abs(b).max(), 1.0)


@pytest.mark.parametrize("m", [1, 2, 4, 6])
def test_llrs2symbol_logits_and_inds2bits(m):
    llrs = GOLD[f"l2s{m}_llrs"]
    assert close(om.llrs2symbol_logits(llrs, m), GOLD[f"l2s{m}_logits"])
    assert np.array_equal(om.llrs2symbol_logits(llrs, m, hard_out=True), GOLD[f"l2s{m}_hard"])
    assert np.array_equal(om.symbol_inds2bits(GOLD[f"i2b{m}_ind"], m), GOLD[f"i2b{m}_bits"])


@pytest.mark.parametrize("m", [2, 4, 6])
def test_moments_and_pam_qam_tables(m):
    mean, var = om.symbol_logits2moments(GOLD[f"mom{m}_logits"], om.qam(m))
    assert close(mean.real, GOLD[f"mom{m}_mean"].real) and close(mean.imag, GOLD[f"mom{m}_mean"].imag) and close(var, GOLD[f"mom{m}_var"])
    p1, p2 = om.qam2pam(GOLD[f"q2p{m}_q"], m)
    assert np.array_equal(p1, GOLD[f"q2p{m}_p1"]) and np.array_equal(p2, GOLD[f"q2p{m}_p2"])
    assert np.array_equal(om.pam2qam(p1, p2, m), GOLD[f"p2q{m}_q"]) and np.array_equal(GOLD[f"p2q{m}_q"], GOLD[f"q2p{m}_q"])
    # logits: float32 adds of the same pairs -> bit for bit
    assert np.array_equal(om.pam2qam(GOLD[f"p2q{m}_a"], GOLD[f"p2q{m}_b"], m, hard_in_out=False), GOLD[f"p2q{m}_logits"])


def _case(ci):
    return {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith(f"m{ci}/")}


@pytest.mark.parametrize("ci", range(len(MIMO)))
def test_mimo_detectors_symbol_output(ci):
    M, K, m = MIMO[ci]
    g = _case(ci)
    y, h, s = g["y"], g["h"], g["s"]
    pts = om.qam(m)
    # EP: l = 6 damped fixed-point iterations in float32 on the reference side, float64 here
    ep = o.ep_detector(y, h, s, m, l=6, output="symbol")
    assert ep.shape == g["ep_logits"].shape and close(ep, g["ep_logits"], 2e-4)
    eph = o.ep_detector(y, h, s, m, l=6, hard_out=True, output="symbol")
    assert np.mean(eph == g["ep_hard"]) >= 0.99
    kb = o.kbest_detector(y, h, s, pts, int(g["kbest_k"]), hard_out=True, output="symbol")
    assert np.array_equal(kb, g["kbest_hard"])
    for meth in ("app", "maxlog"):
        pic = o.mmse_pic(y, h, s, g["pic_prior"], pts, meth, 2, output="symbol")
        assert close(pic, g[f"pic_logits_{meth}"], 1e-4), meth
    pich = o.mmse_pic(y, h, s, g["pic_prior"], pts, "maxlog", 1, hard_out=True, output="symbol")
    assert np.mean(pich == g["pic_hard"]) >= 0.99
    # LinearDetector(output="symbol"): lmmse_equalizer + SymbolDemapper
    xh, ne = mf.lmmse_equalizer(y, h, s)
    lin = om.symbol_demapper(xh, ne, pts)
    assert close(lin, g["lin_logits"], 1e-4)
    assert np.mean(om.symbol_demapper(xh, ne, pts, hard_out=True) == g["lin_hard"]) >= 0.99


def test_ofdm_detectors_symbol_output():
    from tests.test_oracle_ref_exec_ofdm_rx import link
    rx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ofdm_rx_ref_golden.npz"))
    L, g, rg, sm = link(rx, "c4")
    c = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith("c4/")}
    y, no, hh, ev = g["y"], g["no"], g["h_hat_lin"], g["err_var_lin"]
    pts = om.qam(L["m"])
    ep = o.ofdm_ep_detector(rg, sm, y, hh, ev, no, L["m"], l=6, output="symbol")
    assert ep.shape == c["ep_logits"].shape and close(ep, c["ep_logits"], 2e-4)
    eph = o.ofdm_ep_detector(rg, sm, y, hh, ev, no, L["m"], l=6, hard_out=True, output="symbol")
    assert eph.shape == c["ep_hard"].shape and np.mean(eph == c["ep_hard"]) >= 0.995
    kb = o.ofdm_kbest_detector(rg, sm, y, hh, ev, no, pts, L["kbest"], hard_out=True, output="symbol")
    assert kb.shape == c["kbest_hard"].shape and np.mean(kb == c["kbest_hard"]) >= 0.995
    pic = o.ofdm_mmse_pic(rg, sm, y, hh, c["pic_prior"], ev, no, pts, "maxlog", 2, output="symbol")
    assert pic.shape == c["pic_logits"].shape and close(pic, c["pic_logits"], 1e-4)
    pich = o.ofdm_mmse_pic(rg, sm, y, hh, c["pic_prior"], ev, no, pts, "app", 1, hard_out=True, output="symbol")
    assert pich.shape == c["pic_hard"].shape and np.mean(pich == c["pic_hard"]) >= 0.995
