"""GPU parity of the SYMBOL-domain rows - ``LLRs2SymbolLogits``, ``SymbolLogits2Moments``, ``SymbolInds2Bits``, ``QAM2PAM``,
``PAM2QAM`` (mapping.py:969-1314) and ``output="symbol"`` of the EP / K-Best / MMSE-PIC / linear detectors (mimo and OFDM) -
against the reference's own code executed under the NumPy stand-in (tests/golden/symbol_ref_golden.npz, generator
tools/gen_symbol_ref_golden.py; the oracle is held to the same fixture in tests/test_oracle_ref_exec_symbol.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import mapping as om, ofdm as o

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "symbol_ref_golden.npz"))
MIMO = [tuple(int(v) for v in r) for r in GOLD["mimo_cases"]]


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


def close(a, b, tol=1e-5):
    a = np.asarray(a)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b).max()
    bar = 4 * tol * max(np.abs(b).max(), 1.0)
    print(f"max err {err:.3e} (bar {bar:.3e})")
    return err <= bar


def ep_close(a, b):
    """EP: six damped fixed-point iterations with a matrix inverse each, float32 on both sides (NumPy's inverse there, a
    Gauss-Jordan elimination in the kernel): the bar of the bit-output test against the same reference
    (tests/test_gpu_ofdm.py: element-wise relative error, floor 1)"""
    assert a.shape == b.shape, (a.shape, b.shape)
    r = np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1.0)
    print(f"EP logits: max rel {r.max():.3e}, median {np.quantile(r, 0.5):.3e}")
    return r.max() < 6e-2 and np.quantile(r, 0.5) < 2e-3


@pytest.mark.parametrize("m", [1, 2, 4, 6])
def test_llrs2symbol_logits_and_inds2bits(phy, m):
    mp = phy.mapping
    llrs = GOLD[f"l2s{m}_llrs"]
    assert close(_np(mp.LLRs2SymbolLogits(m)(llrs)), GOLD[f"l2s{m}_logits"])
    hard = mp.LLRs2SymbolLogits(m, hard_out=True)(llrs)
    assert hard.dtype == __import__("torch").int32 and np.array_equal(_np(hard), GOLD[f"l2s{m}_hard"])
    assert np.array_equal(_np(mp.SymbolInds2Bits(m)(GOLD[f"i2b{m}_ind"])), GOLD[f"i2b{m}_bits"])
    # a batch that spans several workgroup tiles, against the float64 oracle
    big = (np.random.default_rng(m).normal(size=(1000, 3, m)) * 5).astype(np.float32)
    assert close(_np(mp.LLRs2SymbolLogits(m)(big)), om.llrs2symbol_logits(big, m))
    assert np.mean(_np(mp.LLRs2SymbolLogits(m, hard_out=True)(big)) == om.llrs2symbol_logits(big, m, True)) > 0.999
    assert mp.LLRs2SymbolLogits(m)(np.zeros((0, m), np.float32)).shape == (0, 1 << m)


@pytest.mark.parametrize("m", [2, 4, 6])
def test_moments_and_pam_qam(phy, m):
    mp = phy.mapping
    mean, var = mp.SymbolLogits2Moments("qam", m)(GOLD[f"mom{m}_logits"])
    assert close(_np(mean).real, GOLD[f"mom{m}_mean"].real) and close(_np(mean).imag, GOLD[f"mom{m}_mean"].imag)
    assert close(_np(var), GOLD[f"mom{m}_var"])
    p1, p2 = mp.QAM2PAM(m)(GOLD[f"q2p{m}_q"])
    assert np.array_equal(_np(p1), GOLD[f"q2p{m}_p1"]) and np.array_equal(_np(p2), GOLD[f"q2p{m}_p2"])
    assert np.array_equal(_np(mp.PAM2QAM(m)(p1, p2)), GOLD[f"q2p{m}_q"])
    out = mp.PAM2QAM(m, hard_in_out=False)(GOLD[f"p2q{m}_a"], GOLD[f"p2q{m}_b"])
    assert np.array_equal(_np(out), GOLD[f"p2q{m}_logits"])          # one float32 add per entry: bit for bit


@pytest.mark.parametrize("ci", range(len(MIMO)))
def test_mimo_detectors_symbol_output(phy, ci):
    import torch
    M, K, m = MIMO[ci]
    g = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith(f"m{ci}/")}
    y, h, s = g["y"], g["h"], g["s"]
    mimo = phy.mimo
    ep = _np(mimo.EPDetector("symbol", m, l=6)(y, h, s))
    assert ep_close(ep, g["ep_logits"])
    eph = mimo.EPDetector("symbol", m, hard_out=True, l=6)(y, h, s)
    assert eph.dtype == torch.int32 and np.mean(_np(eph) == g["ep_hard"]) >= 0.95
    kb = mimo.KBestDetector("symbol", K, int(g["kbest_k"]), "qam", m, hard_out=True)(y, h, s)
    assert kb.dtype == torch.int32 and np.mean(_np(kb) == g["kbest_hard"]) >= 0.95
    with pytest.raises(AssertionError):
        mimo.KBestDetector("symbol", K, 4, "qam", m)
    for meth in ("app", "maxlog"):
        pic = _np(mimo.MMSEPICDetector("symbol", meth, 2, "qam", m)(y, h, s, g["pic_prior"]))
        assert close(pic, g[f"pic_logits_{meth}"], 2e-4), meth
    pich = mimo.MMSEPICDetector("symbol", "maxlog", 1, "qam", m, hard_out=True)(y, h, s, g["pic_prior"])
    assert pich.dtype == torch.int32 and np.mean(_np(pich) == g["pic_hard"]) >= 0.95
    lin = _np(mimo.LinearDetector("lmmse", "symbol", "app", "qam", m)(y, h, s))
    assert close(lin, g["lin_logits"], 2e-4)
    linh = _np(mimo.LinearDetector("lmmse", "symbol", "app", "qam", m, hard_out=True)(y, h, s))
    assert np.mean(linh == g["lin_hard"]) >= 0.95


def test_ofdm_detectors_symbol_output(phy):
    import torch
    rx = np.load(os.path.join(os.path.dirname(__file__), "golden", "ofdm_rx_ref_golden.npz"))
    g = {k.split("/", 1)[1]: rx[k] for k in rx.files if k.startswith("c4/")}
    c = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith("c4/")}
    m, T, S = 4, 2, 1
    rg = phy.ofdm.ResourceGrid(num_ofdm_symbols=14, fft_size=76, subcarrier_spacing=15e3, num_tx=T, num_streams_per_tx=S,
                               cyclic_prefix_length=6, num_guard_carriers=[3, 4], dc_null=True, pilot_pattern="kronecker",
                               pilot_ofdm_symbol_indices=[2, 11])
    sm = phy.mimo.StreamManagement(np.ones([1, T]), S)
    y, no, hh, ev = g["y"], g["no"], g["h_hat_lin"], g["err_var_lin"]
    od = phy.ofdm
    ep = _np(od.EPDetector("symbol", rg, sm, m, l=6)(y, hh, ev, no))
    assert ep_close(ep, c["ep_logits"])
    eph = od.EPDetector("symbol", rg, sm, m, l=6, hard_out=True)(y, hh, ev, no)
    assert eph.dtype == torch.int32 and eph.shape == c["ep_hard"].shape and np.mean(_np(eph) == c["ep_hard"]) >= 0.99
    kb = od.KBestDetector("symbol", T * S, 16, rg, sm, constellation_type="qam", num_bits_per_symbol=m, hard_out=True)(y, hh, ev, no)
    assert kb.dtype == torch.int32 and kb.shape == c["kbest_hard"].shape and np.mean(_np(kb) == c["kbest_hard"]) >= 0.99
    const = phy.mapping.Constellation("qam", m)
    pic = _np(od.MMSEPICDetector("symbol", "maxlog", rg, sm, num_iter=2, constellation=const)(y, hh, c["pic_prior"], ev, no))
    assert close(pic, c["pic_logits"], 2e-4)
    pich = od.MMSEPICDetector("symbol", "app", rg, sm, num_iter=1, constellation=const, hard_out=True)(y, hh, c["pic_prior"], ev, no)
    assert pich.dtype == torch.int32 and pich.shape == c["pic_hard"].shape and np.mean(_np(pich) == c["pic_hard"]) >= 0.99
