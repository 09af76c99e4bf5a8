"""Pins oracle/ofdm.py's waveform / channel-conversion functions to the reference's OWN code executed here
(tools/gen_ofdm_time_ref_golden.py -> tests/golden/ofdm_time_ref_golden.npz): OFDMModulator, OFDMDemodulator,
subcarrier_frequencies, time_lag_discrete_time_channel, cir_to_ofdm_channel, cir_to_time_channel, ApplyTimeChannel -
including the fft 72 / l_min -6 ... l_max 10 / cyclic prefix 2 configuration of the one notebook curve (ISI regime) the
MI355X path does not reproduce.  float32 through FFTs and sums: 1e-5 of the signal scale."""
import os

import numpy as np
import pytest

from oracle import ofdm as o

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ofdm_time_ref_golden.npz")
CASES = ("cp2", "cp20", "c4", "small")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def close(a, b, tol=1e-5):
    scale = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() <= tol * scale * 4


@pytest.mark.parametrize("tag", CASES)
def test_time_domain_chain_matches_reference_execution(g, tag):
    fft, nsym, cp, l_min, l_max, B, nrx, nra, ntx, nta, P = (int(v) for v in g[f"{tag}_meta"])
    scs = float(g[f"{tag}_scs"])
    bw = fft * scs
    assert o.time_lag_discrete_time_channel(bw) == (l_min, l_max)
    x, a, tau = g[f"{tag}_x"], g[f"{tag}_a"], g[f"{tag}_tau"]
    xt = o.ofdm_modulate(x, cp)
    assert close(xt, g[f"{tag}_x_time"])
    for norm in (True, False):
        h = o.cir_to_time_channel(bw, a, tau, l_min, l_max, normalize=norm)
        assert close(h, g[f"{tag}_h_time_n{int(norm)}"]), norm
    h = g[f"{tag}_h_time_n1"]
    y = o.apply_time_channel(g[f"{tag}_x_time"].reshape(B, ntx, nta, -1), h)
    assert close(y, g[f"{tag}_y_time"])
    yrg = o.ofdm_demodulate(g[f"{tag}_y_time"], fft, l_min, cp)
    assert close(yrg, g[f"{tag}_y_rg"])
    f = o.subcarrier_frequencies(fft, scs)
    assert np.array_equal(f.astype(np.float32), g[f"{tag}_freqs"])
    a_f = a[..., cp:-1:(fft + cp)][..., :nsym]
    for norm in (True, False):
        hf = o.cir_to_ofdm_channel(f, a_f, tau, normalize=norm)
        assert close(hf, g[f"{tag}_h_freq_n{int(norm)}"], 2e-5), norm
