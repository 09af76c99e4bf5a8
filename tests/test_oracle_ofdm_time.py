"""CPU tests pinning the time-domain part of oracle/ofdm.py with the properties the reference's
own unit tests assert (test/unit/ofdm/test_ofdm.py:14-125: cyclic prefix equals the symbol tail,
modulate -> demodulate is the identity, trailing samples are ignored; test/unit/channel/
test_apply_channel.py:11-59: ApplyTimeChannel equals the explicit loop; test/unit/channel/
test_channel_utils.py:83-135: the time-domain chain reproduces the channel frequency response)."""
import numpy as np
import pytest

from oracle import ofdm as o


def _qpsk(rng, shape):
    return ((rng.integers(0, 2, shape) * 2 - 1) + 1j * (rng.integers(0, 2, shape) * 2 - 1)).astype(np.complex64) / np.sqrt(2).astype(np.float32)


@pytest.mark.parametrize("cp", [0, 1, 12, 71, 72])
def test_cyclic_prefix_and_roundtrip(cp):
    rng = np.random.default_rng(cp)
    x = _qpsk(rng, (8, 14, 72))
    xt = o.ofdm_modulate(x, cp)
    assert xt.shape == (8, 14 * (72 + cp))
    sym = xt.reshape(8, 14, -1)
    if cp:
        assert np.array_equal(sym[..., :cp], sym[..., -cp:])
    assert np.max(np.abs(o.ofdm_demodulate(xt, 72, 0, cp) - x)) < 1e-5
    # trailing samples that do not fill a symbol are dropped (test_overlapping_input)
    xt2 = np.concatenate([xt, xt[..., :10]], axis=-1)
    assert np.max(np.abs(o.ofdm_demodulate(xt2, 72, 0, cp) - x)) < 1e-5


def test_variable_cyclic_prefix():
    rng = np.random.default_rng(1)
    cps = np.arange(72)
    x = _qpsk(rng, (4, 3, 72, 72))
    xt = o.ofdm_modulate(x, cps)
    start = 0
    for i in range(72):
        end = start + cps[i] + 72
        s = xt[..., start:end]
        assert np.array_equal(s[..., :cps[i]], s[..., s.shape[-1] - cps[i]:])
        start = end
    assert np.max(np.abs(o.ofdm_demodulate(xt, 72, 0, cps, 72) - x)) < 1e-5


@pytest.mark.parametrize("tn,l_tot", [(1, 1), (5, 3), (32, 8), (40, 16)])
def test_apply_time_channel_matches_loop(tn, l_tot):
    rng = np.random.default_rng(tn)
    B, rx, ra, tx, ta = 3, 2, 2, 2, 2
    x = (rng.normal(size=(B, tx, ta, tn)) + 1j * rng.normal(size=(B, tx, ta, tn))).astype(np.complex64)
    h = (rng.normal(size=(B, rx, ra, tx, ta, tn + l_tot - 1, l_tot)) + 1j * rng.normal(size=(B, rx, ra, tx, ta, tn + l_tot - 1, l_tot))).astype(np.complex64)
    y = o.apply_time_channel(x, h)
    ref = np.zeros((B, rx, ra, tn + l_tot - 1), np.complex128)
    for t in range(tn + l_tot - 1):
        for l in range(l_tot):
            if t - l < 0:
                break
            if t - l > tn - 1:
                continue
            ref[..., t] += np.sum(x[:, None, None, :, :, t - l] * h[:, :, :, :, :, t, l], axis=(3, 4))
    assert np.allclose(ref, y, atol=1e-5)


@pytest.mark.parametrize("l_min,l_max", [(0, 0), (-3, 7), (-10, 0), (-6, 4)])
def test_time_domain_chain_reproduces_frequency_response(l_min, l_max):
    """Static TDL-A channel, |l_min| + l_max <= cyclic prefix: demod(h_time * mod(x)) =
    H[k] x[k] with H the DFT of the taps (channel/utils.py time_to_ofdm_channel :352-420)."""
    cp, n, nsym, B = 10, 128, 5, 4
    bw = n * 15e3
    L = l_max - l_min + 1
    tn = nsym * (n + cp)
    import json, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = json.load(open(os.path.join(here, "sionna_amd/phy/channel/tr38901/tdl_models.json")))["A"]
    delays = np.asarray(m["delays"], np.float64) * 100e-9
    powers = 10 ** (np.asarray(m["powers"], np.float64) / 10)
    powers /= powers.sum()
    a, tau = o.tdl_cir(3, 0, B, tn + L - 1, bw, delays, powers, 0., 0.)
    h_time = o.cir_to_time_channel(bw, a, tau, l_min, l_max, normalize=True)
    e = np.mean(np.sum(np.abs(h_time) ** 2, axis=6), axis=(2, 4, 5))
    assert np.allclose(e, 1.0, atol=1e-5)
    rng = np.random.default_rng(0)
    x = _qpsk(rng, (B, 1, 1, nsym, n))
    y = o.apply_time_channel(o.ofdm_modulate(x, cp).reshape(B, 1, 1, tn), h_time)
    yf = o.ofdm_demodulate(y, n, l_min, cp)                      # [B,1,1,nsym,n]
    taps = h_time[:, 0, 0, 0, 0, 0, :].astype(np.complex128)     # static channel
    k = np.arange(n) - n // 2
    H = (taps[:, None, :] * np.exp(-2j * np.pi * k[None, :, None] * np.arange(l_min, l_max + 1)[None, None, :] / n)).sum(-1)
    assert np.allclose(yf[:, 0, 0], H[:, None, :] * x[:, 0, 0], atol=2e-5)
