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


# ------------------------------------------------------------------ linear interpolation (LS, "lin")
def _ref_linear_int(h, i, j, time_avg=False):
    """Independent restatement of the reference test's NumPy model (test_ofdm_channel_estimation.py:17-88):
    per OFDM symbol piecewise-linear through its pilots (extrapolating with the outer segments), then the
    same across the pilot-carrying symbols."""
    T, F = h.shape
    hf = np.zeros_like(h)
    for t in range(T):
        cols = np.sort(j[i == t])
        if len(cols) == 1:
            hf[t] = h[t, cols[0]]
        elif len(cols) > 1:
            for f in range(F):
                k = np.searchsorted(cols, f, side="left")          # first pilot >= f
                k1 = min(max(k, 1), len(cols) - 1)
                a, b = cols[k1 - 1], cols[k1]
                hf[t, f] = (f - a) * (h[t, b] - h[t, a]) / (b - a) + h[t, a]
    syms = np.where(np.sum(np.abs(hf), axis=-1))[0]
    if time_avg:
        hf[syms] = np.sum(hf, axis=0) / len(syms)
    if len(syms) == 1:
        return np.repeat(hf[syms], T, axis=0)
    out = np.zeros_like(h)
    for t in range(T):
        k = np.searchsorted(syms, t, side="left")
        k1 = min(max(k, 1), len(syms) - 1)
        a, b = syms[k1 - 1], syms[k1]
        out[t] = (t - a) * (hf[b] - hf[a]) / (b - a) + hf[a]
    return out


def _sparse_pattern():
    mask = np.zeros([4, 1, 14, 64], bool)
    mask[..., [2, 3, 10, 11], :] = True
    pilots = np.zeros([4, 1, int(mask[0, 0].sum())], np.complex64)
    pilots[0, 0, 10] = 1; pilots[0, 0, 234] = 1; pilots[1, 0, 20] = 1; pilots[2, 0, 70] = 1; pilots[3, 0, 120] = 1
    return o.PilotPattern(mask, pilots)


def _kron(num_tx, ns, T, F, idx):
    return o.ResourceGrid(T, F, 30e3, num_tx=num_tx, num_streams_per_tx=ns, pilot_pattern="kronecker",
                          pilot_ofdm_symbol_indices=list(idx)).pilot_pattern


LIN_PATTERNS = {"sparse": _sparse_pattern, "k01": lambda: _kron(4, 1, 14, 64, [2, 11]), "k02": lambda: _kron(4, 1, 14, 64, [2]),
                "k03": lambda: _kron(16, 1, 14, 16, [2]), "k04": lambda: _kron(4, 2, 14, 64, [2, 5, 8]),
                "k05": lambda: _kron(1, 1, 5, 64, range(5)), "k06": lambda: _kron(4, 1, 14, 64, [2, 3, 8, 11])}


@pytest.mark.parametrize("name", sorted(LIN_PATTERNS))
@pytest.mark.parametrize("time_avg", [False, True])
def test_linear_interpolator_matches_reference_model(name, time_avg):
    """Reference test_ofdm_channel_estimation.py:91-330: noise-free LS estimates interpolated linearly equal
    the NumPy model applied to the true channel, for the reference's own pilot patterns."""
    pp = LIN_PATTERNS[name]()
    rng = np.random.default_rng(len(name))
    ntx, ns, T, F = pp.mask.shape
    h_true = (rng.normal(size=(2, ntx, ns, T, F)) + 1j * rng.normal(size=(2, ntx, ns, T, F))).astype(np.complex64)
    # perfect estimates at the pilot positions, zeros where the pilot is zero
    h_p = np.zeros((2, ntx, ns, pp.pilots.shape[-1]), np.complex64)
    for a in range(ntx):
        for b in range(ns):
            ii, jj = np.where(pp.mask[a, b])
            h_p[:, a, b] = h_true[:, a, b, ii, jj] * (np.abs(pp.pilots[a, b]) > 0)
    h_hat, ev = o.LinearInterpolator(pp, time_avg)(h_p, np.ones(h_p.shape, np.float32))
    for a in range(ntx):
        for b in range(ns):
            ii, jj = np.where(pp.mask[a, b])
            nz = np.abs(pp.pilots[a, b]) > 0
            for r in range(2):
                ref = _ref_linear_int(h_true[r, a, b].astype(np.complex128), ii[nz], jj[nz], time_avg)
                assert np.allclose(h_hat[r, a, b], ref, atol=1e-5)
    assert np.allclose(ev[0, 0, 0][np.abs(ev[0, 0, 0]) > 0], 1.0, atol=1e-5) or name == "sparse"
