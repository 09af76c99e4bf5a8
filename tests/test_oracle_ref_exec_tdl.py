"""Pin of the TDL channel (config C4's channel; VERDICT row a14) against the reference's OWN ``channel/tr38901/tdl.py``
executed under the NumPy stand-in for TensorFlow (tests/golden/tdl_ref_stats.npz, tools/gen_tdl_ref_stats.py):

  * deterministic: delays, mean powers, LoS flag, K factor, specular power and Doppler range of all eight models as the
    host class ``sionna_amd.phy.channel.tr38901.TDL`` holds them == the reference object's, to float32 rounding;
  * statistical (the reference's draws come from TensorFlow's generator): per-tap powers, temporal autocorrelation over
    16 samples, independence of the antenna pairs, and - with rx / tx correlation matrices - the 8 x 8 spatial covariance
    of ``oracle/ofdm.py:tdl_cir`` (+ the host class's correlation square root), whose realisations the HIP kernel
    reproduces bit for bit (tests/test_gpu_ofdm.py), within the Monte-Carlo error of 8000 reference realisations."""
import os

import numpy as np
import pytest

from oracle import ofdm as o

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tdl_ref_stats.npz"))
FC, DS, V_MIN, V_MAX, FS, T = (float(v) for v in G["setup"])
T = int(T)
MODELS = ("A", "B", "C", "D", "E", "A30", "B100", "C300")
NUM = 2000


def _tdl(model, **kw):
    from sionna_amd.phy.channel.tr38901 import TDL
    return TDL(model, DS, FC, min_speed=V_MIN, max_speed=V_MAX, num_rx_ant=4, num_tx_ant=2, **kw)


def _exp_corr(n, r):
    """Hermitian Toeplitz correlation matrix: R[i, j] = r^(j - i) above the diagonal, its conjugate below."""
    i = np.arange(n)
    d = i[None, :] - i[:, None]
    return np.where(d >= 0, r ** np.abs(d), np.conj(r) ** np.abs(d)).astype(np.complex64)


@pytest.mark.parametrize("model", MODELS)
def test_tdl_parameters_equal_the_reference_objects(model):
    t, k = _tdl(model), f"{model}_"
    assert np.allclose(np.asarray(t.delays, np.float64), G[k + "delays"], rtol=2e-7, atol=0)
    assert np.allclose(np.asarray(t.mean_powers, np.float64), G[k + "mean_powers"], rtol=1e-6, atol=1e-9)
    assert bool(t.los) == bool(G[k + "los"]) and t.num_clusters == len(G[k + "delays"])
    if t.los:
        assert np.isclose(t.k_factor, G[k + "k_factor"], rtol=1e-6) and np.isclose(t.mean_power_los, G[k + "mean_power_los"], rtol=1e-6)
    assert np.allclose([t._min_doppler, t._max_doppler], G[k + "doppler"], rtol=1e-6)


def _stats(a):
    h = a[:, 0, :, 0].astype(np.complex128)
    B, ra, ta, P, _ = h.shape
    power = np.mean(np.abs(h[..., 0]) ** 2, axis=(0, 1, 2))
    num = np.sum(h * np.conj(h[..., :1]), axis=(0, 1, 2))
    v = h[..., 0].reshape(B, ra * ta, P)
    return power, num, np.einsum("bip,bjp->ij", v, np.conj(v)) / B


@pytest.mark.parametrize("model,corr", [(m, False) for m in MODELS] + [("A", True)])
def test_tdl_statistics_match_the_reference_executed_generator(model, corr):
    kw = dict(rx_corr_mat=_exp_corr(4, 0.7 + 0.2j), tx_corr_mat=_exp_corr(2, 0.5)) if corr else {}
    t, k = _tdl(model, **kw), f"{model}{'_corr' if corr else ''}_"
    acc = None
    for i in range(NUM // 250):                                    # (chunks: the oracle's [B,ra,ta,P,T,N] temporaries stay small)
        a, tau = o.tdl_cir(777, 4 * i, 250, T, FS, t.delays, t._mean_powers, t._min_doppler, t._max_doppler, 4, 2, 20,
                           los_power=(t._los_power if t.los else None), los_aoa=t._los_angle_of_arrival)
        if corr:                                                   # tdl.py:474-492 through the host class's square root
            v = a[:, 0, :, 0].reshape(250, 8, -1)                   # rx-major antenna pairs
            a = np.einsum("ij,bjx->bix", t._corr_sqrt, v).reshape(a[:, 0, :, 0].shape)[:, None, :, None]
        st = _stats(a)
        acc = st if acc is None else tuple(x + y for x, y in zip(acc, st))
    power, rho_num, cov = (x / (NUM // 250) for x in acc)
    rho = rho_num / rho_num[:, :1]
    rp, rr, rc = G[k + "power"], G[k + "rho"], G[k + "cov"]
    # a tap's sample power over N x 8 antenna pairs: relative error ~ 1 / sqrt(8 N) for Rayleigh taps
    assert np.allclose(power, rp, rtol=0.04, atol=2e-4), np.max(np.abs(power - rp) / rp)
    assert abs(power.sum() / rp.sum() - 1) < 0.01
    # Jakes autocorrelation (the specular line on the first tap of the LoS models): every tap, 16 lags
    w = rp / rp.sum()
    assert np.max(np.abs(rho - rr) * np.sqrt(w)[:, None]) < 0.03, np.max(np.abs(rho - rr) * np.sqrt(w)[:, None])
    assert np.allclose(rho[np.argmax(rp)], rr[np.argmax(rp)], atol=0.04)
    # spatial covariance summed over taps: identity x total power without correlation matrices, R_rx (x) conj(R_tx) with
    rel = np.linalg.norm(cov - rc) / np.linalg.norm(rc)
    assert rel < 0.05, rel
    if corr:
        want = np.kron(_exp_corr(4, 0.7 + 0.2j), np.conj(_exp_corr(2, 0.5))) * rp.sum()
        assert np.linalg.norm(rc - want) / np.linalg.norm(want) < 0.05         # the reference itself realises R_rx (x) conj(R_tx)
