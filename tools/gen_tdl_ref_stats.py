#!/usr/bin/env python3
"""Generates tests/golden/tdl_ref_stats.npz from the reference's OWN TDL code (``channel/tr38901/tdl.py`` :23-600, models/
TDL-*.json) executed from /root/reference/src under the NumPy stand-in for TensorFlow (tools/ref_exec):

  * the deterministic parameters of every model (A, B, C, D, E, A30, B100, C300): delays, mean powers (specular part
    included), LoS flag, K factor, Doppler range - compared EXACTLY with the product's / oracle's tables;
  * second-order statistics of NUM realisations (the random draws - Doppler, sinusoid angles and phases - come from a
    NumPy generator instead of TensorFlow's, so realisations cannot be compared): per-tap power, the temporal
    autocorrelation over 16 samples (Jakes: J0(2 pi f_D t); LoS taps carry the specular line), the cross-correlation
    between antenna pairs (zero without correlation matrices), and for one case with ``rx_corr_mat`` / ``tx_corr_mat``
    the 8 x 8 spatial covariance.
tests/test_oracle_ref_exec_tdl.py holds oracle/ofdm.py:tdl_cir (whose realisations the HIP kernel reproduces bit for bit,
tests/test_gpu_ofdm.py) to these numbers.  Run here (needs /root/reference); the fixture travels."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "tdl_ref_stats.npz")
NUM = 8000
MODELS = ("A", "B", "C", "D", "E", "A30", "B100", "C300")
FC, DS, V_MIN, V_MAX, FS, T = 2.6e9, 300e-9, 20.0, 30.0, 4000.0, 16


def load():
    from tools.ref_exec import tf_numpy
    from tools.ref_exec.loader import reference, _Block
    ref = reference()
    ref.load_utils()
    tf = ref.tf
    tf.linalg.matrix_transpose = lambda a, **k: tf_numpy._t(np.swapaxes(np.asarray(a), -1, -2))
    chan = sys.modules["sionna.phy.channel"]
    cu = ref.load("sionna.phy.channel.utils")
    for k, v in vars(cu).items():
        if not k.startswith("_"):
            setattr(chan, k, v)

    class ChannelModel(_Block):
        pass
    chan.ChannelModel = ChannelModel
    base = "/root/reference/src/sionna/phy/channel/tr38901"
    t38 = types.ModuleType("sionna.phy.channel.tr38901")
    t38.__path__, t38.__package__ = [base], "sionna.phy.channel.tr38901"
    sys.modules["sionna.phy.channel.tr38901"] = t38
    chan.tr38901 = t38
    mods = types.ModuleType("sionna.phy.channel.tr38901.models")
    mods.__path__ = [base + "/models"]
    sys.modules["sionna.phy.channel.tr38901.models"] = mods
    t38.models = mods
    return ref, ref.load("sionna.phy.channel.tr38901.tdl")


def stats(a):
    """a [B,1,ra,1,ta,P,T] -> per-tap power [P], autocorrelation [P,T] (normalised), mean |cross-correlation| between
    different antenna pairs at lag 0, [ra ta, ra ta] covariance summed over taps"""
    h = a[:, 0, :, 0].astype(np.complex128)                          # [B,ra,ta,P,T]
    B, ra, ta, P, _ = h.shape
    power = np.mean(np.abs(h[..., 0]) ** 2, axis=(0, 1, 2))
    num = np.sum(h * np.conj(h[..., :1]), axis=(0, 1, 2))            # [P,T]
    rho = num / num[:, :1]
    v = h[..., 0].reshape(B, ra * ta, P)
    cov = np.einsum("bip,bjp->ij", v, np.conj(v)) / B
    return power, rho, cov


def exp_corr(n, r):
    """Hermitian Toeplitz correlation matrix: R[i, j] = r^(j - i) above the diagonal, its conjugate below."""
    i = np.arange(n)
    d = i[None, :] - i[:, None]
    return np.where(d >= 0, r ** np.abs(d), np.conj(r) ** np.abs(d)).astype(np.complex64)


def main():
    ref, tdl = load()
    tf = ref.tf
    out = {"num": np.int64(NUM), "setup": np.array([FC, DS, V_MIN, V_MAX, FS, T])}
    cases = [(m, None) for m in MODELS] + [("A", "corr")]
    for model, corr in cases:
        sys.modules["sionna.phy"].config.tf_rng = tf.random.Generator(hash((model, corr)) % 100000)
        kw = dict(rx_corr_mat=exp_corr(4, 0.7 + 0.2j), tx_corr_mat=exp_corr(2, 0.5)) if corr else {}
        c = tdl.TDL(model, DS, FC, min_speed=V_MIN, max_speed=V_MAX, num_rx_ant=4, num_tx_ant=2, **kw)
        k = f"{model}{'_corr' if corr else ''}_"
        out[k + "delays"] = np.asarray(c.delays).astype(np.float64)
        out[k + "mean_powers"] = np.real(np.asarray(c.mean_powers)).astype(np.float64)
        out[k + "los"] = np.int64(bool(c.los))
        if c.los:
            out[k + "k_factor"] = np.float64(np.asarray(c.k_factor))
            out[k + "mean_power_los"] = np.float64(np.asarray(c.mean_power_los))
        out[k + "doppler"] = np.array([float(np.asarray(c._min_doppler)), float(np.asarray(c._max_doppler))])
        acc = None
        for _ in range(NUM // 1000):
            a, tau = c(1000, T, FS)
            s = stats(np.asarray(a))
            acc = s if acc is None else tuple(x + y for x, y in zip(acc, s))
        n_it = NUM // 1000
        out[k + "power"], out[k + "rho"], out[k + "cov"] = (x / n_it for x in acc)
        assert np.allclose(np.asarray(tau)[0, 0, 0], out[k + "delays"], rtol=1e-6)
        print(model, corr, "taps", len(out[k + "delays"]), "los", bool(c.los), "sum power", float(out[k + "power"].sum()),
              "|rho[tap 1]|", np.round(np.abs(out[k + "rho"][1, :5]), 3))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
