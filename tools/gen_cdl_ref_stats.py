#!/usr/bin/env python3
"""Generates tests/golden/cdl_ref_stats.npz: second-order statistics of the channel coefficients that the reference's OWN
CDL code produces - ``channel/tr38901/cdl.py`` (CDL :23-560), ``channel_coefficients.py`` (ChannelCoefficientsGenerator,
TR 38.901 section 7.5 steps 10-11), ``rays.py``, ``antenna.py`` (AntennaArray / PanelArray / AntennaElement patterns) -
executed from /root/reference/src under the NumPy stand-in for TensorFlow (tools/ref_exec).  The random draws (ray coupling,
initial phases, velocity direction) come from NumPy generators instead of TensorFlow's, so realisations cannot be compared;
their STATISTICS can: per model (A ... E, uplink 4 -> 8 dual-polarised antennas as in the MIMO-OFDM notebook, and one
downlink), from NUM realisations:
  power[n]        mean total power of cluster n over the 32 antenna pairs
  cov[32, 32]     spatial covariance of vec(a) summed over the clusters (rx-major), i.e. E[ vec(H) vec(H)^H ]
  pair_power[8,4] mean power per (rx antenna, tx antenna) pair summed over clusters
  rho[T]          normalised temporal autocorrelation of the coefficients at 30 m/s over T samples of 0.5 ms
  tau[n]          path delays (deterministic)
tests/test_oracle_ref_exec_cdl.py holds oracle/cdl.py (whose realisations the HIP kernel reproduces) to these numbers
within the Monte-Carlo error.  Run here (needs /root/reference); the fixture travels."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "cdl_ref_stats.npz")
NUM = 6000


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

    class Topology:                                                  # system_level_scenario.Topology: a record of tensors
        def __init__(self, velocities, moving_end, los_aoa=None, los_aod=None, los_zoa=None, los_zod=None, los=None,
                     distance_3d=None, tx_orientations=None, rx_orientations=None):
            self.__dict__.update(locals())
    t38.Topology = Topology
    ant = ref.load("sionna.phy.channel.tr38901.antenna")
    t38.Rays = ref.load("sionna.phy.channel.tr38901.rays").Rays
    t38.ChannelCoefficientsGenerator = ref.load("sionna.phy.channel.tr38901.channel_coefficients").ChannelCoefficientsGenerator
    mods = types.ModuleType("sionna.phy.channel.tr38901.models")
    mods.__path__ = [base + "/models"]
    sys.modules["sionna.phy.channel.tr38901.models"] = mods
    t38.models = mods
    cdl = ref.load("sionna.phy.channel.tr38901.cdl")
    return ref, ant, cdl


def stats(a):
    """a [B,1,U,1,S,N,T] complex -> dict of the statistics above (time index 0 for the spatial ones)."""
    a0 = a[:, 0, :, 0, :, :, 0]                                     # [B,U,S,N]
    B, U, S, N = a0.shape
    power = np.mean(np.sum(np.abs(a0) ** 2, axis=(1, 2)), axis=0)   # [N]
    v = a0.reshape(B, U * S, N)
    cov = np.einsum("bin,bjn->ij", v, np.conj(v)) / B
    pair = np.mean(np.sum(np.abs(a0) ** 2, axis=3), axis=0)         # [U,S]
    at = a[:, 0, :, 0, :, :, :]                                     # [B,U,S,N,T]
    num = np.sum(at * np.conj(at[..., :1]), axis=(0, 1, 2, 3))
    rho = num / num[0]
    return power, cov, pair, rho


def main():
    ref, ant, cdl = load()
    tf = ref.tf
    fc = 2.6e9
    ut = ant.AntennaArray(num_rows=1, num_cols=2, polarization="dual", polarization_type="cross", antenna_pattern="38.901", carrier_frequency=fc)
    bs = ant.AntennaArray(num_rows=1, num_cols=4, polarization="dual", polarization_type="cross", antenna_pattern="38.901", carrier_frequency=fc)
    out = {"num": np.int64(NUM)}
    for model, direction in [(m, "uplink") for m in "ABCDE"] + [("B", "downlink")]:
        sys.modules["sionna.phy"].config.tf_rng = tf.random.Generator(hash((model, direction)) % 100000)
        c = cdl.CDL(model, 100e-9, fc, ut, bs, direction, min_speed=30.0)
        acc = None
        for _ in range(NUM // 500):
            a, tau = c(500, 8, 2000.0)
            s = stats(np.asarray(a).astype(np.complex128))
            acc = s if acc is None else tuple(x + y for x, y in zip(acc, s))
        k = f"{model}_{direction}_"
        n_it = NUM // 500
        out[k + "power"], out[k + "cov"], out[k + "pair_power"], out[k + "rho"] = (x / n_it for x in acc)
        out[k + "tau"] = np.asarray(tau)[0, 0, 0]
        print(model, direction, "total power", float(np.sum(out[k + "power"])), "clusters", len(out[k + "power"]),
              "|rho|", np.round(np.abs(out[k + "rho"][:4]), 3))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
