#!/usr/bin/env python3
"""Generates tests/golden/ofdm_time_ref_golden.npz by EXECUTING the reference's own OFDM waveform / channel-conversion
code under the NumPy stand-in for TensorFlow (tools/ref_exec): ``ofdm/modulator.py`` (OFDMModulator :13-124),
``ofdm/demodulator.py`` (OFDMDemodulator :15-203), ``signal/utils.py`` (fft / ifft :150-262), ``channel/utils.py``
(subcarrier_frequencies :17-60, time_lag_discrete_time_channel :121-178, cir_to_ofdm_channel :180-254,
cir_to_time_channel :256-349) and ``channel/apply_time_channel.py`` (ApplyTimeChannel :14-175, without its AWGN).

The cases include the configuration of the one published BER curve the MI355X path does not reproduce (CDL uplink, fft 72,
cyclic prefix 2, l_min -6 ... l_max 10: the ISI regime) and the same with cyclic prefix 20.  All of it is float32
arithmetic through FFTs and sums: compared at 1e-5 of the signal scale.  Run here (needs /root/reference)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "ofdm_time_ref_golden.npz")


def cn(rng, shape, var=1.0):
    return ((rng.normal(size=shape) + 1j * rng.normal(size=shape)) * np.sqrt(var / 2)).astype(np.complex64)


def main():
    from tools.ref_exec.loader import reference
    ref = reference()
    ref.load_utils()
    ref.load_signal()
    mod = ref.load("sionna.phy.ofdm.modulator")
    dem = ref.load("sionna.phy.ofdm.demodulator")
    awgn = types.ModuleType("sionna.phy.channel.awgn")
    awgn.AWGN = lambda **k: None                                  # the fixture runs the channel without noise
    sys.modules["sionna.phy.channel.awgn"] = awgn
    cu = ref.load("sionna.phy.channel.utils")
    atc = ref.load("sionna.phy.channel.apply_time_channel")
    tf = ref.tf
    rng = np.random.default_rng(76)
    out = {}
    # (fft, symbols, cp, spacing, batch, (rx, rx ant, tx, tx ant), paths): the notebook's fft 72 / l_min -6 ... l_max 10 with
    # cyclic prefix 2 and 20 on a few symbols and antennas (the fixture stays below 1 MB), config C4's grid, a tiny multi-link case
    cases = [("cp2", 72, 4, 2, 15e3, 1, (1, 2, 1, 2), 4), ("cp20", 72, 3, 20, 15e3, 1, (1, 2, 1, 2), 4),
             ("c4", 76, 3, 6, 15e3, 1, (1, 4, 1, 2), 3), ("small", 16, 3, 4, 30e3, 2, (2, 2, 2, 1), 3)]
    for tag, fft, nsym, cp, scs, B, (nrx, nra, ntx, nta), P in cases:
        bw = fft * scs
        l_min, l_max = (int(v) for v in cu.time_lag_discrete_time_channel(bw))
        l_tot = l_max - l_min + 1
        N = nsym * (fft + cp)
        x = cn(rng, (B, ntx, nta, nsym, fft))
        a = cn(rng, (B, nrx, nra, ntx, nta, P, N + l_tot - 1), 1.0 / P)
        tau = np.sort(rng.random((B, nrx, ntx, P)) * 9e-7, axis=-1).astype(np.float32)
        out[f"{tag}_meta"] = np.array([fft, nsym, cp, l_min, l_max, B, nrx, nra, ntx, nta, P], np.int32)
        out[f"{tag}_scs"] = np.float64(scs)
        out[f"{tag}_x"], out[f"{tag}_a"], out[f"{tag}_tau"] = x, a, tau
        xt = np.asarray(mod.OFDMModulator(cp)(x))
        out[f"{tag}_x_time"] = xt
        for norm in (True, False):
            h = np.asarray(cu.cir_to_time_channel(bw, tf.constant(a), tf.constant(tau), l_min, l_max, normalize=norm))
            out[f"{tag}_h_time_n{int(norm)}"] = h
            if norm:
                h_keep = h
        y = np.asarray(atc.ApplyTimeChannel(N, l_tot)(xt, h_keep))
        out[f"{tag}_y_time"] = y
        out[f"{tag}_y_rg"] = np.asarray(dem.OFDMDemodulator(fft, l_min, cp)(y))
        # frequency-domain twin of the same CIR, sampled once per OFDM symbol like the notebooks do
        f = np.asarray(cu.subcarrier_frequencies(fft, scs))
        out[f"{tag}_freqs"] = f
        a_f = a[..., cp:-1:(fft + cp)][..., :nsym]
        for norm in (True, False):
            out[f"{tag}_h_freq_n{int(norm)}"] = np.asarray(cu.cir_to_ofdm_channel(tf.constant(f), tf.constant(a_f), tf.constant(tau), normalize=norm))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
