#!/usr/bin/env python3
"""One batch of the CDL-C / cyclic-prefix-2 / time-domain chain of MIMO_OFDM_Transmissions_over_CDL.ipynb cell 76, every
stage's output as the reference's OWN code produced it under the NumPy stand-in (gpurun_in/cp2_chain.npz, written by a
scratch script around tools/ref_exec), against this build stage by stage - each stage fed with the REFERENCE's input to
it, and the chain end to end.  ``--oracle`` runs oracle/ (CPU) instead of the HIP blocks.  Probe for the one open BER
discrepancy."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    a, b = np.broadcast_arrays(a, b) if a.ndim == b.ndim else (a.reshape(b.shape), b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    g = np.load(os.path.join(ROOT, "gpurun_in", "cp2_chain.npz"))
    cp, l_min, l_max, N, k, n = (int(v) for v in g["meta"])
    no = float(g["no"])
    kw = dict(num_tx=1, num_streams_per_tx=4, cyclic_prefix_length=cp, num_guard_carriers=[5, 6], dc_null=True,
              pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    bw = 72 * 15e3
    if "--oracle" in sys.argv:
        from oracle import ofdm as o, mapping as om, ldpc_bp as obp
        from oracle.ldpc5g import LDPC5GCode
        rg = o.ResourceGrid(14, 72, 15e3, **{**kw, "num_guard_carriers": (5, 6)})
        rg.pilot_pattern._pilots = g["pilots"]
        sm = o.StreamManagement(np.array([[1]]), 4)
        code = LDPC5GCode(k, n)
        pts = om.qam(2)
        c = code.encode(g["b"].astype(np.float32).reshape(-1, k)).reshape(g["c"].shape)
        print("encoder      equal:", np.array_equal(c, g["c"]))
        x_rg = o.rg_map(rg, om.mapper(g["c"].astype(np.float32), pts))
        print("mapper + grid equal:", np.array_equal(x_rg, g["x_rg"]))
        print("modulator    ", rel(o.ofdm_modulate(g["x_rg"], cp), g["x_time"]))
        h_time = o.cir_to_time_channel(bw, g["a"], g["tau"], l_min, l_max, normalize=True)
        print("cir->time    ", rel(h_time, g["h_time"]))
        y_clean = o.apply_time_channel(g["x_time"], g["h_time"])
        print("time channel ", rel(y_clean, g["y_clean"]))
        y = o.ofdm_demodulate((g["y_clean"] + g["noise"]).astype(np.complex64), 72, l_min, cp)
        print("demodulator  ", rel(y, g["y"]))
        h_hat, ev = o.ls_estimate(rg, g["y"], np.float32(no))
        print("LS           ", rel(h_hat, g["h_hat"]), rel(ev, g["err_var"]))
        x_hat, no_eff = o.ofdm_lmmse_equalize(rg, sm, g["y"], g["h_hat"], g["err_var"], np.float32(no))
        print("LMMSE        ", rel(x_hat, g["x_hat"]), rel(no_eff, g["no_eff"]))
        llr = om.demapper(g["x_hat"], g["no_eff"], pts, "app")
        print("demapper     ", rel(llr, g["llr"]))
        d = obp.LDPC5GDecoder(code, "boxplus-phi", hard_out=True, return_infobits=True, num_iter=20)
        bh = d.decode5g(g["llr"].reshape(-1, n)).reshape(g["b_hat"].shape)
        print("decoder      equal:", np.array_equal(bh.astype(np.uint8), g["b_hat"]))
        # end to end
        y2 = o.ofdm_demodulate((o.apply_time_channel(o.ofdm_modulate(x_rg, cp), h_time) + g["noise"]).astype(np.complex64), 72, l_min, cp)
        h2, e2 = o.ls_estimate(rg, y2, np.float32(no))
        x2, n2 = o.ofdm_lmmse_equalize(rg, sm, y2, h2, e2, np.float32(no))
        l2 = om.demapper(x2.astype(np.complex64), n2.astype(np.float32), pts, "app")
        print("end to end: y", rel(y2, g["y"]), "x_hat", rel(x2, g["x_hat"]), "llr", rel(l2, g["llr"]))
        return
    import torch
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    npy = lambda t: t.detach().cpu().numpy()
    rg = phy.ofdm.ResourceGrid(num_ofdm_symbols=14, fft_size=72, subcarrier_spacing=15e3, **kw)
    rg.pilot_pattern.pilots = g["pilots"]
    sm = phy.mimo.StreamManagement(np.array([[1]]), 4)
    assert phy.channel.time_lag_discrete_time_channel(rg.bandwidth) == (l_min, l_max) and rg.num_time_samples == N
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, hard_out=True)
    b = g["b"].astype(np.float32)
    print("encoder      equal:", np.array_equal(npy(enc(b)), g["c"]))
    x_rg = phy.ofdm.ResourceGridMapper(rg)(phy.mapping.Mapper("qam", 2)(g["c"].astype(np.float32)))
    print("mapper + grid equal:", np.array_equal(npy(x_rg), g["x_rg"]))
    mod, dem = phy.ofdm.OFDMModulator(cp), phy.ofdm.OFDMDemodulator(72, l_min, cp)
    print("modulator    ", rel(npy(mod(g["x_rg"])), g["x_time"]))
    h_time = phy.channel.cir_to_time_channel(rg.bandwidth, g["a"], g["tau"], l_min=l_min, l_max=l_max, normalize=True)
    print("cir->time    ", rel(npy(h_time), g["h_time"]))
    l_tot = l_max - l_min + 1
    ch_clean = phy.channel.ApplyTimeChannel(N, l_tot=l_tot, add_awgn=False)
    print("time channel ", rel(npy(ch_clean(g["x_time"], g["h_time"])), g["y_clean"]))
    print("demodulator  ", rel(npy(dem((g["y_clean"] + g["noise"]).astype(np.complex64))), g["y"]))
    ls = phy.ofdm.LSChannelEstimator(rg, interpolation_type="nn")
    h_hat, ev = ls(g["y"], np.float32(no))
    print("LS           ", rel(npy(h_hat), g["h_hat"]), rel(npy(ev), g["err_var"]))
    lm = phy.ofdm.LMMSEEqualizer(rg, sm)
    x_hat, no_eff = lm(g["y"], g["h_hat"], g["err_var"], np.float32(no))
    print("LMMSE        ", rel(npy(x_hat), g["x_hat"]), rel(npy(no_eff), g["no_eff"]))
    dm = phy.mapping.Demapper("app", "qam", 2)
    print("demapper     ", rel(npy(dm(g["x_hat"], g["no_eff"])), g["llr"]))
    print("decoder      equal:", np.array_equal(npy(dec(g["llr"])).astype(np.uint8), g["b_hat"]))
    # end to end, the way tests/notebook_curves.py:_CdlModel chains the blocks (noise added by hand)
    y2 = dem((ch_clean(mod(x_rg), h_time).as_subclass(torch.Tensor) + torch.from_numpy(g["noise"]).to(_ffi.device())))
    h2, e2 = ls(y2, np.float32(no))
    x2, n2 = lm(y2, h2, e2, np.float32(no))
    l2 = dm(x2, n2)
    print("end to end: y", rel(npy(y2), g["y"]), "x_hat", rel(npy(x2), g["x_hat"]), "llr", rel(npy(l2), g["llr"]),
          "decisions equal:", np.array_equal(npy(dec(l2)).astype(np.uint8), g["b_hat"]))
    # the channel block's own noise: its variance against `no`
    ch = phy.channel.ApplyTimeChannel(N, l_tot=l_tot, add_awgn=True)
    yn = npy(ch(g["x_time"], g["h_time"], np.float32(no))) - g["y_clean"]
    print("channel noise variance / no:", float(np.mean(np.abs(yn) ** 2) / no), " (reference adds complex noise of variance no)")


if __name__ == "__main__":
    main()
