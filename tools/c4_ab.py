"""A/B of the round-5 C4 kernels against their round-4 forms in ONE process (development options read at launch):
cir_to_ofdm with results staged in registers vs the two-pass kernel (SAMD_C2O_TWO_PASS), the per-RE LMMSE equaliser and the
fused LS-NN + LMMSE (+ demapper) front end with two resource elements per lane vs one (the experiment is behind SAMD_LMMSE_R2: it was slower).  LMMSE outputs must be
bit-identical; the channel transform is compared at 1e-5 of its scale (different grouping of the normalisation sum)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(reps):
        fn()
    a1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(a1) / reps


def main():
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    phy.config.seed = 4
    B, k, n, m = 8192, 768, 1536, 2
    rg = phy.ofdm.ResourceGrid(14, 76, 15e3, num_tx=1, num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                               dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    sm = phy.mimo.StreamManagement([[1]], 2)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    tdl = phy.channel.tr38901.TDL("A", 300e-9, 2.6e9, min_speed=10., num_rx_ant=4, num_tx_ant=2)
    ch = phy.channel.OFDMChannel(tdl, rg, normalize_channel=True, return_channel=True)
    no = phy.utils.ebnodb2no(10.0, m, k / n, rg)
    b = phy.mapping.BinarySource()([B, 1, 2, k])
    y, h = ch(phy.ofdm.ResourceGridMapper(rg)(phy.mapping.Mapper("qam", m)(enc(b))), no)
    est_mat = phy.ofdm.LSChannelEstimator(rg, defer=False)
    est = phy.ofdm.LSChannelEstimator(rg)
    eq = phy.ofdm.LMMSEEqualizer(rg, sm)
    det = phy.ofdm.LinearDetector("lmmse", "bit", "app", rg, sm, constellation_type="qam", num_bits_per_symbol=m)
    h_hat, ev = est_mat(y, no)
    fs = 1.0 / rg.ofdm_symbol_duration
    freqs = phy.channel.subcarrier_frequencies(rg.fft_size, rg.subcarrier_spacing)
    a_t, tau_t = tdl(B, rg.num_ofdm_symbols, fs)
    out = {}

    def T(x):
        return x.as_subclass(torch.Tensor)

    def fused():
        hh, evv = est(y, no)
        return det(y, hh, evv, no)

    rows = []
    for name, fn, opt, exact in (("cir_to_ofdm", lambda: phy.channel.cir_to_ofdm_channel(freqs, a_t, tau_t, normalize=True), "SAMD_C2O_TWO_PASS", False),
                                 ("lmmse_equalizer", lambda: eq(y, h_hat, ev, no)[0], "!SAMD_LMMSE_R2", True),
                                 ("fused_front_end", fused, "!SAMD_LMMSE_R2", True)):
        # opt names the switch that selects the round-4 form; "!opt": the switch selects the round-5 experiment instead
        def run_with(flag):
            if flag:
                _ffi.set_option(opt.lstrip("!"), "1")
            try:
                return T(fn()).clone(), timed(fn)
            finally:
                if flag:
                    _ffi.set_option(opt.lstrip("!"), None)
        new, ms_new = run_with(opt.startswith("!"))
        old, ms_old = run_with(not opt.startswith("!"))
        if exact:
            same = bool(torch.equal(new, old))
        else:
            same = float((new - old).abs().max() / old.abs().max())
        rows.append({"kernel": name, "ms_round4_form": round(ms_old, 4), "ms_round5_form": round(ms_new, 4), "agreement": same})
        print(f"{name:18s} round-4 form {ms_old * 1e3:8.1f} us   round-5 form {ms_new * 1e3:8.1f} us   "
              f"{'bit-identical' if same is True else same}", flush=True)
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
