#!/usr/bin/env python3
"""Probe of the one published curve the MI355X path does not reproduce: MIMO_OFDM_Transmissions_over_CDL.ipynb cell 76,
CDL-C uplink, LS CSI, cyclic prefix 2, TIME domain (the ISI-limited regime).  Varies what the ISI floor is sensitive to -
the number of precursor lags l_min (= how many samples the strongest tap sits behind the FFT window's start) and the delay
spread - and prints the BLER at the high-SNR points next to the reference's, to see which variant (if any) the saved table
corresponds to.  Diagnostic only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import notebook_curves as nc


def main():
    import sionna_amd.phy as phy
    ref = nc.load_tables()["MIMO_OFDM_Transmissions_over_CDL/c76/t3"]["rows"]
    pts = [(8.0, 4), (12.0, 6), (16.0, 8)]
    print("reference:", {e: f"{ref[i]['bler']:.4e}" for e, i in pts})
    variants = (("as published (l_min -6)", {}), ("l_min -5", {"l_min_override": -5}), ("l_min -4", {"l_min_override": -4}),
                ("l_min -7", {"l_min_override": -7}), ("delay spread 50 ns", {"delay_spread": 50e-9}),
                ("delay spread 200 ns", {"delay_spread": 200e-9}))
    if "--decoders" in sys.argv:      # the same link, other check-node arithmetic: is the ISI floor a property of the decoder's numerics?
        variants = (("boxplus-phi (defined exp/log)", {}), ("boxplus-phi-fast (hardware exp/log)", {"cn_update": "boxplus-phi-fast"}),
                    ("boxplus (tanh)", {"cn_update": "boxplus"}), ("minsum", {"cn_update": "minsum"}),
                    ("offset-minsum", {"cn_update": "offset-minsum"}))
    for name, kw in variants:
        phy.config.seed = 7
        m = nc._CdlModel(domain="time", cdl_model="C", perfect_csi=False, speed=3.0, cyclic_prefix_length=2,
                         pilot_ofdm_symbol_indices=[2, 11], **kw)
        out = {}
        for e, _ in pts:
            err = blk = 0
            while blk < 60000 and err < 1500:
                b, bh = m(1024, e)
                d = (b != bh).reshape(-1, b.shape[-1]).any(-1)
                err += int(d.sum()); blk += d.numel()
            out[e] = f"{err / blk:.4e}"
        print(f"{name:28s}", out, flush=True)


if __name__ == "__main__":
    main()
