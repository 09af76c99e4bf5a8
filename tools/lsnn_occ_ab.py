"""Timing of the fused LS-NN + LMMSE + demapper front end (ofdm_lsnn_lmmse_kernel, csrc/mimo.hip) on the C4 shapes for every
square QAM and both demapping methods; one line per (modulation, method): microseconds per block call (20 between one pair of HIP
events; the QPSK / 16-QAM lines are bounded by the host's ~320 us per call pair, the kernel alone is 190 us) and a checksum of the
LLRs.  Run once per library build (SAMD_LIB=...) to compare builds.

Record (profiles/r05_lsnn_occ_ab.txt, round 5): the kernel compiled for 6 waves per SIMD everywhere (the 256-QAM and max-log forms
spill 8-43 registers) against 4 waves for 256-QAM and 5 for max-log (`lsnn_waves` in csrc/mimo.hip, kept): 256-QAM app 810 -> 535 us,
max-log 554 -> 430 us, 64-QAM max-log 437 -> 369 us, 16-QAM max-log 389 -> 345 us, same bits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=20):
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
    print("library:", _ffi.LIB_PATH)
    B, k, n = 8192, 768, 1536
    rg = phy.ofdm.ResourceGrid(14, 76, 15e3, num_tx=1, num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                               dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    sm = phy.mimo.StreamManagement([[1]], 2)
    tdl = phy.channel.tr38901.TDL("A", 300e-9, 2.6e9, min_speed=10., num_rx_ant=4, num_tx_ant=2)
    ch = phy.channel.OFDMChannel(tdl, rg, normalize_channel=True, return_channel=False)
    est = phy.ofdm.LSChannelEstimator(rg)
    for rep in range(2):                                          # the first pass warms the clocks
        for m in (2, 4, 6, 8):
            phy.config.seed = 4
            no = phy.utils.ebnodb2no(10.0, m, k / n, rg)
            x = phy.mapping.QAMSource(m)([B, 1, 2, rg.num_data_symbols])
            y = ch(phy.ofdm.ResourceGridMapper(rg)(x), no)
            for method in ("app", "maxlog"):
                det = phy.ofdm.LinearDetector("lmmse", "bit", method, rg, sm, constellation_type="qam", num_bits_per_symbol=m)

                def fused():
                    hh, evv = est(y, no)
                    return det(y, hh, evv, no)

                us = timed(fused) * 1e3
                llr = fused().as_subclass(torch.Tensor)
                if rep:
                    chk = int(torch.sum(llr.view(torch.int32).to(torch.int64)).item()) & 0xFFFFFFFF
                    print(f"{1 << m:4d}-QAM {method:6s}  {us:7.1f} us   checksum {chk:08x}", flush=True)


if __name__ == "__main__":
    main()
