"""A/B of the OFDM LMMSE kernels on the C4 shapes with SAMD_LMMSE_ITEMS = 1 (one (batch, receiver) pair per thread: rounds 2-4)
against 2 / 4 / 8 pairs per thread, in ONE process: the per-RE equaliser through the C-ABI on prepared arguments (kernel-only
time, 20 launches between one pair of HIP events) and the fused LS-NN + LMMSE + demapper front end through the blocks.
Outputs must be bit-identical.

Result (profiles/r05r2_lmmse_pairs_per_thread.txt): 184-200 us with 2-8 pairs per thread against 189-204 us with one - the table
round trip is not what the kernel waits for, and the loop form needs 88-106 registers (5 / 4 waves per SIMD instead of 7 / 6).
The walk kernels were NOT kept (commit history of csrc/mimo.hip, round 5): SAMD_LMMSE_ITEMS is no longer read by the library, so
this script now times the same kernel six times; kept as the record of how the experiment was run."""
import os
import sys

import numpy as np
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
    est_mat, est = phy.ofdm.LSChannelEstimator(rg, defer=False), phy.ofdm.LSChannelEstimator(rg)
    eq = phy.ofdm.LMMSEEqualizer(rg, sm)
    det = phy.ofdm.LinearDetector("lmmse", "bit", "app", rg, sm, constellation_type="qam", num_bits_per_symbol=m)
    h_hat, ev = est_mat(y, no)
    keep, head, tabs, dims = eq._prepare(y, h_hat, ev, no)
    xk = torch.empty((B, 1, 2, rg.num_data_symbols), dtype=torch.complex64, device=y.device)
    nk = torch.empty((B, 1, 2, rg.num_data_symbols), dtype=torch.float32, device=y.device)
    lib, st = _ffi.lib(), _ffi.stream()

    def kern():
        return lib.samd_ofdm_lmmse_c64(*head, *tabs, *dims, int(eq._mode), _ffi.ptr(xk), _ffi.ptr(nk), st)

    def fused():
        hh, evv = est(y, no)
        return det(y, hh, evv, no)

    ref = {}
    for items in (1, 2, 4, 8, 1, 4):
        _ffi.set_option("SAMD_LMMSE_ITEMS", str(items))
        try:
            ms_k = timed(kern)
            x1, n1 = xk.clone(), nk.clone()
            ms_f = timed(fused, 10)
            l1 = fused().as_subclass(torch.Tensor).clone()
        finally:
            _ffi.set_option("SAMD_LMMSE_ITEMS", None)
        if not ref:
            ref = {"x": x1, "n": n1, "l": l1}
        same = bool(torch.equal(x1, ref["x"]) and torch.equal(n1, ref["n"]) and torch.equal(l1, ref["l"]))
        print(f"pairs per thread {items}:  LMMSE kernel {ms_k * 1e3:7.1f} us   fused front end (block call) {ms_f * 1e3:7.1f} us   "
              f"{'bit-identical to the first' if same else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    main()
