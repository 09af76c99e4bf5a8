"""A/B of the cir_to_ofdm kernels on the C4 shapes in ONE process (development options read at launch):
python tools/c2o_ab.py name:OPT=val[,OPT=val] ...   - every variant is timed (HIP events, 20 launches) and compared with the
first one at 1e-5 of the output's scale (the variants group the path sum differently)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    phy.config.seed = 4
    B = 8192
    tdl = phy.channel.tr38901.TDL("A", 300e-9, 2.6e9, min_speed=10., num_rx_ant=4, num_tx_ant=2)
    freqs = phy.channel.subcarrier_frequencies(76, 15e3)
    a_t, tau_t = tdl(B, 14, 1.0 / 71.4e-6)
    ref = None
    a_d, tau_d = a_t.as_subclass(torch.Tensor).contiguous(), tau_t.as_subclass(torch.Tensor).contiguous()
    fr_d = _ffi.to_device(np.asarray(freqs, np.float32), torch.float32)
    h_d = torch.empty((B, 1, 4, 1, 2, 14, 76), dtype=torch.complex64, device="cuda")
    buf = torch.empty((B, 1, 4, 1, 2, 14, 76), dtype=torch.complex64, device="cuda")       # the output's size: what a plain fill achieves
    for _ in range(3):
        buf.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        buf.fill_(1.0)
    e1.record()
    torch.cuda.synchronize()
    print(f"{'(fill of the output tensor)':28s}              {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
    for spec in (sys.argv[1:] or ["default:"]) * 2:                     # every variant twice: the first pass warms the clocks
        name, _, opts = spec.partition(":")
        kv = [o.split("=") for o in opts.split(",") if o]
        for k, v in kv:
            _ffi.set_option(k, v)
        try:
            for norm in (True, False):
                def fn():                     # the C-ABI entry itself on resident buffers: no host work between launches
                    _ffi.check(_ffi.lib().samd_cir_to_ofdm_c64(_ffi.ptr(a_d), _ffi.ptr(tau_d), _ffi.ptr(fr_d), B, 1, 4, 1, 2, a_d.shape[-2],
                                                               14, 76, int(norm), _ffi.ptr(h_d), _ffi.stream()), "cir_to_ofdm")
                    return h_d
                h = fn(); fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 20
                hv = h.as_subclass(torch.Tensor)
                if ref is None:
                    ref = {}
                if norm not in ref:
                    ref[norm] = hv.clone()
                err = float((hv - ref[norm]).abs().max() / ref[norm].abs().max())
                print(f"{name:28s} normalize={int(norm)}  {ms * 1e3:8.1f} us  max rel diff to first {err:.2e}", flush=True)
        finally:
            for k, _ in kv:
                _ffi.set_option(k, None)


if __name__ == "__main__":
    main()
