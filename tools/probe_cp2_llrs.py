#!/usr/bin/env python3
"""Probe for the open CP-2 discrepancy: run the product chain of tests/notebook_curves.py:_CdlModel
(CDL-C uplink, LS CSI, cyclic prefix 2, time domain) at one Eb/N0 on the GPU and bring back the decoder INPUT (LLRs) of
every block the product's decoder gets wrong, plus a sample of the blocks it gets right - so that the reference's own
decoder code (NumPy exp / log under tools/ref_exec) can be run on exactly those LLRs on the CPU.

    python tools/probe_cp2_llrs.py [ebno_db] [examples per batch] [batches]   -> gpurun_out/cp2_llrs.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import notebook_curves as nc
    import sionna_amd.phy as phy
    ebno = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    m = nc._CdlModel("time", "C", False, 3.0, 2, [2, 11])
    phy.config.seed = 4321
    fails, goods, n_blocks, n_fail = [], [], 0, 0
    for it in range(iters):
        rg = m.rg
        no = phy.utils.ebnodb2no(ebno, m.m, m.coderate, rg)
        b = m.source([batch, 1, m.n_ut, m.k])
        x_rg = m.rg_mapper(m.mapper(m.encoder(b)))
        a, tau = m.cdl(batch, rg.num_time_samples + m.l_tot - 1, rg.bandwidth)
        h_time = phy.channel.cir_to_time_channel(rg.bandwidth, a, tau, l_min=m.l_min, l_max=m.l_max, normalize=True)
        del a
        y = m.demodulator(m.channel_time(m.modulator(x_rg), h_time, no))
        del h_time
        h_hat, err_var = m.ls_est(y, no)
        x_hat, no_eff = m.lmmse(y, h_hat, err_var, no)
        llr = m.demapper(x_hat, no_eff)
        b_hat = m.decoder(llr)
        bt, bh, lt = (t.as_subclass(torch.Tensor).reshape(-1, t.shape[-1]) for t in (b, b_hat, llr))
        bad = (bt != bh).any(-1)
        n_blocks += bt.shape[0]
        n_fail += int(bad.sum())
        idx = torch.nonzero(bad).reshape(-1)
        fails.append((lt[idx].cpu().numpy(), bt[idx].cpu().numpy().astype(np.uint8), bh[idx].cpu().numpy().astype(np.uint8)))
        if it < 4:
            ok = torch.nonzero(~bad).reshape(-1)[:1000]
            goods.append((lt[ok].cpu().numpy(), bt[ok].cpu().numpy().astype(np.uint8)))
        print(f"batch {it}: blocks {n_blocks} block errors {n_fail} BLER {n_fail / n_blocks:.5f}", flush=True)
    out = os.path.join(ROOT, "gpurun_out", "cp2_llrs.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, ebno=ebno, n_blocks=n_blocks, n_fail=n_fail, k=m.k, n=m.n,
                        llr_fail=np.concatenate([f[0] for f in fails]), b_fail=np.packbits(np.concatenate([f[1] for f in fails]), axis=1),
                        bh_fail=np.packbits(np.concatenate([f[2] for f in fails]), axis=1),
                        llr_ok=np.concatenate([g[0] for g in goods]), b_ok=np.packbits(np.concatenate([g[1] for g in goods]), axis=1))
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
