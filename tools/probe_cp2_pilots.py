#!/usr/bin/env python3
"""Probe for the open CP-2 discrepancy: does the error floor of the CDL-C / LS-CSI / cyclic-prefix-2 / time-domain link
depend on WHICH pilot sequence the simulation drew?  The Kronecker pilots are random QPSK symbols drawn ONCE, when the
resource grid is built; with a 2-sample prefix the LS estimates carry inter-symbol / inter-carrier interference from the
neighbouring pilots and data, a deterministic function of that one sequence.  Runs tests/notebook_curves.py:_CdlModel at
one Eb/N0 with the pilots of (a) this build's Philox stream (several seeds), (b) NumPy generators, (c) the sequence the
reference-executed chain used (tests/golden/cp2_ref_exec_mc.npz: BLER 0.0082 there).

    python tools/probe_cp2_pilots.py [ebno_db] [examples per batch] [batches]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bler(m, phy, ebno, batch, iters, seed):
    import torch
    phy.config.seed = seed
    e = nb = 0
    for _ in range(iters):
        b, bh = m(batch, ebno)
        bt, bht = (t.as_subclass(torch.Tensor).reshape(-1, t.shape[-1]) for t in (b, bh))
        e += int((bt != bht).any(-1).sum())
        nb += bt.shape[0]
    return e, nb


def main():
    import notebook_curves as nc
    import sionna_amd.phy as phy
    ebno = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 16

    def model(pilots=None, pilot_seed=None):
        m = nc._CdlModel("time", "C", False, 3.0, 2, [2, 11])
        if pilot_seed is not None:
            m.rg.pilot_pattern = phy.ofdm.KroneckerPilotPattern(m.rg, [2, 11], seed=pilot_seed)
        if pilots is not None:
            m.rg.pilot_pattern.pilots = pilots
        if pilots is not None or pilot_seed is not None:       # the blocks that read the pilots
            m.rg_mapper = phy.ofdm.ResourceGridMapper(m.rg)
            m.ls_est = phy.ofdm.LSChannelEstimator(m.rg, interpolation_type="nn")
            m.lmmse = phy.ofdm.LMMSEEqualizer(m.rg, m.sm)
        return m

    def qpsk(rng, mask_of):
        p = ((1 - 2 * rng.integers(0, 2, mask_of.shape)) + 1j * (1 - 2 * rng.integers(0, 2, mask_of.shape))).astype(np.complex64) / np.sqrt(2)
        return np.where(mask_of != 0, p, 0).astype(np.complex64)

    base = model()
    own = np.asarray(base.rg.pilot_pattern._pilots)
    rows = [("this build's default pilots (Philox seed 0)", base)]
    rows += [(f"Philox pilot seed {s}", model(pilot_seed=s)) for s in (1, 2, 3)]
    rows += [(f"NumPy QPSK pilots, generator {s}", model(pilots=qpsk(np.random.default_rng(100 + s), own))) for s in (0, 1, 2, 3)]
    ref_p = np.load(os.path.join(ROOT, "tests", "golden", "cp2_ref_exec_mc.npz"))["pilots"]
    rows.append(("the reference-executed chain's pilots (BLER 0.0085 +- 0.0005 there)", model(pilots=ref_p)))
    rows.append(("all pilots equal (1+j)/sqrt(2)", model(pilots=np.where(own != 0, (1 + 1j) / np.sqrt(2), 0).astype(np.complex64))))
    print(f"CDL-C uplink, LS-NN CSI, cyclic prefix 2, time domain, {ebno} dB; notebook table: BLER 5.88e-3 at 16 dB")
    for name, m in rows:
        e, nb = bler(m, phy, ebno, batch, iters, seed=777)
        print(f"  {name:72s} block errors {e:5d} / {nb}  BLER {e / nb:.5f} +- {np.sqrt(max(e, 1)) / nb:.5f}", flush=True)


if __name__ == "__main__":
    main()
