#!/usr/bin/env python3
"""Generates tests/golden/polar_bp_ref_golden.npz by EXECUTING the reference's own ``PolarBPDecoder`` (and
``Polar5GDecoder(dec_type="BP")``) from its unmodified source file fec/polar/decoding.py:1440-1771, 1896-1912 under the
NumPy stand-in for TensorFlow (tools/ref_exec).  exp / log of the boxplus are NumPy's float32 routines there, so the
oracle (oracle/polar_bp.py, math="numpy") must reproduce these outputs bit for bit.

Cases: plain Polar codes from the 5G ranking (n = 32 ... 1024, several rates, soft and hard output, 1 ... 20
iterations) and the 5G chain with rate matching (puncturing, shortening, repetition, downlink).  BPSK + AWGN logits at
noise levels where BP leaves some block errors.  Run here (needs /root/reference); the fixture travels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "polar_bp_ref_golden.npz")

PLAIN = [   # k, n, num_iter, batch, noise sigma
    (16, 32, 20, 12, 0.9),
    (32, 64, 5, 12, 0.8),
    (100, 128, 20, 8, 0.45),
    (64, 256, 1, 8, 1.1),
    (256, 512, 10, 6, 0.88),
    (512, 1024, 20, 6, 0.84),
    (300, 1024, 3, 4, 0.9),
]
FIVEG = [   # k, n, channel type, num_iter, batch, sigma
    (30, 45, "uplink", 20, 12, 0.6),          # shortening
    (32, 70, "uplink", 20, 12, 0.75),         # puncturing
    (29, 127, "uplink", 10, 12, 1.1),         # repetition
    (60, 108, "downlink", 20, 12, 0.6),
    (512, 1024, "uplink", 20, 6, 0.82),       # the code of config C5
]


def load():
    from tools.ref_exec.loader import reference
    ref = reference()
    ref.load_utils()
    ref.load("sionna.phy.fec.ldpc.codes", package_dir=True)
    ref.load("sionna.phy.fec.utils")
    ref.load("sionna.phy.fec.crc")
    ref.load("sionna.phy.fec.polar.codes", package_dir=True)
    pu = ref.load("sionna.phy.fec.polar.utils")
    return pu, ref.load("sionna.phy.fec.polar.encoding"), ref.load("sionna.phy.fec.polar.decoding")


def main():
    import warnings
    warnings.simplefilter("ignore")
    pu, pe, pd = load()
    out = {"plain": np.array([(k, n, it, B) for k, n, it, B, _ in PLAIN], np.int32),
           "fiveg": np.array([(k, n, ct == "downlink", it, B) for k, n, ct, it, B, _ in FIVEG], np.int32)}
    for i, (k, n, it, B, sigma) in enumerate(PLAIN):
        rng = np.random.default_rng(300 + i)
        frozen_pos, info_pos = pu.generate_5g_ranking(k, n)
        enc = pe.PolarEncoder(frozen_pos, n)
        u = rng.integers(0, 2, (B, k)).astype(np.float32)
        c = np.asarray(enc(u))
        logits = ((2 * c - 1) + sigma * rng.normal(size=c.shape)).astype(np.float32) * np.float32(2 / sigma ** 2)
        soft = np.asarray(pd.PolarBPDecoder(frozen_pos, n, num_iter=it, hard_out=False)(logits))
        hard = np.asarray(pd.PolarBPDecoder(frozen_pos, n, num_iter=it, hard_out=True)(logits))
        assert soft.dtype == np.float32 and hard.dtype == np.float32
        print(f"plain k={k} n={n} it={it}: block errors {int(np.sum((hard != u).any(-1)))} of {B}", flush=True)
        for kk, v in dict(frozen_pos=np.asarray(frozen_pos).astype(np.int32), u=np.packbits(u.astype(np.uint8), axis=1),
                          logits=logits, soft=soft, hard=np.packbits(hard.astype(np.uint8), axis=1)).items():
            out[f"p{i}/{kk}"] = v
    for i, (k, n, ct, it, B, sigma) in enumerate(FIVEG):
        rng = np.random.default_rng(400 + i)
        enc = pe.Polar5GEncoder(k, n, channel_type=ct)
        u = rng.integers(0, 2, (B, k)).astype(np.float32)
        c = np.asarray(enc(u))
        logits = ((2 * c - 1) + sigma * rng.normal(size=c.shape)).astype(np.float32) * np.float32(2 / sigma ** 2)
        uh, crc = pd.Polar5GDecoder(enc, dec_type="BP", num_iter=it, return_crc_status=True)(logits)
        uh2 = pd.Polar5GDecoder(enc, dec_type="BP", num_iter=it)(logits)
        uh, crc = np.asarray(uh), np.asarray(crc)
        assert np.array_equal(uh, np.asarray(uh2))
        print(f"5G k={k} n={n} {ct} it={it}: block errors {int(np.sum((uh != u).any(-1)))} of {B}, CRC ok {int(crc.sum())}",
              flush=True)
        for kk, v in dict(u=np.packbits(u.astype(np.uint8), axis=1), logits=logits,
                          u_hat=np.packbits(uh.astype(np.uint8), axis=1), crc=crc.astype(np.uint8)).items():
            out[f"g{i}/{kk}"] = v
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
