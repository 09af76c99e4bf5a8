#!/usr/bin/env python3
"""Generates tests/golden/polar5g_ref_golden.npz by EXECUTING the reference's own 5G Polar chain under the NumPy stand-in
for TensorFlow (tools/ref_exec), from its unmodified source files:

    fec/crc.py                 CRCEncoder / CRCDecoder
    fec/polar/utils.py         generate_5g_ranking
    fec/polar/encoding.py      PolarEncoder :19-228, Polar5GEncoder :231-740 (CRC, input interleaver, sub-block
                               interleaver, puncturing / shortening / repetition, channel interleaver)
    fec/polar/decoding.py      PolarSCDecoder :22-303, PolarSCLDecoder :306-1404 - the TENSORFLOW list decoder
                               (:919-1045; ``cpu_only=False``) AND its NumPy twin (:1047-1290; ``cpu_only=True``) -,
                               Polar5GDecoder :1750-2086 (rate recovery, CRC-aided list selection, CRC status)

Cases: config C5 (k=512, n=1024, uplink, SCL-8) and the rate-matching regimes of the reference's own encoder vectors
(puncturing, shortening, repetition; uplink and downlink).  The received logits are the codeword through BPSK + AWGN at a
noise level where the list matters (some blocks fail under SC, fewer under SCL-8).  Stored: bits, codewords, logits, the
decisions of every decoder and the CRC status.  Run here (needs /root/reference); the fixture travels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "polar5g_ref_golden.npz")

CASES = [   # k, n, channel type, batch, noise sigma
    (512, 1024, "uplink", 12, 0.86),          # config C5
    (30, 45, "uplink", 16, 0.70),             # shortening
    (32, 70, "uplink", 16, 0.85),             # puncturing
    (29, 127, "uplink", 16, 1.30),            # repetition
    (400, 1023, "uplink", 8, 1.02),
    (12, 60, "uplink", 16, 1.5),              # CRC6
    (60, 108, "downlink", 16, 0.70),          # CRC24C + input interleaver, no channel interleaver
    (140, 576, "downlink", 8, 1.3),
]


def load():
    from tools.ref_exec.loader import reference
    ref = reference()
    ref.load_utils()
    ref.load("sionna.phy.fec.ldpc.codes", package_dir=True)
    ref.load("sionna.phy.fec.utils")
    ref.load("sionna.phy.fec.crc")
    ref.load("sionna.phy.fec.polar.codes", package_dir=True)
    ref.load("sionna.phy.fec.polar.utils")
    return ref.load("sionna.phy.fec.polar.encoding"), ref.load("sionna.phy.fec.polar.decoding")


def main():
    import warnings
    warnings.simplefilter("ignore")
    pe, pd = load()
    out = {"cases": np.array([(k, n, ct == "downlink", B) for k, n, ct, B, _ in CASES], np.int32)}
    for i, (k, n, ct, B, sigma) in enumerate(CASES):
        rng = np.random.default_rng(100 + i)
        enc = pe.Polar5GEncoder(k, n, channel_type=ct)
        u = rng.integers(0, 2, (B, k)).astype(np.float32)
        c = np.asarray(enc(u))
        logits = ((2 * c - 1) + sigma * rng.normal(size=c.shape)).astype(np.float32) * np.float32(2 / sigma ** 2)
        o = dict(u=np.packbits(u.astype(np.uint8), axis=1), c=np.packbits(c.astype(np.uint8), axis=1), logits=logits,
                 frozen_pos=np.asarray(enc.frozen_pos).astype(np.int32), n_polar=np.int32(enc.n_polar))
        msg = f"k={k} n={n} {ct}: block errors"
        for name, kw in (("sc", dict(dec_type="SC")), ("scl8_tf", dict(dec_type="SCL", list_size=8)),
                         ("scl8_np", dict(dec_type="SCL", list_size=8, cpu_only=True)),
                         ("scl4_tf", dict(dec_type="SCL", list_size=4)), ("hyb8", dict(dec_type="hybSCL", list_size=8))):
            if name in ("scl4_tf",) and n > 600:
                continue
            dec = pd.Polar5GDecoder(enc, return_crc_status=(name != "sc"), **kw)
            r = dec(logits)
            if name != "sc":
                r, crc = r
                o[f"crc_{name}"] = np.asarray(crc).astype(np.uint8)
            uh = np.asarray(r)
            o[f"u_hat_{name}"] = np.packbits(uh.astype(np.uint8), axis=1)
            msg += f"  {name} {int(np.sum((uh != u).any(-1)))}"
        print(msg, f"of {B}", flush=True)
        for kk, v in o.items():
            out[f"{i}/{kk}"] = v
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
