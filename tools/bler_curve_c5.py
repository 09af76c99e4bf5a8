#!/usr/bin/env python3
"""BLER versus Eb/N0 of config C5 (Polar5G uplink k=512 n=1024, CRC-aided SCL list 8, QPSK over AWGN) on the GPU,
with the oracle cross-check: at every SNR point the C oracle (oracle/polar_scl.c, float32 specification arithmetic)
decodes a sample of the SAME device LLRs - hard decisions and CRC status must agree bit for bit, so the two curves
coincide (gap 0 dB by construction); the float64 instantiation of the oracle (the reference NumPy twin's arithmetic)
decodes the same sample as the reference-side witness, whose BLER is listed next to it.
Test / evidence tooling (uses oracle/).

    python tools/bler_curve_c5.py --out profiles/r02b_bler_c5.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bler_c5.json"))
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--num-batches", type=int, default=4)
    ap.add_argument("--oracle-sample", type=int, default=4096)
    args = ap.parse_args()

    import sionna_amd.phy as phy
    from oracle import polar as op, polar_c as pc

    k, n, m = 512, 1024, 2
    phy.config.seed = 20260924
    enc = phy.fec.polar.Polar5GEncoder(k, n)
    dec = phy.fec.polar.Polar5GDecoder(enc, "SCL", list_size=8, return_crc_status=True)
    src, mapper = phy.mapping.BinarySource(), phy.mapping.Mapper("qam", m)
    demap, chan = phy.mapping.Demapper("app", "qam", m), phy.channel.AWGN()
    ocode = op.Polar5GCode(k, n)
    ebnos = [float(x) for x in np.arange(0.0, 3.01, 0.25)]
    res = {"config": "C5: Polar5G uplink k=512 n=1024 (CRC11), SCL list 8, QPSK, AWGN", "batch_size": args.batch,
           "num_batches": args.num_batches, "oracle_sample": args.oracle_sample, "ebno_db": ebnos, "points": []}
    t0 = time.time()
    for e in ebnos:
        no = phy.utils.ebnodb2no(e, m, k / n)
        blk_err = bit_err = crc_fail = total = 0
        for it in range(args.num_batches):
            b = src([args.batch, k])
            llr = demap(chan(mapper(enc(b)), no), no)
            b_hat, status = dec(llr)
            wrong = (b != b_hat)
            blk_err += int(wrong.any(-1).sum())
            bit_err += int(wrong.sum())
            crc_fail += int((~status.bool()).sum())
            total += args.batch
        ns = args.oracle_sample
        llr_s = llr[:ns].cpu().numpy()
        ref32, st32 = pc.polar5g_decode(ocode, llr_s, list_size=8, precision="f32", return_crc_status=True)
        ref64, st64 = pc.polar5g_decode(ocode, llr_s, list_size=8, precision="f64", return_crc_status=True)
        b_s, got_s, stg = b[:ns].cpu().numpy(), b_hat[:ns].cpu().numpy(), status[:ns].cpu().numpy().astype(bool)
        pt = {"ebno_db": e, "bler": blk_err / total, "ber": bit_err / (total * k), "crc_fail_rate": crc_fail / total,
              "codewords": total,
              "oracle_f32": {"identical_codewords": int(np.all(got_s == ref32, axis=1).sum()), "of": ns,
                             "crc_status_identical": bool(np.array_equal(stg, st32.astype(bool))),
                             "bler": float(np.any(ref32 != b_s, axis=1).mean())},
              "gpu_bler_on_sample": float(np.any(got_s != b_s, axis=1).mean()),
              "oracle_f64_reference_twin": {"identical_codewords": int(np.all(got_s == ref64, axis=1).sum()),
                                            "bler": float(np.any(ref64 != b_s, axis=1).mean())}}
        res["points"].append(pt)
        print(json.dumps(pt), flush=True)
    res["seconds"] = round(time.time() - t0, 1)
    res["all_bit_exact_vs_oracle_f32"] = all(p["oracle_f32"]["identical_codewords"] == p["oracle_f32"]["of"] and
                                             p["oracle_f32"]["crc_status_identical"] for p in res["points"])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out, "bit-exact:", res["all_bit_exact_vs_oracle_f32"])


if __name__ == "__main__":
    main()
