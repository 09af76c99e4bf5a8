#!/usr/bin/env python3
"""BER / BLER versus Eb/N0 of config C2 (BG1 k=2816 n=8448, 64-QAM, AWGN, flooding BP
20 iterations) on the GPU through ``sim_ber``, for the north-star min-sum rule (on-chip engine) and the
reference's default boxplus-phi rule (HBM-resident engine), plus the oracle cross-check: at every
SNR point the CPU oracle (oracle/ldpc_bp.c, min-sum) decodes a sample of the SAME device LLRs and
must agree bit for bit.  Test / evidence tooling (uses oracle/); writes one JSON file.

    python tools/ber_curve.py --out profiles/r01_ber_c2.json
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
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ber_c2.json"))
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--max-mc-iter", type=int, default=8)
    ap.add_argument("--oracle-sample", type=int, default=512)
    args = ap.parse_args()

    import sionna_amd.phy as phy
    from oracle.ldpc5g import LDPC5GCode
    from oracle import ldpc_bp as obp, cbind

    k, n, m = 2816, 8448, 6
    phy.config.seed = 20260923
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    src, mapper = phy.mapping.BinarySource(), phy.mapping.Mapper("qam", m)
    demap, chan = phy.mapping.Demapper("app", "qam", m), phy.channel.AWGN()
    code = LDPC5GCode(k, n, m, "bg1")
    ebnos = [float(x) for x in np.arange(2.0, 5.01, 0.5)]
    res = {"config": "C2: LDPC5G BG1 k=2816 n=8448 (num_bits_per_symbol=6), 64-QAM, AWGN, flooding BP 20 iterations",
           "batch_size": args.batch, "max_mc_iter": args.max_mc_iter, "ebno_db": ebnos, "rules": {}}
    for cn in ("minsum", "boxplus-phi"):
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=20)
        last = {}

        def mc(batch_size, ebno_db):
            no = phy.utils.ebnodb2no(ebno_db, m, k / n)
            b = src([batch_size, k])
            llr = demap(chan(mapper(enc(b)), no), no)
            b_hat = dec(llr)
            last["llr"], last["b_hat"] = llr, b_hat
            return b, b_hat

        t0 = time.time()
        checks = []
        ber, bler = [], []
        for e in ebnos:                                  # one sim_ber call per point to grab the last batch's LLRs
            r_ber, r_bler = phy.utils.sim_ber(mc, [e], args.batch, args.max_mc_iter, num_target_block_errors=2000,
                                              verbose=False)
            ber.append(float(r_ber[0])); bler.append(float(r_bler[0]))
            if cn == "minsum":
                s = args.oracle_sample
                llr = last["llr"][:s].cpu().numpy()
                odec = obp.LDPC5GDecoder(code, cn_update="minsum", hard_out=True, num_iter=20)
                ref = cbind.bp_decode(odec, odec.rate_recover(llr))[:, :k]
                checks.append(bool(np.array_equal(ref, last["b_hat"][:s].cpu().numpy())))
        torch.cuda.synchronize()
        res["rules"][cn] = {"engine": "on-chip" if (dec._onchip_ok and dec._cn_mode in (2, 3)) else "generic-hbm",
                            "ber": ber, "bler": bler, "seconds": round(time.time() - t0, 1)}
        if cn == "minsum":
            res["rules"][cn]["oracle_bit_exact_on_sample"] = checks
            res["rules"][cn]["oracle_sample_codewords_per_point"] = args.oracle_sample
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
