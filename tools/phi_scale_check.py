"""boxplus-phi (the reference's default check-node rule) at BASELINE config C2 scale on the GPU against the CPU oracle
(oracle/ldpc_bp.c, glibc float32 exp/log): n=8448, k=2816, 64-QAM, 20 iterations, waterfall SNRs, B codewords.
Prints one JSON line per SNR: fraction of identical hard decisions (all info bits / per codeword), fraction of soft
outputs within 1e-5 relative (+1e-4 absolute floor) and within 1e-3, BLER of both.  Run on the GPU box:
    python tools/phi_scale_check.py [B] > gpurun_out/phi_scale.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(B=2048, ebnos=(3.5, 4.0, 4.5), rules=("boxplus-phi", "minsum")):
    import sionna_amd.phy as phy
    from oracle.ldpc5g import LDPC5GCode
    from oracle import ldpc_bp as obp, cbind
    k, n, m = 2816, 8448, 6
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    code = LDPC5GCode(k, n, m, "bg1")
    src, mapper, awgn, demap = phy.mapping.BinarySource(), phy.mapping.Mapper("qam", m), phy.channel.AWGN(), \
        phy.mapping.Demapper("app", "qam", m)
    out = []
    for ebno in ebnos:
        phy.config.seed = int(ebno * 100)
        no = phy.utils.ebnodb2no(ebno, m, k / n)
        u = src([B, k])
        llr = demap(awgn(mapper(enc(u)), no), no)
        un, ln = u.cpu().numpy(), llr.cpu().numpy()
        for cn in rules:
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=20, hard_out=False)
            got = dec(llr).cpu().numpy()
            odec = obp.LDPC5GDecoder(code, cn_update=cn, num_iter=20, hard_out=False)
            t = time.time()
            ref = cbind.bp_decode(odec, odec.rate_recover(ln))[:, :k]
            t = time.time() - t
            hg, hr = got > 0, ref > 0
            rec = {"rule": cn, "ebno_db": ebno, "codewords": B, "engine": dec.engine if hasattr(dec, "engine") else None,
                   "hard_equal_bits": float(np.mean(hg == hr)),
                   "hard_equal_codewords": float(np.mean(np.all(hg == hr, axis=1))),
                   "soft_within_1e-5": float(np.mean(np.isclose(got, ref, rtol=1e-5, atol=1e-4))),
                   "soft_within_1e-3": float(np.mean(np.isclose(got, ref, rtol=1e-3, atol=1e-3))),
                   "bler_gpu": float(np.mean(np.any(hg != (un > 0), axis=1))),
                   "bler_oracle": float(np.mean(np.any(hr != (un > 0), axis=1))),
                   "ber_gpu": float(np.mean(hg != (un > 0))), "ber_oracle": float(np.mean(hr != (un > 0))),
                   "oracle_seconds": round(t, 2), "oracle_threads": cbind.num_threads()}
            conv = np.all(hr == (un > 0), axis=1)
            rec["hard_equal_on_oracle_decoded_words"] = float(np.mean(np.all(hg[conv] == hr[conv], axis=1))) if conv.any() else None
            rec["soft_within_1e-5_on_oracle_decoded_words"] = float(np.mean(np.isclose(got[conv], ref[conv], rtol=1e-5, atol=1e-4))) if conv.any() else None
            out.append(rec)
            print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 2048)
