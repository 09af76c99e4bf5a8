#!/usr/bin/env python3
"""Generates tests/golden/fec_utils_ref_golden.npz by EXECUTING the reference's own ``fec/utils.py`` (j_fun :184, j_fun_inv
:227, llr2mi :116, bin2int / int2bin (_tf) :532-648, int_mod_2 :1236, make_systematic :797, pcm2gm :986, gm2pcm :908,
verify_gm_pcm :1062, load_parity_check_examples :478, GaussianPriorSource :16 - its mean / variance as a function of
``no`` and ``mi``) under the NumPy stand-in for TensorFlow.  Host-side helpers of the LDPC path (EXIT analysis, generic
linear codes, the Gaussian-LLR shortcut of the BICM notebook).  Run here (needs /root/reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "fec_utils_ref_golden.npz")


def main():
    from tools.ref_exec.loader import reference
    ref = reference()
    ref.load_utils()
    ref.load("sionna.phy.fec.ldpc.codes", package_dir=True)
    fu = ref.load("sionna.phy.fec.utils")
    rng = np.random.default_rng(5)
    out = {}
    mu = np.concatenate([[0.0, 1e-12, 1e-3], np.linspace(0.01, 30, 40), [999.0, 2000.0]]).astype(np.float32)
    mi = np.concatenate([[0.0, 1e-12, 1e-4], np.linspace(0.01, 0.99, 40), [0.999999, 1.0]]).astype(np.float32)
    out["mu"], out["j_fun"] = mu, np.asarray(fu.j_fun(mu), np.float64)
    out["mi"], out["j_fun_inv"] = mi, np.asarray(fu.j_fun_inv(mi), np.float64)
    llr = (rng.normal(size=(6, 500)) * 4 - 3).astype(np.float32)
    s = (1 - 2 * rng.integers(0, 2, llr.shape)).astype(np.float32)
    out["llr"], out["s"] = llr, s
    out["llr2mi"] = np.float64(np.asarray(fu.llr2mi(llr)))
    out["llr2mi_s"] = np.float64(np.asarray(fu.llr2mi(llr, s)))
    out["llr2mi_rows"] = np.asarray(fu.llr2mi(llr, s, reduce_dims=False), np.float64)
    bits = rng.integers(0, 2, (7, 11))
    out["bits"] = bits
    out["bin2int"] = np.array([fu.bin2int(list(b)) for b in bits], np.int64)
    out["bin2int_tf"] = np.asarray(fu.bin2int_tf(bits)).astype(np.int64)
    ints = np.array([0, 1, 5, 12, 255, 1023], np.int64)
    out["ints"] = ints
    out["int2bin"] = np.array([fu.int2bin(int(v), 12) for v in ints], np.int64)
    out["int2bin_tf"] = np.asarray(fu.int2bin_tf(ints.astype(np.int32), 12)).astype(np.int64)
    x = np.array([-3.0, -2.0, -1.49, -0.5, 0.0, 0.5, 1.0, 1.51, 2.0, 7.0, 8.49], np.float32)
    out["mod2_in"], out["mod2"] = x, np.asarray(fu.int_mod_2(x)).astype(np.float32)
    for i in range(4):
        pcm, k, n, r = fu.load_parity_check_examples(i)
        out[f"pcm{i}_shape"] = np.array(pcm.shape)
        out[f"pcm{i}_sum"] = np.array([int(pcm.sum()), k, n])
        gm = fu.pcm2gm(pcm)
        out[f"gm{i}"] = np.packbits(np.asarray(gm).astype(np.uint8), axis=1)
        out[f"gm{i}_ok"] = np.int64(bool(fu.verify_gm_pcm(gm, pcm)))
        msys, swaps = fu.make_systematic(pcm, is_pcm=True)
        out[f"sys{i}"] = np.packbits(np.asarray(msys).astype(np.uint8), axis=1)
        out[f"swaps{i}"] = np.array(swaps, np.int64).reshape(-1, 2)
        out[f"pcm_back{i}"] = np.packbits(np.asarray(fu.gm2pcm(gm)).astype(np.uint8), axis=1)
    # GaussianPriorSource: the parameters it derives (mean / standard deviation of its output over 400k draws)
    src = fu.GaussianPriorSource()
    rows = []
    for kind, v in (("no", 0.5), ("no", 2.0), ("mi", 0.3), ("mi", 0.9)):
        y = np.asarray(src([400000], **{kind: np.float32(v)}))
        rows.append([kind == "mi", v, float(y.mean()), float(y.std())])
    out["gps"] = np.array(rows)
    np.savez_compressed(OUT, **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:12]})
    print("gps (is_mi, value, mean, std):", out["gps"])
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
