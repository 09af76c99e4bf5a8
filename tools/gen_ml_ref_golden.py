#!/usr/bin/env python3
"""Generates tests/golden/ml_ref_golden.npz by EXECUTING the reference's own ``MaximumLikelihoodDetector``
(/root/reference/src/sionna/phy/mimo/detection.py:145-537, with ``whiten_channel`` of mimo/utils.py and ``SymbolLogits2LLRs`` /
``LLRs2SymbolLogits`` of mapping.py) under the NumPy stand-in for TensorFlow (tools/ref_exec): random channels, noise
covariances and priors; bit and symbol outputs, "app" and "maxlog", soft and hard, with and without prior - and the reference's
``KBestDetector(use_real_rep=True)`` (:539-1037 with complex2real_channel, List2LLRSimple, PAM2QAM) on inputs of the same kind.  Run here (needs
/root/reference); the fixture travels.  tests/test_oracle_ref_exec_ml.py holds oracle/ofdm.py::ml_detector to it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "ml_ref_golden.npz")


class _PlainNp:
    """NumPy for the reference's detection module with reductions applied to plain arrays (the stand-in's tensor class does
    not support the dtype view inside np.std)"""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def std(a, *args, **kw):
        return np.std(np.asarray(a), *args, **kw)


def load():
    from tools.ref_exec import tf_numpy
    from tools.ref_exec.loader import reference
    ref = reference()
    ref.load_utils()
    tf = ref.tf
    tf.linalg.matrix_transpose = lambda a, **k: tf_numpy._t(np.swapaxes(np.asarray(a), -1, -2))
    mp = ref.load("sionna.phy.mapping")
    mimo = sys.modules["sionna.phy.mimo"]
    for n in ("utils", "equalization", "detection"):
        m = ref.load(f"sionna.phy.mimo.{n}")
        if n == "detection":
            m.np = _PlainNp()
        for k, v in vars(m).items():
            if not k.startswith("_"):
                setattr(mimo, k, v)
    return tf, mp, mimo


CASES = [  # (num_rx_ant, num_streams, num_bits_per_symbol, output, method, hard_out, with_prior)
    (4, 2, 2, "bit", "app", False, False), (4, 2, 2, "bit", "maxlog", False, True), (4, 2, 2, "symbol", "app", False, True),
    (4, 2, 4, "bit", "app", False, True), (4, 2, 4, "bit", "maxlog", True, False), (2, 2, 4, "symbol", "maxlog", True, False),
    (4, 4, 2, "bit", "app", False, False), (8, 2, 2, "symbol", "app", False, False), (2, 1, 6, "bit", "app", False, True),
]


KBEST_REAL = [  # (num_rx_ant, num_streams, num_bits_per_symbol, k, output, hard_out)
    (4, 2, 2, 8, "bit", False), (4, 2, 4, 16, "bit", False), (4, 2, 4, 8, "bit", True), (2, 2, 4, 12, "symbol", True),
    (4, 4, 2, 16, "bit", False), (8, 2, 6, 32, "bit", False), (2, 1, 4, 4, "bit", False),
]


def main():
    tf, mp, mimo = load()
    rng = np.random.default_rng(2025)
    out = {"cases": np.array(repr(CASES))}
    for ci, (M, K, nb, output, method, hard, with_prior) in enumerate(CASES):
        n = 24
        h = ((rng.normal(size=(n, M, K)) + 1j * rng.normal(size=(n, M, K))) / np.sqrt(2)).astype(np.complex64)
        pts = np.asarray(mp.Constellation("qam", nb).points)
        x = pts[rng.integers(0, 1 << nb, (n, K))]
        a = ((rng.normal(size=(n, M, M)) + 1j * rng.normal(size=(n, M, M))) / np.sqrt(2)).astype(np.complex64)
        s = (0.05 * (a @ np.conj(np.swapaxes(a, -1, -2)) / M + np.eye(M))).astype(np.complex64)
        w = np.linalg.cholesky(s.astype(np.complex128)) @ ((rng.normal(size=(n, M, 1)) + 1j * rng.normal(size=(n, M, 1))) / np.sqrt(2))
        y = (np.einsum("nmk,nk->nm", h, x) + w[..., 0]).astype(np.complex64)
        prior = None
        if with_prior:
            prior = (rng.normal(size=(n, K, nb if output == "bit" else 1 << nb)) * 2).astype(np.float32)
        det = mimo.MaximumLikelihoodDetector(output, method, K, "qam", nb, hard_out=hard)
        res = np.asarray(det(y, h, s, prior) if with_prior else det(y, h, s))
        out.update({f"c{ci}_y": y, f"c{ci}_h": h, f"c{ci}_s": s, f"c{ci}_out": res, f"c{ci}_points": pts.astype(np.complex64)})
        if with_prior:
            out[f"c{ci}_prior"] = prior
        print(ci, (M, K, nb, output, method, hard, with_prior), res.shape, res.dtype, float(np.abs(res).max()))
    # KBestDetector(use_real_rep=True) (mimo/detection.py:539-1037) on the same kind of inputs
    kb = []
    for ci, (M, K, nb, kk, output, hard) in enumerate(KBEST_REAL):
        n = 24
        h = ((rng.normal(size=(n, M, K)) + 1j * rng.normal(size=(n, M, K))) / np.sqrt(2)).astype(np.complex64)
        pts = np.asarray(mp.Constellation("qam", nb).points)
        x = pts[rng.integers(0, 1 << nb, (n, K))]
        a = ((rng.normal(size=(n, M, M)) + 1j * rng.normal(size=(n, M, M))) / np.sqrt(2)).astype(np.complex64)
        s = (0.05 * (a @ np.conj(np.swapaxes(a, -1, -2)) / M + np.eye(M))).astype(np.complex64)
        w = np.linalg.cholesky(s.astype(np.complex128)) @ ((rng.normal(size=(n, M, 1)) + 1j * rng.normal(size=(n, M, 1))) / np.sqrt(2))
        y = (np.einsum("nmk,nk->nm", h, x) + w[..., 0]).astype(np.complex64)
        det = mimo.KBestDetector(output, K, kk, "qam", nb, hard_out=hard, use_real_rep=True)
        res = np.asarray(det(y, h, s))
        out.update({f"k{ci}_y": y, f"k{ci}_h": h, f"k{ci}_s": s, f"k{ci}_out": res})
        kb.append((M, K, nb, kk, output, hard))
        print("kbest real", ci, kb[-1], res.shape, res.dtype, float(np.abs(res).max()))
    out["kbest_real_cases"] = np.array(repr(kb))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
