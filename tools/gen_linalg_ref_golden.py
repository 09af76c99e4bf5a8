#!/usr/bin/env python3
"""Generates tests/golden/linalg_ref_golden.npz by EXECUTING the reference's own ``inv_cholesky`` / ``matrix_pinv``
(/root/reference/src/sionna/phy/utils/linalg.py:8-59), ``whiten_channel`` and the complex <-> real-valued representation
helpers (mimo/utils.py:11-356) and ``lmmse_matrix`` (mimo/equalization.py:11-99) under the NumPy stand-in for TensorFlow
(tools/ref_exec), in double precision, on random well-conditioned problems.  Run here (needs /root/reference); the fixture
travels.  tests/test_oracle_ref_exec_linalg.py holds oracle/linalg.py to it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "linalg_ref_golden.npz")

SIZES = [(1, 1), (2, 1), (2, 2), (4, 2), (4, 4), (8, 3), (16, 8), (16, 16), (3, 5)]      # (M, K); K > M only for lmmse_matrix / whiten


def problems(rng, n, m, k):
    cn = lambda *s: (rng.normal(size=s) + 1j * rng.normal(size=s)) * np.sqrt(0.5)
    h = cn(n, m, k)
    e = cn(n, m, m)
    s = e @ np.conj(np.swapaxes(e, -1, -2)) + 0.3 * np.eye(m)
    y = cn(n, m)
    return y, h, s


def main():
    from tools.ref_exec.loader import reference
    ref = reference()
    ref.load_utils()
    tf = ref.tf
    lin = sys.modules.get("sionna.phy.utils.linalg") or ref.load("sionna.phy.utils.linalg")
    mu = ref.load("sionna.phy.mimo.utils")
    mimo = sys.modules["sionna.phy.mimo"]
    for k_, v_ in vars(mu).items():
        if not k_.startswith("_"):
            setattr(mimo, k_, v_)
    eq = ref.load("sionna.phy.mimo.equalization")
    rng = np.random.default_rng(20260925)
    out = {"sizes": np.asarray(SIZES)}
    A = np.asarray
    for i, (m, k) in enumerate(SIZES):
        y, h, s = problems(rng, 6, m, k)
        out[f"y{i}"], out[f"h{i}"], out[f"s{i}"] = y, h, s
        c = lambda a: tf.constant(a, dtype=tf.complex128)
        out[f"inv_chol{i}"] = A(lin.inv_cholesky(c(s)))
        out[f"inv_chol_real{i}"] = A(lin.inv_cholesky(tf.constant(s.real + np.eye(m), dtype=tf.float64)))
        if k <= m:
            out[f"pinv{i}"] = A(lin.matrix_pinv(c(h)))
        yw, hw, sw = mu.whiten_channel(c(y), c(h), c(s))
        out[f"yw{i}"], out[f"hw{i}"] = A(yw), A(hw)
        assert np.allclose(A(sw), np.eye(m))
        out[f"g{i}"] = A(eq.lmmse_matrix(c(h), c(s), precision="double"))
        out[f"g_white{i}"] = A(eq.lmmse_matrix(c(h), None, precision="double"))
        yr, hr, sr = mu.complex2real_channel(c(y), c(h), c(s))
        out[f"yr{i}"], out[f"hr{i}"], out[f"sr{i}"] = A(yr), A(hr), A(sr)
        yc, hc, sc = mu.real2complex_channel(yr, hr, sr)
        assert np.allclose(A(yc), y) and np.allclose(A(hc), h) and np.allclose(A(sc), s)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
