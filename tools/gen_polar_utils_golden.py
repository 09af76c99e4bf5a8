#!/usr/bin/env python3
"""Generates tests/golden/polar_utils_ref_golden.npz by EXECUTING the reference's own ``generate_polar_transform_mat``,
``generate_dense_polar`` and ``generate_rm_code`` (/root/reference/src/sionna/phy/fec/polar/utils.py:114-290; plain NumPy
functions, loaded from the source file without importing the package).  Matrices are stored bit-packed.  Run here (needs
/root/reference); the fixture travels.  tests/test_oracle_polar.py holds sionna_amd.phy.fec.polar.utils to it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "/root/reference/src/sionna/phy/fec/polar/utils.py"
OUT = os.path.join(ROOT, "tests", "golden", "polar_utils_ref_golden.npz")


def load_reference_functions():
    import numbers
    text = open(SRC).read()
    body = text[text.index("def generate_polar_transform_mat"):]           # (generate_5g_ranking above it reads a package resource)
    ns = {"np": np, "numbers": numbers, "comb": __import__("scipy.special", fromlist=["comb"]).comb}
    try:
        import matplotlib.pyplot as plt
        ns["plt"] = plt
    except Exception:                                                    # noqa: BLE001  (only used with verbose=True)
        pass
    exec(compile(body, SRC, "exec"), ns)
    return ns


def main():
    ns = load_reference_functions()
    from sionna_amd.phy.fec.polar.utils import generate_5g_ranking      # the ranking itself is pinned by tests/golden (round 1)
    out = {}
    for n_lift in range(0, 9):
        g = np.asarray(ns["generate_polar_transform_mat"](n_lift))
        out[f"tm{n_lift}_shape"] = np.asarray(g.shape)
        out[f"tm{n_lift}"] = np.packbits(g.astype(np.uint8).reshape(-1))
    cases = [(32, 16), (64, 6), (128, 64), (256, 230), (512, 51)]
    out["dense_cases"] = np.asarray(cases)
    for i, (n, k) in enumerate(cases):
        frozen, _ = generate_5g_ranking(k, n)
        pcm, gm = ns["generate_dense_polar"](frozen, n, verbose=False)
        out[f"dp{i}_pcm"] = np.packbits(np.asarray(pcm).astype(np.uint8).reshape(-1))
        out[f"dp{i}_gm"] = np.packbits(np.asarray(gm).astype(np.uint8).reshape(-1))
    rm = [(0, 3), (1, 5), (3, 7), (2, 8), (5, 5)]
    out["rm_cases"] = np.asarray(rm)
    for i, (r, m) in enumerate(rm):
        f, inf, n, k, d = ns["generate_rm_code"](r, m)
        out[f"rm{i}_frozen"], out[f"rm{i}_info"], out[f"rm{i}_nkd"] = np.asarray(f), np.asarray(inf), np.asarray([n, k, d])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
