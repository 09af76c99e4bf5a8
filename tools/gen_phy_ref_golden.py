#!/usr/bin/env python3
"""Generates tests/golden/phy_ref_golden.npz by EXECUTING the reference's own mapping / equalisation / utility code under
the NumPy stand-in for TensorFlow (tools/ref_exec): ``mapping.py`` (Constellation/qam/pam :15-193, Mapper :422-519,
Demapper :521-691, SymbolLogits2LLRs :794-967), ``mimo/equalization.py`` (lmmse_matrix :11-99, lmmse_equalizer :101-233,
zf_equalizer :235-298, mf_equalizer :300-368), ``mimo/utils.py`` (whiten_channel :292-356), ``utils/linalg.py``
(inv_cholesky :8-32, matrix_pinv :34-58), ``utils/misc.py`` (ebnodb2no :171-252, hard_decisions :254-272).

Exact pieces (constellation points, mapper output, hard decisions, ebnodb2no) are compared bit for bit; the float
pipelines (demapper LLRs, equaliser outputs) at the north star's 1e-5 relative bar - NumPy's exp/log/BLAS stand for
Eigen's there, see tools/ref_exec/tf_numpy.py.  Run here (needs /root/reference); the fixture travels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "phy_ref_golden.npz")


def cn(rng, shape, var=1.0):
    return ((rng.normal(size=shape) + 1j * rng.normal(size=shape)) * np.sqrt(var / 2)).astype(np.complex64)


def main():
    from tools.ref_exec.loader import reference
    ref = reference()
    utils = ref.load_utils()
    mp = ref.load("sionna.phy.mapping")
    ref.load("sionna.phy.mimo.utils")
    eq = ref.load("sionna.phy.mimo.equalization")
    rng = np.random.default_rng(424242)
    out = {}

    # ---- constellations, mapper, demapper
    for m in (2, 4, 6, 8):
        out[f"qam{m}_points"] = np.asarray(mp.Constellation("qam", m).points)
        bits = rng.integers(0, 2, (4, 6 * m)).astype(np.float32)
        x = np.asarray(mp.Mapper("qam", m)(bits))
        out[f"qam{m}_bits"], out[f"qam{m}_x"] = bits.astype(np.uint8), x
        for no in (0.5, 0.05):
            y = (x + cn(rng, x.shape, no)).astype(np.complex64)
            out[f"qam{m}_y_no{no}"] = y
            for meth in ("app", "maxlog"):
                out[f"qam{m}_{meth}_no{no}"] = np.asarray(mp.Demapper(meth, "qam", m)(y, np.float32(no)))
            out[f"qam{m}_hard_no{no}"] = np.asarray(mp.Demapper("app", "qam", m, hard_out=True)(y, np.float32(no))).astype(np.uint8)
        # per-symbol noise variance and prior LLRs
        no_t = (0.02 + rng.random(x.shape)).astype(np.float32)
        prior = (rng.normal(size=x.shape + (m,)) * 2).astype(np.float32)
        y = (x + cn(rng, x.shape) * np.sqrt(no_t)).astype(np.complex64)
        out[f"qam{m}_y_t"], out[f"qam{m}_no_t"], out[f"qam{m}_prior"] = y, no_t, prior
        for meth in ("app", "maxlog"):
            out[f"qam{m}_{meth}_prior"] = np.asarray(mp.Demapper(meth, "qam", m)(y, no_t, prior))
    for m in (1, 2, 3):
        out[f"pam{m}_points"] = np.asarray(mp.Constellation("pam", m).points)
    pts = cn(rng, (8,))
    cst = mp.Constellation("custom", 3, points=pts, normalize=True, center=True)
    out["custom3_in"], out["custom3_points"] = pts, np.asarray(cst())
    y = cn(rng, (5, 7))
    out["custom3_y"] = y
    out["custom3_app"] = np.asarray(mp.Demapper("app", constellation=cst)(y, np.float32(0.3)))

    # ---- MIMO equalisers: (num_rx, num_streams) of the hot path and the notebooks, white and coloured noise
    for (M, K) in ((4, 2), (2, 1), (1, 1), (8, 4), (16, 4), (4, 4)):
        B = 24
        h = cn(rng, (B, M, K))
        x = cn(rng, (B, K))
        a = cn(rng, (B, M, M), 0.3)
        s_col = (a @ np.conj(np.swapaxes(a, -1, -2)) + (0.05 + rng.random((B, 1, 1))) * np.eye(M)).astype(np.complex64)
        s_wht = ((0.01 + rng.random((B, 1, 1))) * np.eye(M)).astype(np.complex64)
        for tag, s in (("col", s_col), ("wht", s_wht)):
            n = np.linalg.cholesky(s.astype(np.complex128)) @ cn(rng, (B, M, 1)).astype(np.complex128)
            y = (np.einsum("bmk,bk->bm", h, x) + n[..., 0]).astype(np.complex64)
            p = f"mimo{M}x{K}_{tag}_"
            out[p + "y"], out[p + "h"], out[p + "s"] = y, h, s
            for wi in (True, False):
                xh, ne = eq.lmmse_equalizer(y, h, s, whiten_interference=wi)
                out[p + f"lmmse_w{int(wi)}_x"], out[p + f"lmmse_w{int(wi)}_no"] = np.asarray(xh), np.asarray(ne)
            xh, ne = eq.zf_equalizer(y, h, s)
            out[p + "zf_x"], out[p + "zf_no"] = np.asarray(xh), np.asarray(ne)
            xh, ne = eq.mf_equalizer(y, h, s)
            out[p + "mf_x"], out[p + "mf_no"] = np.asarray(xh), np.asarray(ne)
            if (M, K) == (4, 2):
                out[p + "inv_chol"] = np.asarray(utils.inv_cholesky(s))
                out[p + "pinv"] = np.asarray(utils.matrix_pinv(h))

    # ---- utilities
    grid = [(e, m, r) for e in (-3.0, 0.0, 2.5, 10.0) for m in (1, 2, 4, 6) for r in (1.0, 0.5, 1 / 3)]
    out["ebno_grid"] = np.array(grid, np.float64)
    out["ebno_no"] = np.array([np.asarray(utils.ebnodb2no(e, m, r)) for e, m, r in grid], np.float32)
    llr = np.array([-2.0, -0.0, 0.0, 1e-30, 3.0, -1e-30], np.float32)
    out["hard_in"], out["hard_out"] = llr, np.asarray(utils.hard_decisions(llr))

    # SymbolLogits2LLRs as a block of its own (mapping.py:794-967): logits on the points, with / without bit priors
    for m in (1, 2, 4, 6):
        z = (3.0 * rng.normal(size=(3, 7, 1 << m))).astype(np.float32)
        pr_row = (2.0 * rng.normal(size=(3, 7, m))).astype(np.float32)
        pr_vec = (2.0 * rng.normal(size=(m,))).astype(np.float32)
        out[f"l2l{m}_z"], out[f"l2l{m}_prior"], out[f"l2l{m}_prior_vec"] = z, pr_row, pr_vec
        for meth in ("app", "maxlog"):
            blk = mp.SymbolLogits2LLRs(meth, m)
            out[f"l2l{m}_{meth}"] = np.asarray(blk(z))
            out[f"l2l{m}_{meth}_prior"] = np.asarray(blk(z, pr_row))
            out[f"l2l{m}_{meth}_prior_vec"] = np.asarray(blk(z, pr_vec))
        out[f"l2l{m}_hard"] = np.asarray(mp.SymbolLogits2LLRs("app", m, hard_out=True)(z, pr_row)).astype(np.uint8)

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
