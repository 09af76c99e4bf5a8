"""CPU oracle of the Polar belief-propagation decoder (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``PolarBPDecoder`` of the reference (src/sionna/phy/fec/polar/decoding.py:1440-1771): flooding BP on the
factor graph of the n x n polar transform with ``log2 n`` stages of n / 2 two-by-two butterflies
(``_decode_bp`` :1605-1724), the boxplus of ``_boxplus_tf`` (:1587-1603) evaluated literally,

    boxplus(x, y) = log(1 + exp(x + y)) - log(exp(x) + exp(y)),   x, y clipped to +-19.3 first,

float32 throughout, the schedule of the reference: per iteration one left-to-right sweep (stages 0 .. S-1, producing the
R messages of columns 1 .. S from the frozen-bit priors of column 0 and the L messages of the PREVIOUS iteration, zeros
in the first) and one right-to-left sweep (stages S-1 .. 0, producing the L messages of columns S-1 .. 0 from the
channel values in column S and the R messages of THIS iteration); decisions from column 0 of L after the last iteration.

Two arithmetics for exp / log:
  math="numpy"  NumPy's float32 exp / log - what the reference's source computes when it is EXECUTED under the NumPy
                stand-in for TensorFlow (tools/gen_polar_bp_ref_golden.py); the pin of this restatement
                (tests/test_oracle_ref_exec_polar_bp.py: bit for bit).
  math="spec"   the defined Cephes-style float32 exp / log of oracle/ldpc_bp.c (spec_expf / spec_logf, the arithmetic
                the boxplus-phi rule of the LDPC decoders is specified on; <= 1 ulp from NumPy's).  The HIP kernel
                (csrc/polar_bp.hip) follows it operation for operation, so GPU == oracle is an array_equal.
"""
import numpy as np

from . import cbind

F = np.float32
LLR_MAX = F(19.3)          # decoding.py:1527


def _exp_log(math):
    if math == "f64":                                # precision = "double": float64, NumPy's (libm) exp / log
        return np.exp, np.log
    if math == "numpy":
        return (lambda v: np.exp(v, dtype=F)), (lambda v: np.log(v, dtype=F))
    if math == "spec":
        return cbind.spec_exp_f32, cbind.spec_log_f32
    raise ValueError("math must be 'numpy' or 'spec'")


def boxplus(x, y, math="spec"):
    """_boxplus_tf (decoding.py:1587-1603), element-wise, float32."""
    exp, log = _exp_log(math)
    F = np.float64 if math == "f64" else np.float32
    x = np.clip(np.asarray(x, F), -F(LLR_MAX), F(LLR_MAX))
    y = np.clip(np.asarray(y, F), -F(LLR_MAX), F(LLR_MAX))
    out = log(F(1.) + exp(x + y))
    return out - log(exp(x) + exp(y))


def stage_indices(n, s):
    """Upper / lower node of the n / 2 butterflies of stage s (decoding.py:1646-1648)."""
    r = np.arange(n // 2)
    i1 = r * 2 - np.mod(r, 2 ** s)
    return i1, i1 + 2 ** s


def bp_decode(llr_logits, frozen_pos, n, num_iter=20, hard_out=True, math="spec", return_all=False):
    """PolarBPDecoder.call (decoding.py:1735-1771): logits [..., n] -> [..., k] (hard bits as float32, or soft logits).
    return_all: also the final L column 0 for every position (internal LLR sign), for diagnostics."""
    F = np.float64 if math == "f64" else np.float32
    llr = np.asarray(llr_logits, F)
    lead = llr.shape[:-1]
    ch = (F(-1.) * llr.reshape(-1, n)).astype(F)                       # :1752 logits -> LLRs
    B = ch.shape[0]
    S = int(np.log2(n))
    frozen_pos = np.asarray(frozen_pos).astype(int)
    info_pos = np.setdiff1d(np.arange(n), frozen_pos)
    L = np.zeros((S + 1, B, n), F)                                       # column S = channel
    R = np.zeros((S + 1, B, n), F)                                       # column 0 = priors (:1632-1636)
    L[S] = ch
    R[0][:, frozen_pos] = F(LLR_MAX)
    idx = [stage_indices(n, s) for s in range(S)]
    for _ in range(int(num_iter)):
        for s in range(S):                                               # left to right (:1641-1683)
            i1, i2 = idx[s]
            l1, l2 = L[s + 1][:, i1], L[s + 1][:, i2]                    # previous iteration (zeros in the first)
            r1, r2 = R[s][:, i1], R[s][:, i2]
            R[s + 1][:, i1] = boxplus(r1, l2 + r2, math)
            R[s + 1][:, i2] = boxplus(r1, l1, math) + r2
        for s in range(S - 1, -1, -1):                                   # right to left (:1685-1713)
            i1, i2 = idx[s]
            l1, l2 = L[s + 1][:, i1], L[s + 1][:, i2]
            r1, r2 = R[s][:, i1], R[s][:, i2]
            L[s][:, i1] = boxplus(l1, l2 + r2, math)
            L[s][:, i2] = boxplus(r1, l1, math) + l2
    u = L[0][:, info_pos]
    out = np.where(u > 0, F(0.), F(1.)) if hard_out else (F(-1.) * u)   # :1719-1723
    out = out.astype(F).reshape(lead + (len(info_pos),))
    return (out, L[0].reshape(lead + (n,))) if return_all else out
