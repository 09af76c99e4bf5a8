"""oracle/ofdm.py::ml_detector against the reference's own MaximumLikelihoodDetector EXECUTED under the NumPy stand-in for
TensorFlow (tools/gen_ml_ref_golden.py -> tests/golden/ml_ref_golden.npz): bit / symbol outputs, "app" / "maxlog", soft / hard,
with and without prior.  The reference computes in float32 (complex64 whitening, float32 exponents of magnitude up to ~7e2);
the oracle in float64: soft values within 2e-3 absolute + 2e-4 relative, hard decisions equal wherever the float64 margin
exceeds that error."""
import ast
import os

import numpy as np
import pytest

from oracle import ofdm as o

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ml_ref_golden.npz"))
CASES = ast.literal_eval(str(G["cases"]))


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_ml_oracle_matches_reference_execution(ci):
    M, K, nb, output, method, hard, with_prior = CASES[ci]
    y, h, s, pts, ref = G[f"c{ci}_y"], G[f"c{ci}_h"], G[f"c{ci}_s"], G[f"c{ci}_points"], G[f"c{ci}_out"]
    prior = G[f"c{ci}_prior"] if with_prior else None
    got = o.ml_detector(y, h, s, pts, method, prior, output, hard)
    assert got.shape == ref.shape
    if not hard:
        assert np.allclose(got, ref, rtol=2e-4, atol=2e-3), float(np.max(np.abs(got - ref)))
        return
    soft = o.ml_detector(y, h, s, pts, method, prior, output, False)
    if output == "bit":
        sure = np.abs(soft) > 1e-2
        assert np.array_equal(got[sure], ref[sure]) and sure.mean() > 0.95
    else:
        top2 = np.sort(soft, -1)[..., -2:]
        sure = (top2[..., 1] - top2[..., 0]) > 1e-2
        assert np.array_equal(got[sure], ref[sure]) and sure.mean() > 0.95


KB = ast.literal_eval(str(G["kbest_real_cases"]))


@pytest.mark.parametrize("ci", range(len(KB)))
def test_kbest_real_rep_oracle_matches_reference_execution(ci):
    """oracle/ofdm.py::kbest_detector_real against the reference's KBestDetector(use_real_rep=True) executed: LLRs (clipped to
    +-20) within 5e-3 where the candidate lists agree - the float32 reference and the float64 oracle may keep a different k-th
    path at a near-tie, which moves single LLRs: at least 97 % of the entries must agree."""
    M, K, nb, kk, output, hard = KB[ci]
    y, h, s, ref = G[f"k{ci}_y"], G[f"k{ci}_h"], G[f"k{ci}_s"], G[f"k{ci}_out"]
    got = o.kbest_detector_real(y, h, s, nb, kk, hard, output=output)
    assert got.shape == ref.shape
    if hard:
        assert np.mean(got == ref) > 0.97
    else:
        assert np.mean(np.isclose(got, ref, rtol=1e-3, atol=5e-3)) > 0.97, float(np.mean(np.isclose(got, ref, rtol=1e-3, atol=5e-3)))
