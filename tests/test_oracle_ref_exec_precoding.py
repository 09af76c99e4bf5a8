"""Pins the RZF precoder oracle (oracle/precoding.py) to the reference's OWN ``rzf_precoder`` / ``RZFPrecoder`` executed under
the NumPy stand-in (tests/golden/precoding_ref_golden.npz, tools/gen_precoding_ref_golden.py).  There is no HIP kernel for
this row yet: oracle + fixture are the target it will be held to (the downlink tables of the CDL notebook need it)."""
import os

import numpy as np
import pytest

from oracle import precoding as op

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "precoding_ref_golden.npz"))


def close(a, b, tol=2e-5):
    return a.shape == b.shape and np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1.0)


@pytest.mark.parametrize("i", range(5))
def test_rzf_precoder_matches_reference_execution(i):
    g = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith(f"m{i}/")}
    xp, gm = op.rzf_precoder(g["x"], g["h"], g["alpha"])
    assert close(gm, g["g"]) and close(xp, g["x_precoded"])
    assert np.allclose(np.sum(np.abs(gm) ** 2, axis=-2), 1.0)                       # unit-norm precoding vectors
    if float(np.max(g["alpha"])) == 0.0:                                             # zero forcing: H G is diagonal
        hg = g["h"].astype(np.complex128) @ gm
        off = hg - np.einsum("...kk->...k", hg)[..., None] * np.eye(hg.shape[-1])
        assert np.abs(off).max() < 1e-9


@pytest.mark.parametrize("tag,alpha", [("zf", 0.0), ("rzf", 0.2)])
def test_ofdm_rzf_precoder_matches_reference_execution(tag, alpha):
    xp, heff = op.ofdm_rzf_precoder(GOLD["o/x_rg"], GOLD["o/h"], GOLD["o/precoding_ind"], GOLD["o/effective_subcarrier_ind"], alpha)
    assert close(xp, GOLD[f"o_{tag}/x_precoded"], 1e-4) and close(heff, GOLD[f"o_{tag}/h_eff"], 1e-4)
