"""Pins the Polar BP oracle (oracle/polar_bp.py) to the reference's OWN ``PolarBPDecoder`` /
``Polar5GDecoder(dec_type="BP")`` executed here: tests/golden/polar_bp_ref_golden.npz comes from
tools/gen_polar_bp_ref_golden.py, which runs fec/polar/decoding.py:1440-1771, 1896-1912 from the reference's source file
under the NumPy stand-in for TensorFlow.  With NumPy's float32 exp / log (math="numpy": the arithmetic of that execution)
the restatement must give the same soft outputs BIT FOR BIT; with the defined exp / log the HIP kernel follows
(math="spec", <= 1 ulp from NumPy's per call) the soft outputs stay within 1e-4 of scale and the decisions differ only
where a soft output sits at the decision threshold."""
import os

import numpy as np
import pytest

from oracle import polar as op, polar_bp as obp

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "polar_bp_ref_golden.npz"))
PLAIN = [tuple(int(v) for v in r) for r in GOLD["plain"]]
FIVEG = [tuple(int(v) for v in r) for r in GOLD["fiveg"]]
unpack = lambda a, w: np.unpackbits(a, axis=1)[:, :w].astype(np.float32)


def grp(prefix):
    return {key.split("/", 1)[1]: GOLD[key] for key in GOLD.files if key.startswith(prefix + "/")}


@pytest.mark.parametrize("i", range(len(PLAIN)))
def test_bp_decoder_matches_reference_execution(i):
    k, n, it, B = PLAIN[i]
    g = grp(f"p{i}")
    soft = obp.bp_decode(g["logits"], g["frozen_pos"], n, it, hard_out=False, math="numpy")
    assert soft.dtype == np.float32 and np.array_equal(soft, g["soft"])
    hard = obp.bp_decode(g["logits"], g["frozen_pos"], n, it, hard_out=True, math="numpy")
    assert np.array_equal(hard, unpack(g["hard"], k))
    # the defined arithmetic: same decoder within rounding noise of the transcendental calls
    spec = obp.bp_decode(g["logits"], g["frozen_pos"], n, it, hard_out=False, math="spec")
    assert np.max(np.abs(spec - g["soft"])) <= 1e-4 * 19.3
    flips = (spec > 0) != (g["soft"] > 0)
    assert np.all(np.abs(g["soft"][flips]) < 1e-3)


@pytest.mark.parametrize("i", range(len(FIVEG)))
def test_polar5g_bp_chain_matches_reference_execution(i):
    k, n, down, it, B = FIVEG[i]
    g = grp(f"g{i}")
    code = op.Polar5GCode(k, n, "downlink" if down else "uplink")
    uh = op.polar5g_decode(code, g["logits"], "BP", num_iter=it, bp_math="numpy")
    assert np.array_equal(uh, unpack(g["u_hat"], k))
    # CRC status of Polar5GDecoder(return_crc_status=True) (decoding.py:2063-2067): a CRC check of the decisions
    u_crc = op.polar5g_decode(code, g["logits"], "BP", num_iter=it, bp_math="numpy", keep_crc=True)
    assert np.array_equal(op.crc_check(u_crc, code.crc_degree)[1].reshape(-1).astype(np.uint8), g["crc"].reshape(-1))


def test_boxplus_literal_form():
    """_boxplus_tf (decoding.py:1587-1603) against float64: clip at +-19.3, symmetric, boxplus(x, 0) = 0."""
    rng = np.random.default_rng(0)
    x, y = (rng.normal(size=4000) * 12).astype(np.float32), (rng.normal(size=4000) * 12).astype(np.float32)
    xc, yc = np.clip(x.astype(np.float64), -19.3, 19.3), np.clip(y.astype(np.float64), -19.3, 19.3)
    want = np.log1p(np.exp(xc + yc)) - np.logaddexp(xc, yc)
    for math in ("numpy", "spec"):
        got = obp.boxplus(x, y, math)
        assert np.max(np.abs(got - want)) < 2e-5, math       # float32 cancellation of two logs of magnitude <= 38.6
        assert np.array_equal(got, obp.boxplus(y, x, math))
        assert np.max(np.abs(obp.boxplus(x, np.zeros_like(x), math))) < 1e-6
