"""Pins the 5G Polar rows (config C5 and the rate-matching regimes) to the reference's OWN code executed here:
tests/golden/polar5g_ref_golden.npz comes from tools/gen_polar5g_ref_golden.py, which runs CRCEncoder / CRCDecoder,
Polar5GEncoder, PolarSCDecoder, PolarSCLDecoder - the TensorFlow list decoder (polar/decoding.py:919-1045) AND its NumPy
twin (:1047-1290) - and Polar5GDecoder (SC, SCL-8, SCL-4, hybrid SCL-8, CRC status) from the reference's source files
under the NumPy stand-in for TensorFlow, at noise levels where SC fails on blocks the list recovers.  The oracle must give
the same codewords, decisions and CRC status bit for bit."""
import os

import numpy as np
import pytest

from oracle import polar as op, polar_c as opc

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "polar5g_ref_golden.npz"))
CASES = [tuple(int(v) for v in row) for row in GOLD["cases"]]


def case(i):
    k, n, down, B = CASES[i]
    g = {key.split("/", 1)[1]: GOLD[key] for key in GOLD.files if key.startswith(f"{i}/")}
    unpack = lambda a, w: np.unpackbits(a, axis=1)[:, :w].astype(np.float32)
    return k, n, ("downlink" if down else "uplink"), g, unpack


@pytest.mark.parametrize("i", range(len(CASES)))
def test_polar5g_chain_matches_reference_execution(i):
    k, n, ct, g, unpack = case(i)
    code = op.Polar5GCode(k, n, ct)
    assert code.n_polar == int(g["n_polar"]) and np.array_equal(code.frozen_pos, g["frozen_pos"])
    u = unpack(g["u"], k)
    assert np.array_equal(code.encode(u), unpack(g["c"], n))
    # the reference's TensorFlow list decoder and its NumPy twin agree on these inputs (so one expectation serves both)
    assert np.array_equal(g["u_hat_scl8_tf"], g["u_hat_scl8_np"]) and np.array_equal(g["crc_scl8_tf"], g["crc_scl8_np"])
    logits = g["logits"]
    sc = opc.polar5g_decode(code, logits, dec_type="SC")
    assert np.array_equal(sc, unpack(g["u_hat_sc"], k))
    for name, L in (("scl8_tf", 8), ("scl4_tf", 4)):
        if f"u_hat_{name}" not in g:
            continue
        uh, st = opc.polar5g_decode(code, logits, list_size=L, return_crc_status=True)
        assert np.array_equal(uh, unpack(g[f"u_hat_{name}"], k)), name
        assert np.array_equal(st.astype(np.uint8), g[f"crc_{name}"].reshape(-1)), name
    # hybrid (decoding.py:1292-1340): SC first, the list decoder only for the blocks whose CRC fails
    uh8, st8 = opc.polar5g_decode(code, logits, list_size=8, return_crc_status=True)
    _, sc_ok = opc.polar5g_decode(code, logits, dec_type="SC", return_crc_status=True)
    hyb = np.where(sc_ok[:, None], sc, uh8)
    assert np.array_equal(hyb, unpack(g["u_hat_hyb8"], k))
    assert np.array_equal(np.where(sc_ok, True, st8).astype(np.uint8), g["crc_hyb8"].reshape(-1))
    # the list recovers blocks that SC loses (the cases are chosen for it)
    assert np.sum((uh8 != u).any(-1)) <= np.sum((sc != u).any(-1))


@pytest.mark.parametrize("i", [1, 2, 3, 5])
def test_python_oracle_agrees_on_the_small_cases(i):
    k, n, ct, g, unpack = case(i)
    code = op.Polar5GCode(k, n, ct)
    assert np.array_equal(op.polar5g_decode(code, g["logits"][:4], "SC"), unpack(g["u_hat_sc"], k)[:4])
    assert np.array_equal(op.polar5g_decode(code, g["logits"][:4], "SCL", 8), unpack(g["u_hat_scl8_tf"], k)[:4])
