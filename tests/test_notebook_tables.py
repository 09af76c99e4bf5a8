"""CPU checks of the reference-table fixture and of the curve-comparison statistics (tests/notebook_curves.py)."""
import math

import numpy as np

import notebook_curves as nc


def test_fixture_has_the_cited_tables():
    t = nc.load_tables()
    assert len(t) >= 120
    r = t["5G_Channel_Coding_Polar_vs_LDPC_Codes/c12/t0"]
    assert r["label"] == "5G LDPC BP-20" and r["ipynb_line"] == 420
    # 5G_Channel_Coding_Polar_vs_LDPC_Codes.ipynb:420-429
    assert (r["rows"][0]["block_errors"], r["rows"][0]["num_blocks"]) == (8643, 10000)
    assert (r["rows"][8]["bit_errors"], r["rows"][8]["num_bits"]) == (9871, 48640000)
    for row in r["rows"]:
        assert math.isclose(row["bler"], row["block_errors"] / row["num_blocks"], rel_tol=2e-4)
        assert math.isclose(row["ber"], row["bit_errors"] / row["num_bits"], rel_tol=2e-4)


def test_every_curve_has_a_table_on_its_grid():
    t = nc.load_tables()
    for c in nc.CURVES:
        rows = t[c.key]["rows"]
        assert 2 <= len(rows) <= len(c.ebno), c.key
        for r, x in zip(rows, c.ebno):
            assert abs(r["ebno_db"] - x) < 0.06, (c.key, r["ebno_db"], x)       # printed column is rounded to 1-3 decimals


def _synthetic(ebno, shift_db, n, rng):
    p = np.clip(10 ** (-(np.asarray(ebno) - shift_db) * 1.2), 0, 1) * 0.9
    e = rng.binomial(n, p)
    return [{"block_errors": int(a), "num_blocks": int(n), "bit_errors": int(a) * 7, "num_bits": int(n) * 100} for a in e]


def test_statistics_accept_equal_curves_and_reject_a_shift():
    rng = np.random.default_rng(7)
    ebno = np.arange(0, 3.01, 0.25)
    ref = _synthetic(ebno, 0.0, 200_000, rng)
    same = nc.compare(ref, _synthetic(ebno, 0.0, 800_000, rng), ebno)
    assert same["ok"], same
    assert abs(same["crossings"]["1e-02"]["delta_db"]) < 0.03
    moved = nc.compare(ref, _synthetic(ebno, 0.1, 800_000, rng), ebno)         # a 0.1 dB offset must be caught
    assert not moved["ok"]
    assert 0.05 < moved["crossings"]["1e-02"]["delta_db"] < 0.15


def test_crossing_interpolates_in_the_log_domain():
    x, s = nc.crossing([0, 1], [1000, 10], [1000, 1000], 0.1)
    assert abs(x - 0.5) < 1e-9 and s > 0
    assert nc.crossing([0, 1], [1000, 900], [1000, 1000], 0.1) == (None, None)
    assert nc.z_score(5, 100, 5, 100) is None
    assert abs(nc.z_score(100, 1000, 400, 4000)) < 1e-12
