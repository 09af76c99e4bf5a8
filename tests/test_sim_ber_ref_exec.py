"""The Monte-Carlo driver against the reference's OWN ``sim_ber`` (src/sionna/phy/utils/misc.py:329-865), executed from
its source file under the NumPy stand-in for TensorFlow on scripted ``mc_fun``s (tools/gen_sim_ber_golden.py ->
tests/golden/sim_ber_ref_golden.json).  ``sionna_amd.phy.utils.sim_ber`` driven with the same scripted bits must return the
same BER / BLER bit for bit, call ``mc_fun`` with the same Eb/N0 sequence (= take every stop decision at the same
iteration), hand the callback the same counters, and print the same table (runtime column aside): max iterations,
target bit / block errors, error-free early stop, target BER / BLER, early_stop off, soft estimates, callback
skip / stop, multi-dimensional bits, double precision."""
import json
import os

import numpy as np
import pytest
import torch

from tools.gen_sim_ber_golden import run

with open(os.path.join(os.path.dirname(__file__), "golden", "sim_ber_ref_golden.json")) as _f:
    GOLD = json.load(_f)["scenarios"]


@pytest.mark.parametrize("case", GOLD, ids=[c["config"]["name"] for c in GOLD])
@pytest.mark.parametrize("as_tensor", [True, False], ids=["torch", "numpy"])
def test_sim_ber_equals_reference_execution(case, as_tensor):
    from sionna_amd.phy.utils import sim_ber
    sc, ref = case["config"], case["result"]
    got = run(sim_ber, sc, to_tensor=(lambda a: torch.from_numpy(a)) if as_tensor else (lambda a: a))
    assert got["calls"] == ref["calls"], "mc_fun was called a different number of times / at different Eb/N0"
    assert got["ber_dtype"] == ref["ber_dtype"]
    assert got["ber_hex"] == ref["ber_hex"] and got["bler_hex"] == ref["bler_hex"], (got["ber"], ref["ber"], got["bler"], ref["bler"])
    assert got["callback"] == ref["callback"]
    assert got["table"] == ref["table"]
