#!/usr/bin/env python3
"""Generates tests/golden/sim_ber_ref_golden.json by EXECUTING the reference's own Monte-Carlo driver
(src/sionna/phy/utils/misc.py:329-865 ``sim_ber`` with utils/metrics.py ``count_errors`` / ``count_block_errors`` and
``hard_decisions``) under the NumPy stand-in for TensorFlow, on SCRIPTED ``mc_fun``s: the bits a call returns are a pure
function of (scenario seed, Eb/N0, number of the call), so the product's ``sim_ber`` can be driven with the very same
sequence (tests/test_sim_ber_ref_exec.py).  Recorded per scenario: the returned BER / BLER (float32 bit patterns), the
Eb/N0 of every ``mc_fun`` call in order, every callback invocation (iteration, point, the four counters), and the table
the run printed (final line of every point + the stop messages; the runtime column blanked).

Scenarios cover every stopping rule and status of the driver: max iterations, target bit / block errors, early stop on an
error-free point, target BER / BLER, early_stop=False, soft estimates, callbacks that skip a point / stop the run,
multi-dimensional bit tensors, a single Eb/N0 point, and double precision.  Run here (needs /root/reference)."""
import contextlib
import io
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "sim_ber_ref_golden.json")


def scripted_bits(seed, ebno_db, call, batch_size, shape_tail, soft):
    """The transmitted and received bits of one call: error probability 0.2 * 10^(-ebno/4) per bit (0 above 9.5 dB),
    concentrated on a few blocks.  Pure NumPy: used by the generator and by the test."""
    rng = np.random.default_rng([int(seed), int(round(float(ebno_db) * 1000)) + 100000, int(call)])
    shape = (int(batch_size),) + tuple(shape_tail)
    b = rng.integers(0, 2, shape).astype(np.float32)
    p = 0.2 * 10 ** (-float(ebno_db) / 4) if float(ebno_db) < 9.5 else 0.0
    bad_block = rng.random(shape[:-1] + (1,)) < 0.3
    flip = (rng.random(shape) < p / 0.3) & bad_block
    b_hat = np.where(flip, 1 - b, b).astype(np.float32)
    if soft:                                                   # logits: positive = bit 1; a few exact zeros (decided 0)
        mag = rng.uniform(0.1, 4.0, shape).astype(np.float32)
        b_hat = ((2 * b_hat - 1) * mag).astype(np.float32)
        zero = rng.random(shape) < 0.01
        b_hat[zero] = 0.0
    return b, b_hat


SCENARIOS = [
    dict(name="max_iterations", ebno=[0, 2, 4], batch=8, tail=[16], max_mc_iter=5),
    dict(name="target_bit_errors", ebno=[0, 1, 2, 3, 4, 5], batch=16, tail=[32], max_mc_iter=20, num_target_bit_errors=40),
    dict(name="target_block_errors", ebno=[0, 2, 4, 6], batch=16, tail=[32], max_mc_iter=20, num_target_block_errors=7),
    dict(name="both_targets", ebno=[1, 3, 5, 7], batch=8, tail=[24], max_mc_iter=30, num_target_bit_errors=25, num_target_block_errors=9),
    dict(name="no_error_early_stop", ebno=[6, 8, 10, 12, 14], batch=8, tail=[16], max_mc_iter=4),
    dict(name="early_stop_off", ebno=[8, 10, 12], batch=8, tail=[16], max_mc_iter=3, early_stop=False, target_ber=1e-1),
    dict(name="target_ber", ebno=[0, 2, 4, 6, 8], batch=16, tail=[32], max_mc_iter=6, target_ber=2e-2),
    dict(name="target_bler", ebno=[0, 2, 4, 6, 8], batch=16, tail=[32], max_mc_iter=6, target_bler=2.5e-1),
    dict(name="soft_estimates", ebno=[0, 3, 6], batch=8, tail=[20], max_mc_iter=4, soft_estimates=True, num_target_block_errors=10),
    dict(name="multi_dim_bits", ebno=[0, 4], batch=4, tail=[2, 3, 12], max_mc_iter=3),
    dict(name="single_point_quiet", ebno=[2.5], batch=8, tail=[16], max_mc_iter=2, verbose=False),
    dict(name="callback_next_snr", ebno=[0, 2, 4], batch=8, tail=[16], max_mc_iter=6, callback="next_after_2"),
    dict(name="callback_stop", ebno=[0, 2, 4], batch=8, tail=[16], max_mc_iter=6, callback="stop_at_point_1"),
    dict(name="callback_continue", ebno=[0, 2], batch=8, tail=[16], max_mc_iter=3, callback="continue"),
    dict(name="double_precision", ebno=[0.123456789, 3.3], batch=8, tail=[16], max_mc_iter=3, precision="double"),
    dict(name="fractional_ebno", ebno=[-1.25, 0.5, 2.75], batch=8, tail=[16], max_mc_iter=3, num_target_bit_errors=1000),
]


def make_callback(kind, log, consts):
    def cb(mc_iter, snr_idx, ebno_dbs, bit_errors, block_errors, nb_bits, nb_blocks):
        log.append([int(mc_iter), int(snr_idx), [int(v) for v in np.asarray(bit_errors)], [int(v) for v in np.asarray(block_errors)],
                    [int(v) for v in np.asarray(nb_bits)], [int(v) for v in np.asarray(nb_blocks)]])
        if kind == "next_after_2" and int(mc_iter) == 1:
            return consts["next"]
        if kind == "stop_at_point_1" and int(snr_idx) == 1 and int(mc_iter) == 2:
            return consts["stop"]
        return consts["cont"]
    return cb


def clean_table(text):
    """What a run printed, made comparable: carriage-return progress lines dropped (only the last state of a line stays),
    the runtime column blanked."""
    out = []
    for line in text.split("\n"):
        line = line.split("\r")[-1].rstrip()
        cells = line.split("|")
        if len(cells) == 9 and "runtime" not in cells[7]:
            cells[7] = " " * len(cells[7])
            line = "|".join(cells).rstrip()
        out.append(line)
    return "\n".join(out).strip("\n")


def run(sim_ber, sc, to_tensor=lambda a: a):
    """Drive a ``sim_ber`` (the reference's or the product's) through one scenario."""
    calls, cb_log = [], []
    count = {}

    def mc_fun(batch_size, ebno_db):
        e = float(np.asarray(ebno_db))
        c = count.get(e, 0)
        count[e] = c + 1
        calls.append(e)
        b, b_hat = scripted_bits(sc.get("seed", 7), e, c, int(np.asarray(batch_size)), sc["tail"], sc.get("soft_estimates", False))
        return to_tensor(b), to_tensor(b_hat)

    kw = {k: sc[k] for k in ("soft_estimates", "num_target_bit_errors", "num_target_block_errors", "target_ber", "target_bler",
                             "early_stop", "verbose", "precision") if k in sc}
    if "callback" in sc:
        kw["callback"] = make_callback(sc["callback"], cb_log, dict(next=sim_ber.CALLBACK_NEXT_SNR, stop=sim_ber.CALLBACK_STOP,
                                                                    cont=sim_ber.CALLBACK_CONTINUE))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ber, bler = sim_ber(mc_fun, np.array(sc["ebno"], np.float64), sc["batch"], sc["max_mc_iter"], **kw)
    ber, bler = np.asarray(ber), np.asarray(bler)
    return dict(ber_dtype=str(ber.dtype), ber_hex=ber.tobytes().hex(), bler_hex=bler.tobytes().hex(),
                ber=[float(v) for v in ber], bler=[float(v) for v in bler], calls=calls, callback=cb_log,
                table=clean_table(buf.getvalue()))


def main():
    from tools.ref_exec.loader import reference
    ref = reference()
    utils = ref.load_utils()
    out = {"_comment": "reference sim_ber (utils/misc.py:329-865) executed under tools/ref_exec on scripted mc_funs; "
                       "tools/gen_sim_ber_golden.py", "scenarios": []}
    for sc in SCENARIOS:
        res = run(utils.sim_ber, sc)
        out["scenarios"].append(dict(config=sc, result=res))
        print(f"== {sc['name']}: calls {len(res['calls'])}, ber {res['ber']}, bler {res['bler']}")
        print(res["table"])
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
