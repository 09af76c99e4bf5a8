#!/usr/bin/env python3
"""Run the curves of tests/notebook_curves.py on the GPU and write the overlay against the reference's published
tables (tests/golden/notebook_ber.json) to a JSON file under profiles/ (copy from gpurun_out/).

    python tools/ber_vs_reference.py --mult 4 --out gpurun_out/r04_ber_vs_reference.json [--groups awgn,ofdm] [--keys ...]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _js(o):
    import numpy as np
    if isinstance(o, np.bool_):
        return bool(o)
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.floating):
        return float(o)
    raise TypeError(type(o).__name__)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mult", type=float, default=4.0)
    ap.add_argument("--out", default="gpurun_out/ber_vs_reference.json")
    ap.add_argument("--groups", default="")
    ap.add_argument("--keys", default="")
    ap.add_argument("--max-work", type=float, default=2.5e11)
    ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    import notebook_curves as nc
    tabs = nc.load_tables()
    groups = set(filter(None, a.groups.split(",")))
    keys = set(filter(None, a.keys.split(",")))
    doc = {"mult": a.mult, "seed": a.seed, "criteria": {"z_point": nc.Z_POINT, "db_tol": nc.DB_TOL, "chi2_p_min": nc.CHI2_P_MIN},
           "curves": {}}
    n_ok = n_all = 0
    for c in nc.CURVES:
        if groups and c.group not in groups:
            continue
        if keys and c.key not in keys:
            continue
        ref = tabs[c.key]["rows"]
        print(f"== {c.key}: {c.name}", flush=True)
        t0 = time.perf_counter()
        try:
            ours = nc.run_curve(c, ref, mult=a.mult, max_work=a.max_work, seed=a.seed, verbose=True)
        except Exception as e:  # pylint: disable=broad-except
            print(f"   FAILED TO RUN: {type(e).__name__}: {e}", flush=True)
            doc["curves"][c.key] = {"name": c.name, "error": f"{type(e).__name__}: {e}"}
            n_all += 1
            continue
        res = nc.evaluate(c, ref, ours)
        res.update(name=c.name, cite=c.cite, seconds=time.perf_counter() - t0, ours=ours,
                   reference=[{k: r[k] for k in ("ber", "bler", "bit_errors", "num_bits", "block_errors", "num_blocks")} for r in ref])
        doc["curves"][c.key if c.key not in doc["curves"] else c.key + "#" + c.group] = res
        n_all += 1
        n_ok += bool(res["ok"])
        cr = "  ".join(f"{k}: {v['delta_db']:+.3f} dB (tol {v['tol_db']:.3f})" for k, v in res["crossings"].items())
        bcr = "  ".join(f"{k}: {v['delta_db']:+.3f}" for k, v in res.get("ber_crossings", {}).items())
        print(f"   {'OK ' if res['ok'] else 'MISS'} max|z| {res['max_abs_z']:.2f} over {res['n_z']} pts, >3s: {res['n_beyond_3sigma']}, "
              f"chi2 p {res['chi2_p']:.3g};  BLER crossings {cr};  BER crossings {bcr}  [{res['seconds']:.1f} s]", flush=True)
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1, default=_js)
    doc["summary"] = {"curves": n_all, "ok": n_ok}
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1, default=_js)
    print(f"{n_ok}/{n_all} curves agree -> {a.out}")


if __name__ == "__main__":
    main()
