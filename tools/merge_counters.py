#!/usr/bin/env python3
"""Merge the entries of a partial PMC collection (gpurun_out/pmc_<tag>/counters.json from tools/gpu_pmc_mini.sh) into
profiles/counters.json: the listed kernels are replaced (each entry carries its own `from` and `collected_on`), every other
entry stays as it was collected.   usage: python tools/merge_counters.py gpurun_out/pmc_<tag>/counters.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "counters.json")
new = json.load(open(sys.argv[1]))
cur = json.load(open(dst))
for key, rec in new["kernels"].items():
    if not rec.get("valu_insts_per_unit"):
        print("skip (no counters):", key)
        continue
    rec["collected_on"] = new.get("collected_on")
    rec["tag"] = new.get("tag")
    cur["kernels"][key] = rec
    print("merged", key, "from", rec.get("from"))
json.dump(cur, open(dst, "w"), indent=1)
