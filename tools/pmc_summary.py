#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        rd = csv.DictReader(fh)
        per_dispatch = defaultdict(dict)
        for row in rd:
            k = row.get("Kernel_Name", "").replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            per_dispatch[(row.get("Dispatch_Id"), k)][row["Counter_Name"]] = float(row["Counter_Value"])
        for (_, k), d in per_dispatch.items():
            for c, v in d.items():
                acc[k][c].append(v)
print(f"# mean PMC value per dispatch, from {root}")
for k in sorted(acc, key=lambda k: -sum(len(v) for v in acc[k].values())):
    if not any(s in k for s in ("ldpc", "cn_pass", "vn_pass", "demap", "polar_scl", "lmmse", "tdl", "cir_to")):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
