#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (``--kernel-trace --stats`` run) into a text table.

usage: python tools/prof_summary.py gpurun_out/prof/<name>_results.db > profiles/<name>_kernel_stats.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]} (durations in microseconds)")
print(f"{'calls':>6} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel")
for name, calls, total, avg, pct in rows:
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print(f"{calls:>6} {total:>14.1f} {avg:>12.1f} {pct:>7.2f}  {short}")
