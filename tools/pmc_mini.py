#!/usr/bin/env python3
"""The two secondary workloads whose kernels need fresh PMC counters, in ONE short process (the rocprofv3 target of
tools/gpu_pmc_mini.sh): the C4 chain (ofdm_lmmse_diag_kernel / ofdm_lsnn_lmmse_kernel) and the Polar BP decoder
(polar_bp_kernel).  Prints the two bench sub-lines."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = argparse.Namespace(steps=4, warmup=1, batch=65536, ebno_db=4.5, no_cpu_baseline=True, num_iter=20, cn_update="minsum",
                          also="none", no_extra=True, gpus=1)
which = sys.argv[1:] or ["c4", "c5_bp"]
out = {}
for name in which:
    fn = {"c4": bench.bench_c4, "c5_bp": bench.bench_c5_bp}[name]
    out[name] = fn(args, short=True)
print(json.dumps(out))
