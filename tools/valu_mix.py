#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel and the issue time it implies on gfx950 (development aid / roofline input).

    python tools/valu_mix.py <file.s> <kernel-name-substring> [...]  > profiles/r03_valu_mix.json

`file.s`: `hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S <tu>.hip`.  The VALU
instructions of the kernel's body are counted per issue class and weighted with the rates MEASURED on MI355X by
tools/ubench/valu_rate.hip (profiles/r03b/valu_rate_r03b.txt, ns per wave64 instruction and SIMD with 4 waves / SIMD and
8 independent chains): the peak of "1 instruction per 2 cycles" that the round-2 roofline used is reached by no
instruction of this kernel - v_add / v_sub / v_and / v_or / v_xor issue every ~2.7 cycles, everything else (v_fma, v_min,
v_med3, v_cndmask, shifts, 3-operand bit operations, conversions, packed fp32) every 4.2-4.8, transcendentals every 8.3.
The static mix stands for the dynamic one (the unrolled bodies all have the same composition; what differs is how often
each runs)."""
import json
import re
import sys

# ns per wave64 instruction and SIMD (profiles/r03b/valu_rate_r03b.txt); the nominal-clock cycle figures of that file
# divided by 2.4 GHz
FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_mov_b32", "v_add_co_u32", "v_sub_co_u32", "v_not_b32", "v_addc_co_u32", "v_subb_co_u32"}
NS = {"full": 1.14, "half": 1.85, "trans": 3.46, "cmp": 1.45, "cndmask": 1.45, "pk": 1.98}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def classify(op):
    op = re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)
    if op in TRANS:
        return "trans"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "cndmask"
    if op in FULL:
        return "full"
    return "half"


def kernel_body(lines, sub):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    with open(path) as f:
        lines = f.read().split("\n")
    out = {"_comment": "static VALU mix x measured issue times (tools/valu_mix.py, rates: profiles/r03b/valu_rate_r03b.txt)",
           "ns_per_class": NS, "kernels": {}}
    for sub in subs:
        body = kernel_body(lines, sub)
        counts, ops = {}, {}
        for l in body:
            m = re.match(r"\s+(v_[a-z0-9_]+)", l)
            if not m or m.group(1).startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                continue
            c = classify(m.group(1))
            counts[c] = counts.get(c, 0) + 1
            ops[m.group(1)] = ops.get(m.group(1), 0) + 1
        n = sum(counts.values())
        ns = sum(counts[c] * NS[c] for c in counts) / max(n, 1)
        out["kernels"][sub] = {"valu_insts_static": n, "class_counts": counts, "ns_per_valu_inst_est": round(ns, 3),
                               "cycles_at_2p4ghz": round(ns * 2.4, 2),
                               "top_ops": dict(sorted(ops.items(), key=lambda kv: -kv[1])[:14])}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
