#!/usr/bin/env python3
"""Convert the 3GPP TS 38.212 base-graph shift tables into a compact binary table.

The numbers are the standard's Tables 5.3.2-2 / 5.3.2-3 (base graph 1 / 2): for every
non-empty base-graph entry (row, column) the circular shift for each of the 8 lifting
set indices i_LS.  The reference ships the same tables as ``5G_bg{1,2}.csv``
(/root/reference/src/sionna/phy/fec/ldpc/codes, parsed at
src/sionna/phy/fec/ldpc/encoding.py:284-320); this script reads those CSVs once and
stores ``row``, ``col`` (int16) and ``shift[8]`` (int16) arrays per base graph in
``sionna_amd/phy/fec/ldpc/codes/bg_tables.npz`` so that nothing under /root/reference
is needed at run time (it does not exist on the GPU box).

Usage:  python tools/gen_tables.py [/root/reference]
"""
import os
import sys
import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = os.path.join(ref, "src/sionna/phy/fec/ldpc/codes")
out = {}
for bg, (nrow, ncol) in (("bg1", (46, 68)), ("bg2", (42, 52))):
    rows, cols, shifts = [], [], []
    r_ind = 0
    with open(os.path.join(src, f"5G_{bg}.csv")) as f:
        lines = f.read().splitlines()[2:]          # two header lines
    for ln in lines:
        t = ln.split(";")
        if t[0].strip() != "":
            r_ind = int(t[0])
        rows.append(r_ind)
        cols.append(int(t[1]))
        shifts.append([int(v) for v in t[2:10]])
    rows = np.array(rows, np.int16)
    cols = np.array(cols, np.int16)
    shifts = np.array(shifts, np.int16)
    assert rows.max() == nrow - 1 and cols.max() == ncol - 1
    out[f"{bg}_row"], out[f"{bg}_col"], out[f"{bg}_shift"] = rows, cols, shifts
    print(bg, "entries:", len(rows))
dst = os.path.join(os.path.dirname(__file__), "..", "sionna_amd/phy/fec/ldpc/codes/bg_tables.npz")
np.savez_compressed(dst, **out)
print("wrote", os.path.normpath(dst))
