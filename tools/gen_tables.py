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

# ---- TDL power-delay profiles (3GPP TR 38.901 Tables 7.7.2-1..5 and the A30/B100/C300 variants
# of TS 38.101/38.104), shipped by the reference as channel/tr38901/models/TDL-*.json
# (parsed at channel/tr38901/tdl.py:539-592).  Re-packed into one JSON.
import json
tdl = {}
mdir = os.path.join(ref, "src/sionna/phy/channel/tr38901/models")
for name in ("A", "B", "C", "D", "E", "A30", "B100", "C300"):
    with open(os.path.join(mdir, f"TDL-{name}.json")) as f:
        p = json.load(f)
    tdl[name] = {"los": int(p["los"]), "scale_delays": int(p["scale_delays"]),
                 "num_clusters": int(p["num_clusters"]), "delays": [float(v) for v in p["delays"]],
                 "powers": [float(v) for v in p["powers"]]}
dst = os.path.join(os.path.dirname(__file__), "..", "sionna_amd/phy/channel/tr38901/tdl_models.json")
with open(dst, "w") as f:
    json.dump(tdl, f)
print("wrote", os.path.normpath(dst))

# ---- 5G Polar sequence (38.212 Table 5.3.1.2-1): reliability order of the 1024 bit channels,
# shipped by the reference as fec/polar/codes/polar_5G.csv (rows "W;Q": reliability rank W and
# channel index Q; parsed at fec/polar/utils.py:72-95).  Stored as int16 array q[w] = channel index.
rows = np.genfromtxt(os.path.join(ref, "src/sionna/phy/fec/polar/codes/polar_5G.csv"), delimiter=";").astype(int)
q = np.zeros(1024, np.int16)
q[rows[:, 0]] = rows[:, 1]
assert sorted(q.tolist()) == list(range(1024))
dst = os.path.join(os.path.dirname(__file__), "..", "sionna_amd/phy/fec/polar/codes/polar_5g_sequence.npy")
os.makedirs(os.path.dirname(dst), exist_ok=True)
np.save(dst, q)
print("wrote", os.path.normpath(dst))


# ---- example parity-check matrices of fec/utils.load_parity_check_examples (coordinate form)
ex = np.load(os.path.join(ref, "src/sionna/phy/fec/ldpc/codes/example_codes.npy"), allow_pickle=True)
pk = {}
for i in range(len(ex)):
    pcm = np.array(ex[i])
    r, c = np.nonzero(pcm)
    pk[f"shape_{i}"] = np.array(pcm.shape, np.int32)
    pk[f"rc_{i}"] = np.stack([r, c]).astype(np.int32)
np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 "sionna_amd/phy/fec/ldpc/codes/example_pcms.npz"), **pk)
print("example pcms:", [tuple(pk[f"shape_{i}"]) for i in range(len(ex))])

# ---- CDL cluster tables (3GPP TR 38.901 Tables 7.7.1-1..5), shipped by the reference as
# channel/tr38901/models/CDL-*.json (parsed at channel/tr38901/cdl.py:385-555).  Re-packed into one JSON.
cdl = {}
for name in "ABCDE":
    with open(os.path.join(mdir, f"CDL-{name}.json")) as f:
        p = json.load(f)
    cdl[name] = {k: (int(p[k]) if k in ("los", "num_clusters") else
                     ([float(v) for v in p[k]] if isinstance(p[k], list) else float(p[k]))) for k in p}
dst = os.path.join(os.path.dirname(__file__), "..", "sionna_amd/phy/channel/tr38901/cdl_models.json")
with open(dst, "w") as f:
    json.dump(cdl, f, indent=0)
print("wrote", os.path.normpath(dst))
