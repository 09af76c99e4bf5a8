#!/usr/bin/env python3
"""Extract every saved ``sim_ber`` table from the reference's tutorial notebooks into
``tests/golden/notebook_ber.json``.

The reference ships its notebooks WITH their outputs: each ``sim_ber`` / ``PlotBER.simulate`` call printed the
table of ``/root/reference/src/sionna/phy/utils/misc.py:524-554`` (EbNo, BER, BLER, bit errors, num bits, block
errors, num blocks, runtime, status).  Those tables are reference-produced numbers (TensorFlow, NVIDIA GPU, the
reference's own kernels) for exactly the link-level paths SURVEY.md section 8 scopes, so they are the only
reference OUTPUT for BER/BLER available offline.  This script only reads ``/root/reference/tutorials/phy/*.ipynb``
(in THIS container) and writes a small JSON fixture; nothing at test time touches /root/reference.

Each table is keyed ``<notebook stem>/c<cell index>/t<ordinal in cell>`` and keeps: the label printed before
it (``Running: ...`` lines or the last non-empty line printed before the header), the rows, and the first
line number of the table inside the .ipynb file (for file:line citations).

Run:  python tools/extract_notebook_tables.py            (rewrites the fixture)
      python tools/extract_notebook_tables.py --list     (prints one line per table)
"""
import argparse
import glob
import json
import os
import re
import sys

REF_NB = "/root/reference/tutorials/phy"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "notebook_ber.json")

ROW = re.compile(
    r"^\s*(-?\d+(?:\.\d+)?)\s*\|\s*([0-9.eE+-]+)\s*\|\s*([0-9.eE+-]+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|"
    r"\s*([0-9.]+)\s*\|\s*(.*?)\s*$")


def cell_text(cell):
    """All stream/text outputs of a code cell, concatenated in order."""
    parts = []
    for o in cell.get("outputs", []):
        if "text" in o:
            parts.append("".join(o["text"]))
        elif "data" in o and "text/plain" in o["data"]:
            parts.append("".join(o["data"]["text/plain"]))
    return "\n".join(parts)


def parse_tables(text):
    """Yield (label, rows) for every sim_ber table in a block of printed text."""
    lines = text.splitlines()
    i, last_label = 0, ""
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("EbNo [dB]") and "BLER" in ln:
            rows = []
            j = i + 1
            if j < len(lines) and set(lines[j].strip()) <= {"-"}:
                j += 1
            while j < len(lines):
                m = ROW.match(lines[j])
                if not m:
                    break
                g = m.groups()
                rows.append({"ebno_db": float(g[0]), "ber": float(g[1]), "bler": float(g[2]),
                             "bit_errors": int(g[3]), "num_bits": int(g[4]), "block_errors": int(g[5]),
                             "num_blocks": int(g[6]), "status": g[8], "_raw": lines[j].strip()[:60]})
                j += 1
            yield last_label, rows
            i = j
            continue
        s = ln.strip()
        if s and not s.startswith("Simulation stopped") and not s.startswith("Note:") and not s.startswith("Warning"):
            last_label = s[len("Running: "):] if s.startswith("Running: ") else s
        i += 1


def first_line_of(path, needle_rows):
    """1-based line in the .ipynb file of the first data row of a table (for citations)."""
    if not needle_rows:
        return None
    needle = needle_rows[0]["_raw"]
    with open(path) as f:
        for n, ln in enumerate(f, 1):
            if needle in ln:
                return n
    return None


def extract(nb_dir=REF_NB):
    out = {}
    for path in sorted(glob.glob(os.path.join(nb_dir, "*.ipynb"))):
        stem = os.path.splitext(os.path.basename(path))[0]
        nb = json.load(open(path))
        for ci, cell in enumerate(nb["cells"]):
            if cell["cell_type"] != "code":
                continue
            txt = cell_text(cell)
            if "EbNo [dB]" not in txt:
                continue
            for ti, (label, rows) in enumerate(parse_tables(txt)):
                if not rows:
                    continue
                line = first_line_of(path, rows)
                for r in rows:
                    del r["_raw"]
                out[f"{stem}/c{ci}/t{ti}"] = {
                    "notebook": os.path.basename(path), "cell": ci, "ordinal": ti, "label": label,
                    "ipynb_line": line, "rows": rows}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--out", default=OUT)
    a = ap.parse_args()
    if not os.path.isdir(REF_NB):
        sys.exit(f"{REF_NB} not present (run in the build container)")
    tabs = extract()
    if a.list:
        for k, v in tabs.items():
            r = v["rows"]
            print(f"{k:70s} L{v['ipynb_line']}: {v['label'][:50]!r:54s} {len(r):2d} pts "
                  f"{r[0]['ebno_db']:.1f}..{r[-1]['ebno_db']:.1f} dB, blocks {r[0]['num_blocks']}")
        return
    doc = {"_source": "saved sim_ber outputs of /root/reference/tutorials/phy/*.ipynb (reference-produced; extracted by "
                      "tools/extract_notebook_tables.py)", "tables": tabs}
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
    print(f"{len(tabs)} tables, {sum(len(v['rows']) for v in tabs.values())} rows -> {a.out} ({os.path.getsize(a.out)} bytes)")


if __name__ == "__main__":
    main()
