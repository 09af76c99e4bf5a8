"""Every ``path/file.py:first-last`` citation of the reference in include/sionna_amd.h, the documents, the kernels, the host
package and the oracle names a file that exists under /root/reference/src/sionna/phy and lines that exist in it.  Runs where the
reference is present (this container); skipped on the GPU box."""
import glob
import os
import re

import pytest

REF = "/root/reference/src/sionna/phy"
ROOT = os.path.join(os.path.dirname(__file__), "..")
CITE = re.compile(r"((?:[a-z0-9_]+/)*[a-z0-9_]+\.py):(\d+)(?:-(\d+))?")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_cited_lines_exist():
    by_name = {}
    for root, _, files in os.walk(REF):
        for f in files:
            if f.endswith(".py"):
                by_name.setdefault(f, []).append(os.path.relpath(os.path.join(root, f), REF))
    length = {}
    sources = ["include/sionna_amd.h", "DESIGN.md", "COVERAGE.md", "INTEGRATION.md", "README.md"]
    for pattern in ("sionna_amd/phy/**/*.py", "oracle/*.py", "oracle/*.c", "sionna_amd/csrc/*.hip", "sionna_amd/csrc/*.h", "sionna_amd/csrc/*.cpp"):
        sources += [os.path.relpath(p, ROOT) for p in glob.glob(os.path.join(ROOT, pattern), recursive=True)]
    checked, bad = 0, []
    for src in sources:
        with open(os.path.join(ROOT, src)) as f:
            text = f.read()
        for m in CITE.finditer(text):
            path = m.group(1)
            for prefix in ("root/reference/src/sionna/phy/", "src/sionna/phy/", "sionna/phy/"):
                if path.startswith(prefix):
                    path = path[len(prefix):]
            cands = [r for r in by_name.get(os.path.basename(path), []) if r.endswith(path)]
            if not cands:
                continue                      # a file of this repository or of the reference's tests
            last = int(m.group(3) or m.group(2))
            for c in cands:
                if c not in length:
                    with open(os.path.join(REF, c)) as g:
                        length[c] = sum(1 for _ in g)
            checked += 1
            if all(length[c] < last for c in cands):
                bad.append(f"{src}: {m.group(0)} (the file has {max(length[c] for c in cands)} lines)")
    assert checked > 500 and not bad, "\n".join(bad[:20])


def test_tests_named_in_the_documents_exist():
    """every ``::test_name`` that COVERAGE.md / DESIGN.md / README.md quote is a test function of this suite (a name ending in ``_`` or
    followed by an ellipsis is a prefix)"""
    defined = set()
    for p in glob.glob(os.path.join(ROOT, "tests", "*.py")):
        with open(p) as f:
            defined |= set(re.findall(r"^def (test_\w+)", f.read(), re.M))
    missing, seen = [], 0
    for doc in ("COVERAGE.md", "DESIGN.md", "README.md"):
        with open(os.path.join(ROOT, doc)) as f:
            text = f.read()
        for m in re.finditer(r"::(test_\w+)(…|\.\.\.)?", text):
            name, prefix = m.group(1), bool(m.group(2)) or m.group(1).endswith("_")
            seen += 1
            if not (name in defined or (prefix and any(d.startswith(name) for d in defined))):
                missing.append(f"{doc}: {name}")
    assert seen > 50 and not missing, missing
