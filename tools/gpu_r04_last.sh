#!/bin/bash
# round 4, the one trip that the remaining GPU minutes allow: the WHOLE GPU suite (old + the second half's new tests) spread
# over 8 worker processes (pytest-xdist; -v so that a cut-off run still shows what finished), then the default bench line
# (with the new c5_bp sub-line).  No rocprofv3 / PMC passes: no minutes for them; profiles/counters.json keeps r04zz's.
TAG=${1:-r04last}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (xdist)"
timeout 420 python -m pytest tests -m gpu -v -n 8 --dist load --tb=short -r fEx -p no:cacheprovider > gpurun_out/pytest_$TAG.txt 2>&1
echo "pytest rc $?"
grep -E "FAILED|ERROR|crashed" gpurun_out/pytest_$TAG.txt | head -40
tail -3 gpurun_out/pytest_$TAG.txt
echo "== bench"
timeout 200 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc $?"; head -c 400 gpurun_out/bench_$TAG.json; echo; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
    print({k: (v.get("value"), v.get("ms_per_step"), v.get("bler"), v.get("error")) for k, v in d.get("extra", {}).items()})
except Exception as e:
    print("bench line unreadable:", e)
PY
