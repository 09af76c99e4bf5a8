#!/bin/bash
# Polar SCL-8 (config C5): does the L2 scratch footprint - and with it the memory-side traffic - limit the decode rate?
# Sweeps the resident set (workgroups per CU x 40 KB of top-stage scratch per resident codeword) and the number of tree
# stages kept in L2; per setting: the bench line's rate and ms, then FETCH_SIZE / WRITE_SIZE of polar_scl_reg_kernel in
# two separate PMC passes (MI355X_MICROARCH.md: 2 x FETCH + WRITE).  Output: gpurun_out/<tag>_c5_footprint.txt
TAG=${1:-r04}
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/${TAG}_c5_footprint
mkdir -p $OUT
cd /tmp
{
echo "# per_cu  gstages  resident_MB  decodes/s  ms_per_launch  FETCH_KB  WRITE_KB  hbm_GB_per_launch(2F+W)"
for cfg in ${CFGS:-"32 4" "24 4" "16 4" "12 4" "8 4" "4 4" "32 3" "16 3" "32 2" "32 5"}; do
  set -- $cfg; pc=$1; g=$2
  line=$(SAMD_SCL_PER_CU=$pc SAMD_SCL_GSTAGES=$g timeout 300 python $ROOT/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1)
  rate=$(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"].get("ms_per_launch", d["ms_per_step"]))')
  for c in FETCH_SIZE WRITE_SIZE; do
    SAMD_SCL_PER_CU=$pc SAMD_SCL_GSTAGES=$g timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/p_${pc}_${g}_$c -o p -- \
      python $ROOT/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/p_${pc}_${g}_$c.log 2>&1
  done
  fw=$(python - <<PY
import csv, glob
def mean(c):
    v = []
    for f in glob.glob("$OUT/p_${pc}_${g}_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "polar_scl_reg_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    return sum(v) / len(v) if v else float("nan")
f, w = mean("FETCH_SIZE"), mean("WRITE_SIZE")
print(f"{f:.4g} {w:.4g} {(2 * f + w) * 1024 / 1e9:.3f}")
PY
)
  mb=$(python -c "print(round(256*$pc*40/1024,1))")
  echo "$pc $g $mb $rate $fw"
done
} | tee $ROOT/gpurun_out/${TAG}_c5_footprint.txt
rm -rf $OUT/p_*/  # keep the text summary only
