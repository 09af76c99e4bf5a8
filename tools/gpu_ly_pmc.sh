#!/bin/bash
# instruction / scalar-data cache and wait counters of the layered on-chip kernel (layered-10 min-sum at C2)
TAG=${1:-r03x}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lypmc_$TAG
mkdir -p $OUT
cd /tmp
cat > /tmp/ly_one.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import sionna_amd.phy as phy
k, n, m, B = 2816, 8448, 6, 4096
phy.config.seed = 1
enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
no = phy.utils.ebnodb2no(4.5, m, k / n)
u = phy.mapping.BinarySource()([B, k])
llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=os.environ.get("LY_CN", "minsum"), num_iter=10, cn_schedule="layered")
for _ in range(3): dec(llr)
torch.cuda.synchronize()
PY
run() { local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python /tmp/ly_one.py > $OUT/$name.log 2>&1; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ly_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]][row["Dispatch_Id"]].append(float(row["Counter_Value"]))
for c, d in sorted(acc.items()):
    v = [sum(x) for x in d.values()]
    print(f"{c:32s} {sum(v)/len(v):16.0f}  (mean of {len(v)} launches of 4096 decodes)")
PY
