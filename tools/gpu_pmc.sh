#!/bin/bash
# SQ counters per kernel for one workload (two separate --pmc passes; no tracing domains mixed in).
#   tools/gpu_pmc.sh [workload] [python tool, default tools/time_decode_sub.py]
WL=${1:-zipf255}
TOOL=${2:-tools/time_decode_sub.py}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$WL
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o p1 -- python3 $ROOT/$TOOL $WL 3 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT -o p2 -- python3 $ROOT/$TOOL $WL 3 > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    for k, d in agg.items():
        if "fill" in k: continue
        print(k, "launches", len(calls[k]), {c: f"{v / len(calls[k]):.3e}" for c, v in d.items()})
PY
