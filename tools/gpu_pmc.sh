#!/bin/bash
# PMC counters for one workload via tools/time_decode.py (separate passes; no tracing domains mixed in)
WL=${1:-zipf255}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$WL
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o p1 -- python3 $ROOT/tools/time_decode.py $WL > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT -o p2 -- python3 $ROOT/tools/time_decode.py $WL > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        if "decode_kernel" in k or "pack_kernel" in k or "hist256" in k or "tree" in k:
            print(k, {c: f"{v:.3e}" for c, v in d.items()})
PY
