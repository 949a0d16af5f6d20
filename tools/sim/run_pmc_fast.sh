cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fast
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -o p1 -- python3 $GRAFT_REPO_ROOT/tools/time_lean.py --mib 1024 zipf255 uniform256 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/sq2 -o p2 -- python3 $GRAFT_REPO_ROOT/tools/time_lean.py --mib 1024 zipf255 uniform256 > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/sq*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    for k, d in agg.items():
        if "decode_fast_kernel" in k or "pack_kernel" in k:
            print("%s launches %d per launch: %s" % (k, len(calls[k]), {c: "%.3e" % (v / len(calls[k])) for c, v in d.items()}))
PY
