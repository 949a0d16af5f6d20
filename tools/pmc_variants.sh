#!/bin/bash
# Instruction counts (SQ counters, one --pmc pass) and kernel times of experimental builds of the library.
#   tools/pmc_variants.sh <workload> name1 name2 ...     (tools/_ablate/lib_<name>.so, "default" = the shipped build)
WL=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcvar_$WL
mkdir -p $OUT
cd /tmp
for name in "$@"; do
  if [ "$name" = default ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$ROOT/tools/_ablate/lib_$name.so; fi
  python3 $ROOT/tools/time_decode_sub.py $WL 10 > $OUT/time_$name.log 2>&1
  tail -1 $OUT/time_$name.log
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o $name -- python3 $ROOT/tools/time_decode_sub.py $WL 3 > $OUT/pmc_$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    for k, d in agg.items():
        if "decode_sub" not in k and "decode_fast" not in k: continue
        print(os.path.basename(f).split("_counter")[0], k, {c: f"{v / len(calls[k]):.3e}" for c, v in d.items()})
PY
