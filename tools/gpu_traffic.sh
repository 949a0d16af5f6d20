#!/bin/bash
# HBM traffic of every kernel of one bench step from the TCC counters (separate --pmc passes as
# MI355X_MICROARCH.md prescribes; FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
# wide coalesced reads at half their bytes -> doubled below, WRITE_SIZE is exact for 16-B stores).
WL=${1:-const41}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/traffic_$WL
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o rd -- python3 $ROOT/bench.py --steps 3 --warmup 1 --workload $WL --no-cpu-baseline --no-verify > $OUT/rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o wr -- python3 $ROOT/bench.py --steps 3 --warmup 1 --workload $WL --no-cpu-baseline --no-verify > $OUT/wr.log 2>&1
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for tag, ctr in (("rd", "FETCH_SIZE"), ("wr", "WRITE_SIZE")):
    f = glob.glob("$OUT/%s_counter_collection.csv" % tag)[0]
    agg = collections.defaultdict(float); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        agg[k] += float(r["Counter_Value"]); calls[k] += 1
    for k in agg: res[k][ctr + "_KiB_per_launch"] = agg[k] / calls[k]
out = {}
for k, d in res.items():
    if not k.endswith("kernel"): continue
    rd = d.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024 * 2     # gfx950 correction
    wr = d.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
    out[k] = {"read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes": rd + wr, **d}
json.dump({"workload": "$WL", "kernels": out}, open("$OUT/traffic.json", "w"), indent=1)
for k, v in out.items(): print(k, "read %.1f MiB  write %.1f MiB" % (v["read_bytes_corrected"] / 2**20, v["write_bytes"] / 2**20))
PY
