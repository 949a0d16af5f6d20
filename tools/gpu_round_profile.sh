#!/bin/bash
# Everything profiles/<tag>/ holds, collected on the MI355X box in one go:
#   tools/gpu_round_profile.sh r02x [workloads...]        (default: zipf255 uniform256 const41)
# 1. rocprofv3 --kernel-trace --stats of `bench.py --workload W --secondary none` per workload
#    -> <tag>_<W>_kernel_stats.csv + <tag>_<W>_bench.json (the JSON line of the same process)
# 2. HBM traffic per kernel launch from the TCC counters, separate --pmc passes (FETCH_SIZE doubled:
#    the gfx950 correction of MI355X_MICROARCH.md) -> traffic.json, stamped with the kernel sources' digest
# 3. SQ counters of every kernel on zipf255 (two --pmc passes) -> <tag>_pmc_zipf255.txt
# 4. the default bench command with no profiler attached -> <tag>_final_bench.json
TAG=${1:-r02}
shift
WLS=${@:-zipf255 uniform256 const41}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp
COMMON="--secondary none --no-cpu-baseline --no-other-decode --no-index-free --no-python-layer --no-live-traffic"
for WL in $WLS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$WL -o run -- python3 $ROOT/bench.py --steps 20 --warmup 5 --workload $WL $COMMON > $OUT/${TAG}_${WL}_bench.json 2> $OUT/kt_$WL.err
  cp $(find $OUT/kt_$WL -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${WL}_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rd_$WL -o rd -- python3 $ROOT/bench.py --steps 3 --warmup 1 --workload $WL $COMMON --no-verify > $OUT/rd_$WL.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/wr_$WL -o wr -- python3 $ROOT/bench.py --steps 3 --warmup 1 --workload $WL $COMMON --no-verify > $OUT/wr_$WL.log 2>&1
done
# the same step with the block index alone (decode_fast_kernel): kernel stats and traffic of zipf255
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_index -o run -- python3 $ROOT/bench.py --steps 10 --warmup 3 --workload zipf255 --decode selfsync $COMMON > $OUT/${TAG}_zipf255_index_bench.json 2> $OUT/kt_index.err
cp $(find $OUT/kt_index -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_zipf255_index_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rd_index -o rd -- python3 $ROOT/bench.py --steps 3 --warmup 1 --workload zipf255 --decode selfsync $COMMON --no-verify > $OUT/rd_index.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/wr_index -o wr -- python3 $ROOT/bench.py --steps 3 --warmup 1 --workload zipf255 --decode selfsync $COMMON --no-verify > $OUT/wr_index.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -o p1 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --workload zipf255 $COMMON --no-verify > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/sq2 -o p2 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --workload zipf255 $COMMON --no-verify > $OUT/sq2.log 2>&1
python3 $ROOT/bench.py > $OUT/${TAG}_final_bench.json 2> $OUT/final.err
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$ROOT")
import bench
names = {"hist_tree_kernel": "hist_tree", "hist_lanes_kernel": "hist256", "tree_wave_kernel": "tree", "pack_kernel": "pack",
         "decode_sub_kernel": "decode", "decode_kernel": "decode_selfsync", "decode_fast_kernel": "decode_index",
         "decode_prepare_kernel": "prepare_scan", "decode_fix_kernel": "decode_fix"}
def per_kernel(path, ctr):
    agg = collections.defaultdict(float); calls = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        agg[k] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
    return {k: agg[k] / len(calls[k]) for k in agg}
out = {"kernel_source_digest": bench.kernel_source_digest(), "blocksize": 65536, "bytes_per_gpu": 1 << 30,
       "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of bench.py (tools/gpu_round_profile.sh); "
              "KiB per launch averaged over the launches; FETCH_SIZE doubled (gfx950 counts wide coalesced reads at half their bytes)",
       "workloads": {}}
for wl in "$WLS".split():
    try:
        rd = per_kernel(glob.glob("$OUT/rd_%s/*counter_collection.csv" % wl)[0], "FETCH_SIZE")
        wr = per_kernel(glob.glob("$OUT/wr_%s/*counter_collection.csv" % wl)[0], "WRITE_SIZE")
    except Exception as e:
        print("traffic of", wl, "missing:", e); continue
    d = {}
    for k in set(rd) | set(wr):
        if k not in names: continue
        r, w = rd.get(k, 0.0) * 1024 * 2, wr.get(k, 0.0) * 1024
        d[names[k]] = {"read": round(r), "write": round(w), "hbm": round(r + w)}
    out["workloads"][wl] = d
    print(wl, {k: "%.1f / %.1f MiB" % (v["read"] / 2**20, v["write"] / 2**20) for k, v in d.items()})
try:
    rd = per_kernel(glob.glob("$OUT/rd_index/*counter_collection.csv")[0], "FETCH_SIZE")
    wr = per_kernel(glob.glob("$OUT/wr_index/*counter_collection.csv")[0], "WRITE_SIZE")
    r, w = rd.get("decode_fast_kernel", 0.0) * 1024 * 2, wr.get("decode_fast_kernel", 0.0) * 1024
    out["workloads"]["zipf255"]["decode_index"] = {"read": round(r), "write": round(w), "hbm": round(r + w)}
    print("zipf255 decode_fast_kernel (block index alone): %.1f / %.1f MiB" % (r / 2**20, w / 2**20))
except Exception as e:
    print("traffic of the index-alone decode missing:", e)
json.dump(out, open("$OUT/traffic.json", "w"), indent=1)
with open("$OUT/${TAG}_pmc_zipf255.txt", "w") as f:
    for p in sorted(glob.glob("$OUT/sq*/*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
        for k, d in agg.items():
            if k.split("<")[0] in names:
                f.write("%s launches %d per launch: %s\n" % (k, len(calls[k]), {c: "%.3e" % (v / len(calls[k])) for c, v in d.items()}))
print(open("$OUT/${TAG}_pmc_zipf255.txt").read())
PY
tail -c 600 $OUT/${TAG}_final_bench.json
ls $OUT
