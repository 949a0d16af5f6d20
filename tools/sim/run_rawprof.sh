cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rawprof
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o raw -- python3 $GRAFT_REPO_ROOT/tools/time_raw_decode.py > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python3 - <<PY
import csv, glob
for p in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(p)))
    for r in rows[:24]:
        print("%-60s calls %5s avg %10.1f us total %10.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
