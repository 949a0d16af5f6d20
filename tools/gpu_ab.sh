#!/bin/bash
# One parameterised A/B script for the MI355X box (replaces the run_*.sh one-offs of rounds 3-4):
#   tools/gpu_ab.sh [-t "<pytest args>"] [-w "zipf255 uniform256"] [-r reps] [-p "<kernel name part> ..."] [-s script.py] name1 name2 ...
# Every name is a build of the library: "default" = libhuffman_amd/libhuffman.so, anything else
# tools/_ablate/lib_<name>.so (tools/ablate_decode.sh builds those HERE before the call; they travel with the snapshot).
#   -t  run pytest with these arguments first (quoted), log -> gpurun_out/ab/pytest.log
#   -w  workloads for tools/time_decode_sub.py (encode + sub-index decode of 1 GiB, kernel times, round-trip check)
#   -r  repetitions per timing (default 10)
#   -p  also collect SQ counters (two --pmc passes) for kernels whose name contains one of these parts
#   -s  another timing script to run per build instead of tools/time_decode_sub.py (gets the workloads as arguments)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/ab
mkdir -p $OUT
TESTS=""; WLS="zipf255"; REPS=10; PMC=""; SCRIPT=tools/time_decode_sub.py
while getopts "t:w:r:p:s:" o; do
  case $o in t) TESTS=$OPTARG;; w) WLS=$OPTARG;; r) REPS=$OPTARG;; p) PMC=$OPTARG;; s) SCRIPT=$OPTARG;; esac
done
shift $((OPTIND - 1))
if [ -n "$TESTS" ]; then
  ( time python -m pytest $TESTS ) > $OUT/pytest.log 2>&1
  tail -15 $OUT/pytest.log
fi
for round in 1 2; do          # twice, interleaved: a box has moods
  for name in "$@"; do
    if [ "$name" = default ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$ROOT/tools/_ablate/lib_$name.so; fi
    for wl in $WLS; do
      if [ "$SCRIPT" = tools/time_decode_sub.py ]; then
        echo -n "$name: "; python3 $SCRIPT $wl $REPS 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
      else
        echo -n "$name: "; python3 $SCRIPT $wl 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400
      fi
    done
  done
done | tee $OUT/times.txt
if [ -n "$PMC" ]; then
  cd /tmp
  for name in "$@"; do
    if [ "$name" = default ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$ROOT/tools/_ablate/lib_$name.so; fi
    for wl in $WLS; do
      rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1_${name}_$wl -o p -- python3 $ROOT/tools/time_decode_sub.py $wl 3 > $OUT/pmc1_${name}_$wl.log 2>&1
      rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/pmc2_${name}_$wl -o p -- python3 $ROOT/tools/time_decode_sub.py $wl 3 > $OUT/pmc2_${name}_$wl.log 2>&1
    done
  done
  cd $ROOT
  python3 - "$PMC" <<PY | tee $OUT/pmc.txt
import csv, glob, collections, os, sys
parts = sys.argv[1].split()
for d in sorted(glob.glob("$OUT/pmc[12]_*")):
    if not os.path.isdir(d): continue
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
        for k, c in agg.items():
            if any(x in k for x in parts):
                print(os.path.basename(d), k, "launches", len(calls[k]), {n: "%.3e" % (v / len(calls[k])) for n, v in sorted(c.items())})
PY
fi
