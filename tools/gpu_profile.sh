#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summaries -> gpurun_out/prof_<tag>
TAG=${1:-r01}
WL=${2:-const41}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --steps 5 --warmup 2 --workload $WL --no-cpu-baseline > $OUT/bench.json 2> $OUT/stderr.log
tail -1 $OUT/bench.json
find $OUT -name "*kernel_stats*" | head -3
cat $(find $OUT -name "*kernel_stats.csv" | head -1) | head -20
