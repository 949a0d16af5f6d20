#!/bin/bash
# Runs tools/diag_pack.py against every library under libhuffman_amd/_variants/ (built with
# HUF_LIB_PATH=... HUF_EXTRA_FLAGS=... python -m libhuffman_amd.build) and against the default build.
# usage: tools/diag_pack_variants.sh [seconds per variant] [variant names...]
secs=${1:-45}; shift
out=gpurun_out/diag; mkdir -p $out
names="$@"; [ -z "$names" ] && names="$(ls libhuffman_amd/_variants/*.so | xargs -n1 basename | sed 's/\.so$//') default"
for v in $names; do
  if [ $v = default ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/libhuffman_amd/_variants/$v.so; fi
  timeout $((secs + 120)) python tools/diag_pack.py $secs 7 3 > $out/$v.log 2>&1
  echo "== $v: rc $? $(tail -n 1 $out/$v.log)"
done
