cd $GRAFT_REPO_ROOT
for v in "" 96 128 160 192; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib_pre$v.so; fi
  echo "pre-run-in: ${v:-0}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 2>&1 | grep -v amdgpu.ids | cut -c1-110
  timeout 300 python tools/time_lean.py --mib 256 logtext logtext@1m 2>&1 | grep -v amdgpu.ids | cut -c1-110
done
