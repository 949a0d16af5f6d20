cd $GRAFT_REPO_ROOT
for v in _nopref "" _nopref ""; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-prefetch}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 2>&1 | grep -v amdgpu.ids | cut -c1-300
  timeout 300 python tools/time_lean.py --mib 256 logtext zipf255@16k 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
