cd $GRAFT_REPO_ROOT
for v in _cur "" _cur "" _cur ""; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo -n "variant: ${v:-lutfirst}  "
  python tools/time_decode_sub.py zipf255 20 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
