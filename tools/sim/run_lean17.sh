cd $GRAFT_REPO_ROOT
for v in _r16 _r21 _w16; do
  export HUF_LIB_PATH=$PWD/tools/_ablate/lib_lean$v.so
  echo "variant: ${v:-full}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 2>&1 | grep -v amdgpu.ids | cut -c1-110
done
