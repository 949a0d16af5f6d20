cd $GRAFT_REPO_ROOT
for v in _skel2 _skel3 _tabonly; do
  export HUF_LIB_PATH=$PWD/tools/_ablate/lib_lean$v.so
  echo "variant: ${v:-full}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 2>&1 | grep -v amdgpu.ids | cut -c1-110 | head -1
done
