cd $GRAFT_REPO_ROOT
for v in "" _n3 _n2 _n1; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-n4}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
