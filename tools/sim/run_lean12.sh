cd $GRAFT_REPO_ROOT
for v in "" _noSETTLE _noRUNIN _noIMAGE _noFLUSH _noSHARE _noSS; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib_lean$v.so; fi
  echo "variant: ${v:-full}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 2>&1 | grep -v amdgpu.ids | cut -c1-140
done
