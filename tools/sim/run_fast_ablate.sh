cd $GRAFT_REPO_ROOT
for v in "" _f_nowrite _f_nostage _f_nostore; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-full}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 2>&1 | grep -v amdgpu.ids | grep lean= | cut -c1-110
done
