cd $GRAFT_REPO_ROOT
for v in "" _c64 _c128 _c0; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-default(96)}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 2>&1 | grep -v amdgpu.ids | cut -c1-150
  timeout 300 python tools/time_lean.py --mib 256 logtext logtext@1m zipf255@16k 2>&1 | grep -v amdgpu.ids | cut -c1-150
done
unset HUF_LIB_PATH
python tools/sim/dbg_lean2.py 2>&1 | grep -v "amdgpu.ids" | grep -v " ok$"
python tools/sim/dbg_lean.py 2>&1 | grep -v "amdgpu.ids" | grep -v " ok$"
