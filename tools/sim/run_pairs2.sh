cd $GRAFT_REPO_ROOT
HUF_LIB_PATH=$PWD/tools/_ablate/lib_dfastdbg.so python tools/dbg_dfast.py zipf255 uniform256 uniform255 logtext 2>&1 | grep -v amdgpu.ids
for v in "" _jump2; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-pairs}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 2>&1 | grep -v amdgpu.ids | cut -c1-400
  timeout 300 python tools/time_lean.py --mib 256 logtext 2>&1 | grep -v amdgpu.ids | cut -c1-400
  timeout 300 python tools/time_runs_selfsync.py 0 4096 16384 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
unset HUF_LIB_PATH
bash tools/sim/run_pmc_fast.sh 2>&1 | tail -6
