cd $GRAFT_REPO_ROOT
for v in _chains1 ""; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-chains2}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 2>&1 | grep -v amdgpu.ids | cut -c1-300
  timeout 300 python tools/time_lean.py --mib 256 logtext logtext@1m zipf255@16k zipf255@4k 2>&1 | grep -v amdgpu.ids | cut -c1-300
  python tools/time_raw_decode.py 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python tools/time_runs_selfsync.py 0 4096 16384 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
unset HUF_LIB_PATH
HUF_LIB_PATH=$PWD/tools/_ablate/lib_dfastdbg.so python tools/dbg_dfast.py zipf255 uniform256 logtext 2>&1 | grep -v amdgpu.ids | cut -c1-600
python tools/sim/dbg_lean2.py 2>&1 | grep -v "amdgpu.ids" | grep -v " ok$" | tail -5
python tools/sim/dbg_lean.py 2>&1 | grep -v "amdgpu.ids" | grep -v " ok$" | tail -5
python -m pytest tests/test_gpu_parity.py tests/test_gpu_subindex.py tests/test_gpu_bigraw.py -m gpu -x -q 2>&1 | tail -5
HUF_LIB_PATH=$PWD/tools/_ablate/lib_fastprof.so python tools/phase_fast.py zipf255 uniform256 2>&1 | grep -v amdgpu.ids
