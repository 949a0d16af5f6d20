cd $GRAFT_REPO_ROOT
for v in _r04c ""; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-pairs}"
  timeout 300 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 2>&1 | grep -v amdgpu.ids | cut -c1-400
  timeout 300 python tools/time_lean.py --mib 256 logtext logtext@1m zipf255@16k 2>&1 | grep -v amdgpu.ids | cut -c1-400
  python tools/time_raw_decode.py 2>&1 | grep -v amdgpu.ids | tail -1
done
unset HUF_LIB_PATH
python tools/sim/dbg_lean2.py 2>&1 | grep -v "amdgpu.ids" | grep -v " ok$" | tail -5
python tools/sim/dbg_lean.py 2>&1 | grep -v "amdgpu.ids" | grep -v " ok$" | tail -5
python -m pytest tests/test_gpu_parity.py tests/test_gpu_subindex.py tests/test_gpu_bigraw.py -m gpu -x -q 2>&1 | tail -3
