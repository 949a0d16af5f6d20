cd $GRAFT_REPO_ROOT
for cfg in "2 8192 3000 1 0" "4 8192 3000 1 0" "4 60000 50000 1 0" "4 60000 17 1 0"; do
  timeout 120 python tools/sim/dbg_false.py $cfg 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -2 | cut -c1-200
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigraw.py -m gpu -x -q 2>&1 | tail -3
python tools/time_raw_decode.py zipf255 uniform256 2>&1 | grep -v amdgpu.ids | tail -2
