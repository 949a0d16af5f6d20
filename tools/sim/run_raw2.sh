cd $GRAFT_REPO_ROOT
python tools/time_raw_decode.py zipf255 uniform256 2>&1 | grep -v amdgpu.ids | tail -2
HUF_LIB_PATH=$PWD/tools/_ablate/lib_r04c.so python tools/time_raw_decode.py zipf255 uniform256 2>&1 | grep -v amdgpu.ids | tail -2
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigraw.py tests/test_huffmanfile.py -m gpu -x -q 2>&1 | tail -3
bash tools/sim/run_rawprof.sh 2>&1 | grep -v "^W2026" | head -16
