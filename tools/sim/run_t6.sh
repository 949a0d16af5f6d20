cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bigraw.py tests/test_huffmanfile.py -m gpu -x -q 2>&1 | tail -3
python tools/time_raw_decode.py zipf255 uniform256 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/sim/run_rawprof.sh 2>&1 | grep -v "^W2026" | grep "discover\|probe\|scan_counts\|walk\|cand_lens\|link"
