cd $GRAFT_REPO_ROOT
for v in "" _hl32; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-default(16-bit)}"
  python tools/time_encode_stages.py 2>&1 | grep -v amdgpu.ids | tail -6
done
unset HUF_LIB_PATH
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "histogram or encode or blocks_of_4_mib or full_size_stream" 2>&1 | tail -3
