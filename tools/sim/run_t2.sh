cd $GRAFT_REPO_ROOT
for v in _r04c ""; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  echo "variant: ${v:-new}"
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "false_header" 2>&1 | grep -v "^W2026" | grep -i -m5 "fault\|passed\|failed\|Aborted\|assert"
done
