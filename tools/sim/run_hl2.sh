cd $GRAFT_REPO_ROOT
for v in "" _hl_a8_w6 _hl_a4_w8 _hl_a4_w6 _hl_a8_w8; do
  if [ -z "$v" ]; then unset HUF_LIB_PATH; else export HUF_LIB_PATH=$PWD/tools/_ablate/lib$v.so; fi
  python tools/time_encode_stages.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-100
done
