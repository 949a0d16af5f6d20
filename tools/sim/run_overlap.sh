cd $GRAFT_REPO_ROOT
for v in ov8 ov4 ov2 ov_nohist ov_notree; do
  echo -n "$v overlap=1  "
  HUF_LIB_PATH=$PWD/tools/_ablate/lib_$v.so HUF_GPU_OVERLAP_TREE=1 python tools/time_encode_stages.py zipf255 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
echo -n "ov8 overlap=0  "
HUF_LIB_PATH=$PWD/tools/_ablate/lib_ov8.so HUF_GPU_OVERLAP_TREE=0 python tools/time_encode_stages.py zipf255 2>&1 | grep -v amdgpu.ids | cut -c1-200
