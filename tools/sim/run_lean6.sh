cd $GRAFT_REPO_ROOT
echo default; python tools/sim/dbg_lean2.py 2>&1 | grep -v amdgpu.ids
echo forced slow staging; HUF_LIB_PATH=$PWD/tools/_ablate/lib_leanslow.so python tools/sim/dbg_lean2.py 2>&1 | grep -v amdgpu.ids
