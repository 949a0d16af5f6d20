cd $GRAFT_REPO_ROOT
timeout 600 python tools/time_lean.py --mib 64 logtext zipf255@1m 2>&1 | grep -v amdgpu.ids
echo old; HUF_GPU_LEAN_DECODE=0 timeout 600 python tools/time_lean.py --mib 64 logtext zipf255@1m 2>&1 | grep -v amdgpu.ids
