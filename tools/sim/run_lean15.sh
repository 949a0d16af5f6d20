cd $GRAFT_REPO_ROOT
timeout 600 python tools/time_lean.py --mib 64 logtext zipf255 uniform256 uniform255 zipf255@16k zipf255@1m zipf255@4k logtext@1m 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 2>&1 | grep -v amdgpu.ids
python tools/sim/dbg_lean2.py 2>&1 | grep -v "amdgpu.ids" | grep -v " ok$"
HUF_LIB_PATH=$PWD/tools/_ablate/lib_lean_tabonly.so timeout 300 python tools/time_lean.py --mib 1024 zipf255 2>&1 | grep -v amdgpu.ids | cut -c1-110 | head -1
