cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lean10
timeout 600 python tools/time_lean.py --mib 64 logtext zipf255 uniform256 uniform255 zipf255@16k zipf255@1m zipf255@4k logtext@1m 2>&1 | grep -v amdgpu.ids > gpurun_out/lean10/small.log
timeout 600 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 2>&1 | grep -v amdgpu.ids > gpurun_out/lean10/big.log
HUF_LIB_PATH=$PWD/tools/_ablate/lib_leanprof.so timeout 600 python tools/phase_lean.py zipf255 uniform256 2>&1 | grep -v amdgpu.ids > gpurun_out/lean10/phase.log
python tools/sim/dbg_lean2.py 2>&1 | grep -v "amdgpu.ids" | grep -v " ok$" > gpurun_out/lean10/dbg2.log
cat gpurun_out/lean10/*.log
