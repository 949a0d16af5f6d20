cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lean4
HUF_LIB_PATH=$PWD/tools/_ablate/lib_leanprof.so timeout 600 python tools/phase_lean.py zipf255 uniform256 uniform255 > gpurun_out/lean4/phase.log 2>&1
timeout 600 python tools/time_lean.py --mib 64 zipf255 uniform256 uniform255 logtext zipf255@16k zipf255@1m zipf255@4k logtext@1m > gpurun_out/lean4/small.log 2>&1
cat gpurun_out/lean4/*.log
