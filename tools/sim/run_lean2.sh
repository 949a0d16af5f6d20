cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lean2
HUF_LIB_PATH=$PWD/tools/_ablate/lib_leanprof.so timeout 600 python tools/phase_lean.py zipf255 uniform256 > gpurun_out/lean2/phase.log 2>&1
cat gpurun_out/lean2/*.log
