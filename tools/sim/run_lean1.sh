cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lean1
timeout 600 python tools/time_lean.py --mib 64 zipf255 uniform256 uniform255 logtext zipf255@16k zipf255@1m > gpurun_out/lean1/small.log 2>&1
echo "rc=$?" >> gpurun_out/lean1/small.log
timeout 600 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 > gpurun_out/lean1/big.log 2>&1
echo "rc=$?" >> gpurun_out/lean1/big.log
HUF_GPU_LEAN_DECODE=0 timeout 600 python tools/time_lean.py --mib 1024 zipf255 uniform256 uniform255 > gpurun_out/lean1/big_old.log 2>&1
cat gpurun_out/lean1/*.log
