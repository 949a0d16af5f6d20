cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stress_r04
( python tests/stress/soak.py 600 20261005; echo "rc=$?" ) > gpurun_out/stress_r04/soak.log 2>&1
( HUF_GPU_LEAN_DECODE=1 python tests/stress/soak.py 420 555; echo "rc=$?" ) > gpurun_out/stress_r04/soak_lean.log 2>&1
( python tests/stress/stress_encode.py 200 41; echo "rc=$?" ) > gpurun_out/stress_r04/encode.log 2>&1
( python tests/stress/stress_decode.py 200; echo "rc=$?" ) > gpurun_out/stress_r04/decode.log 2>&1
( HUF_GPU_LEAN_DECODE=1 python tests/stress/stress_decode.py 120; echo "rc=$?" ) > gpurun_out/stress_r04/decode_lean.log 2>&1
( python tests/stress/stress_deep_codes.py 300 4244; echo "rc=$?" ) > gpurun_out/stress_r04/deep.log 2>&1
( python tests/stress/stress_fd.py 100 11; echo "rc=$?" ) > gpurun_out/stress_r04/fd.log 2>&1
( python tests/stress/stress_offsets.py 2000; echo "rc=$?" ) > gpurun_out/stress_r04/offsets.log 2>&1
for f in gpurun_out/stress_r04/*.log; do echo "== $f"; tail -3 $f; done
