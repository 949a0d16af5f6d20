cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stress_r04e
( python tests/stress/soak_handmade.py 300 99; echo "rc=$?" ) > gpurun_out/stress_r04e/handmade.log 2>&1
( python tests/stress/stress_encode.py 150 51; echo "rc=$?" ) > gpurun_out/stress_r04e/encode.log 2>&1
( python tests/stress/stress_fd.py 100 12; echo "rc=$?" ) > gpurun_out/stress_r04e/fd.log 2>&1
( python tests/stress/stress_offsets.py 1000; echo "rc=$?" ) > gpurun_out/stress_r04e/offsets.log 2>&1
( HUF_GPU_LEAN_DECODE=1 python tests/stress/soak.py 200 556; echo "rc=$?" ) > gpurun_out/stress_r04e/soak_lean.log 2>&1
for f in gpurun_out/stress_r04e/*.log; do echo "== $f"; grep -v "amdgpu.ids\|^W2026" $f | tail -2 | cut -c1-300; done
