cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stress_r04d
( python tests/stress/soak_handmade.py 420 4711; echo "rc=$?" ) > gpurun_out/stress_r04d/handmade.log 2>&1
( python tests/stress/soak.py 480 20261006; echo "rc=$?" ) > gpurun_out/stress_r04d/soak.log 2>&1
( python tests/stress/stress_decode.py 150; echo "rc=$?" ) > gpurun_out/stress_r04d/decode.log 2>&1
( python tests/stress/stress_deep_codes.py 200 4245; echo "rc=$?" ) > gpurun_out/stress_r04d/deep.log 2>&1
for f in gpurun_out/stress_r04d/*.log; do echo "== $f"; grep -v "amdgpu.ids\|^W2026" $f | tail -3 | cut -c1-400; done
