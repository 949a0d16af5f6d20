cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  echo -n "HUF_GPU_OVERLAP_TREE=$v  "
  HUF_GPU_OVERLAP_TREE=$v python tools/time_encode_stages.py zipf255 uniform256 const41 2>&1 | grep -v amdgpu.ids | cut -c1-200
done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-python-layer --no-live-traffic 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('bench', d['value'], d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
