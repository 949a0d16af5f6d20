cd $GRAFT_REPO_ROOT
timeout 400 python tools/time_host_link.py 2>&1 | grep -v amdgpu.ids | grep "munmap\|first touch"
python -c "
import bench, json
print(json.dumps(bench.huffmanfile_layer(1 << 30, 1 << 20)))
print(json.dumps(bench.huffmanfile_layer(1 << 30, 1 << 20)))
" 2>&1 | grep -v amdgpu.ids | cut -c1-600
