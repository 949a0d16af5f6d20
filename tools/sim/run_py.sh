cd $GRAFT_REPO_ROOT
python -c "
import bench, json
print('standalone', json.dumps(bench.huffmanfile_layer(1 << 30, 1 << 20)))
print('again     ', json.dumps(bench.huffmanfile_layer(1 << 30, 1 << 20)))
" 2>&1 | grep -v amdgpu.ids | cut -c1-400
python -c "
import torch, bench, json
x = torch.empty(1 << 30, dtype=torch.uint8, device='cuda'); x.fill_(1); torch.cuda.synchronize()
print('with torch ', json.dumps(bench.huffmanfile_layer(1 << 30, 1 << 20)))
" 2>&1 | grep -v amdgpu.ids | cut -c1-400
