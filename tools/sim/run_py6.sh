cd $GRAFT_REPO_ROOT
for dev in 0 0,0 0,0,0; do
  for mb in 32 64 128 256; do
    echo -n "HUF_GPU_DEVICES=$dev round $mb MiB: "
    HUF_GPU_DEVICES=$dev HUF_GPU_BATCH_MB=$mb python -c "
import bench
r = bench.huffmanfile_layer(1 << 30, 1 << 20)
print(r['value'], 'compress', r['compress_GiBps'], 'decompress', r['decompress_GiBps'], 'drop', r['dropping_the_results_ms_per_pair'], r['bit_exact_roundtrip'])
" 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
echo -n "defaults: "
python -c "
import bench
r = bench.huffmanfile_layer(1 << 30, 1 << 20)
print(r['value'], 'compress', r['compress_GiBps'], 'decompress', r['decompress_GiBps'], 'drop', r['dropping_the_results_ms_per_pair'], r['bit_exact_roundtrip'])
" 2>&1 | grep -v amdgpu.ids | tail -1
