cd $GRAFT_REPO_ROOT
for lanes in 0 2 4 6 8 12; do
  for mb in 32 128; do
    echo -n "lanes $lanes round $mb MiB: "
    HUF_GPU_COPY_LANES=$lanes HUF_GPU_BATCH_MB=$mb python -c "
import bench
r = bench.huffmanfile_layer(1 << 30, 1 << 20)
print(r['value'], 'compress', r['compress_GiBps'], 'decompress', r['decompress_GiBps'], 'drop', r['dropping_the_results_ms_per_pair'], r['bit_exact_roundtrip'])
" 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
