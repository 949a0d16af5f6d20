cd $GRAFT_REPO_ROOT
for cfg in "2 8192 3000 0 0" "4 8192 3000 0 0" "4 8192 3000 1 0" "2 60000 50000 0 0" "4 60000 50000 1 0" "2 8192 3000 1 0"; do
  timeout 120 python tools/sim/dbg_false.py $cfg 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -3 | cut -c1-200
done
