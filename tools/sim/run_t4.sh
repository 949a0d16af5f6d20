cd $GRAFT_REPO_ROOT
for cfg in "2 8192 0" "2 8192 1" "4 8192 1" "2 60000 1" "2 1024 1" "2 100 1"; do
  timeout 120 python tools/sim/dbg_false2.py $cfg 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -3 | cut -c1-200
done
