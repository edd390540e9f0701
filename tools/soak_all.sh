R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for t in soak_parity soak_mixed_lengths soak_move_search soak_move_tiny soak_outputs soak_boundaries soak_index_params soak_long_reads soak_tiny_texts soak_stress; do
  echo "=== $t.py"
  timeout -k 10 240 python3 tools/$t.py 2>&1 | grep -v "amdgpu.ids" | tail -6
  echo "exit: $?"
done
