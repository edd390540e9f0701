# the soak tools that exercise the FM-index matcher (tools/soak_all.sh runs these and the b-move ones)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for t in soak_parity soak_outputs soak_boundaries soak_index_params soak_long_reads soak_tiny_texts soak_stress; do
  echo "=== $t.py"
  timeout -k 10 200 python3 tools/$t.py 2>&1 | grep -v "amdgpu.ids" | tail -4
  echo "exit: $?"
done
