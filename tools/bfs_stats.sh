#!/bin/bash
# What do the expansions of the frontier search produce?  Diagnostic build (-DCMB_BFS_STATS) on the GPU box:
#   tools/bfs_stats.sh [genome Mbp] [reads]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-variable -DCMB_BFS_STATS \
    -o /tmp/libcolumba_amd_stats.so columba_amd/csrc/columba_amd.hip columba_amd/csrc/move_backend.hip columba_amd/csrc/pair_sam.hip columba_amd/csrc/pair_best.hip
python3 - "$@" <<'PY'
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
import columba_amd as ca
ca.LIB_PATH = "/tmp/libcolumba_amd_stats.so"
from columba_amd import indexbuild as ib, synth
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
dev = torch.device("cuda", 0)
g, starts = synth.genome_human_like(int(mbp * 1e6), seed=2025, device=dev)
ix = ib.build_index(g, seq_starts=starts, device=dev, with_bwt=False)
del g
index = ca.Index(ix)
buf, _ = synth.sample_reads_fast(torch.from_numpy(ix.text[:-1]).to(dev), R, 150, seed=3, device=dev)
offs = np.arange(R + 1, dtype=np.uint64) * np.uint64(150)
for name, k in (("multiple_opt", 4), ("columba", 4), ("multiple_opt", 2)):
    b = ca.Batch(index, ca.SearchStrategy(name, "edit", "dynamic"), k, packed=(buf, offs))
    st = (C.c_ulonglong * 8)()
    ca.lib().cmb_debug_bfs_stats(st, 1)
    b.run()
    ca.lib().cmb_debug_bfs_stats(st, 1)
    occ, o, cnt = b.results()
    t = sum(st[:4]) or 1
    print(f"{name} k={k}: expansions {t} (DFS_EXPANSIONS {cnt['DFS_EXPANSIONS']}, NODE_COUNTER {cnt['NODE_COUNTER']}): "
          f"nothing {st[0]/t:.3f}, one plain node {st[1]/t:.3f} (same matrix block {st[4]/t:.3f}), "
          f"one final-column node {st[2]/t:.3f}, other {st[3]/t:.3f}", flush=True)
    b.close()
PY
