#!/bin/bash
# What do the rounds of the walking frontier kernel (bfsExpandWalk) cost?  Diagnostic build (-DCMB_BFS_STATS) on the GPU box:
#   tools/walk_stats.sh [genome Mbp] [reads] [extra -D flags]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-variable -DCMB_BFS_STATS $3"
/opt/rocm/bin/hipcc $FL -c -o /tmp/ws_main.o columba_amd/csrc/columba_amd.hip &
/opt/rocm/bin/hipcc $FL -c -o /tmp/ws_move.o columba_amd/csrc/move_backend.hip &
/opt/rocm/bin/hipcc $FL -c -o /tmp/ws_pair.o columba_amd/csrc/pair_sam.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libcolumba_amd_stats.so /tmp/ws_main.o /tmp/ws_move.o /tmp/ws_pair.o
python3 - "$@" <<'PY'
import ctypes as C, sys, time, numpy as np, torch
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
for name, k in (("multiple_opt", 4),):
    b = ca.Batch(index, ca.SearchStrategy(name, "edit", "dynamic"), k, packed=(buf, offs))
    b.run()  # (sizes the pools)
    st = (C.c_ulonglong * 16)()
    ca.lib().cmb_debug_bfs_stats(st, 1)
    b.run()
    ca.lib().cmb_debug_bfs_stats(st, 1)
    occ, o, cnt = b.results()
    s = list(st)
    E = cnt["DFS_EXPANSIONS"]
    print(f"{name} k={k}: DFS_EXPANSIONS {E}; timings {b.timings()}")
    r = max(s[0], 1)
    print(f"  wave-rounds {s[0]}, expanding lanes per round {s[1] / r:.1f} of 64, children per round {s[7] / r:.1f}, child-loop turns per round {s[4] / r:.2f}, walking-on lanes per round {s[5] / r:.1f}")
    print(f"  nodes taken {s[6]}, expansions per node taken {E / max(s[6], 1):.2f}")
    print("  cycles per round %.0f: take+request %.0f | issue %.0f | wait %.0f | parents %.0f | children %.0f | parents again %.0f | loop end %.0f" %
          (s[3] / r, s[8] / r, s[9] / r, s[2] / r, s[10] / r, s[11] / r, s[12] / r, s[13] / r), flush=True)
    print("  raw per round/tile:", {j: round(s[j] / r, 1) for j in range(16)}, flush=True)
    b.close()
PY
