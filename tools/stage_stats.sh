#!/bin/bash
# What do the rows of the verification stages do?  Diagnostic build (-DCMB_STAGE_STATS) on the GPU box:
#   tools/stage_stats.sh [genome Mbp] [reads]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-variable -DCMB_STAGE_STATS \
    -Iinclude -o /tmp/libcolumba_amd_sstats.so columba_amd/csrc/columba_amd.hip columba_amd/csrc/move_backend.hip columba_amd/csrc/pair_sam.hip columba_amd/csrc/pair_best.hip
python3 - "$@" <<'PY'
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
import columba_amd as ca
ca.LIB_PATH = "/tmp/libcolumba_amd_sstats.so"
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
b = ca.Batch(index, ca.SearchStrategy("multiple_opt", "edit", "dynamic"), 4, packed=(buf, offs))
st = (C.c_ulonglong * 24)()
ca.lib().cmb_debug_stage_stats(st, 1)
b.run()
ca.lib().cmb_debug_stage_stats(st, 1)
for i, name in enumerate(("first stage", "middle stages", "final-column stages")):
    w, l, am, lm, asl, ls, lb, ab = [st[8 * i + j] for j in range(8)]
    w = w or 1
    print(f"{name}: wave rows {w}, lanes/row {l / w:.1f}, rows with a RAC miss {am / w:.3f} ({lm / w:.1f} lanes), "
          f"rows with a miss beyond the first HP bit {asl / w:.3f} ({ls / w:.2f} lanes), "
          f"rows with a diagonal cell > maxED {ab / w:.3f} ({lb / w:.2f} lanes)", flush=True)
b.close()
PY
