#!/bin/bash
# Diagnostic build (-DCMB_BOUNDS) on the GPU box: the data-dependent indices of the frontier kernels are checked instead of
# trusted; prints the first violation (site number of CMB_IDX in dev_bfs_edit.hpp, index, capacity).
#   tools/bounds_check.sh <strategy> <k> <reads> <genome bp> [read lengths...]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-variable -DCMB_BOUNDS \
    -Iinclude -o /tmp/libcolumba_amd_bounds.so columba_amd/csrc/columba_amd.hip columba_amd/csrc/move_backend.hip columba_amd/csrc/pair_sam.hip columba_amd/csrc/pair_best.hip
python3 - "$@" <<'PY'
import ctypes as C, sys, numpy as np
sys.path.insert(0, ".")
import columba_amd as ca
ca.LIB_PATH = "/tmp/libcolumba_amd_bounds.so"
from columba_amd import indexbuild as ib, synth
spec, k, n, gbp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(float(sys.argv[4]))
lens = [int(x) for x in sys.argv[5:]] or [40, 60, 100, 150, 151, 200, 256]
g, starts = synth.genome_rep(seed=77, n=gbp, scale=1.5)
ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
dev = ca.Index(ix)
rng = np.random.default_rng(5)
for ln in lens:
    reads = synth.sample_reads(g, n, ln, seed=int(rng.integers(1 << 30)), n_frac=0.02, edit_choices=(0, 1, 2, 3, k - 1, k, k, k + 1))
    try:
        occ, offs, cnt = ca.match_batch(dev, ca.SearchStrategy(spec, "edit", "dynamic"), k, reads)
        msg = f"{len(occ)} occurrences, NODE_COUNTER {cnt['NODE_COUNTER']}"
    except ca.CmbError as e:
        msg = "error: " + str(e)[:120]
    st = (C.c_ulonglong * 4)()
    ca.lib().cmb_debug_oob(st)
    print(f"length {ln}: {msg}; first violation: site {st[0]}, index {st[1]}, capacity {st[2]}", flush=True)
    if st[0]:
        break
PY
