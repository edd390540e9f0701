"""CPU soak of the 32-bit narrow-block experiment (no GPU): the oracle's edit-distance search with every part whose upper bound is at
most 7 on ONE 32-bit matrix of 8-row blocks (ORC_NARROW_BLOCKS=32: LEFT 15, DIAG 14, onlyVerticalGapsLeft answered as the reference's
64-bit matrix would; a phase whose first column does not fit falls back to the reference's matrix, as the device re-runs on GeoN) against
the run on the reference's matrices: occurrences and every counter.  This is what dev_bfs_edit.hpp's GeoN32 rests on.
   python3 tools/soak_narrow32.py [configurations [reads per configuration [max k]]]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp

L = C.CDLL(op.build())
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
max_k = int(sys.argv[3]) if len(sys.argv) > 3 else 7
rng = np.random.default_rng(78)
g, starts = synth.genome_rep(seed=43, n=600_000, scale=2.0)
ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cpu")
orc = {sw: op.OracleIndex(ix, switch_point=sw) for sw in (0, 4)}
bad = 0
rows = 0
tot = [0, 0]
for c in range(n_cfg):
    k = int(rng.integers(1, max_k + 1))
    length = int(rng.choice([30, 40, 60, 100, 150, 151, 250, 320, 400, 480]))
    if length < 5 * k:
        length = 150
    part = ["dynamic", "uniform", "static"][int(rng.integers(0, 3))]
    sw = int(rng.choice([0, 4]))
    spec = ["columba", "multiple_opt", "kuch1", "kuch2", "minU", "pigeon", "01*0", "kianfar"][int(rng.integers(0, 8))]
    if spec in ("kuch1", "kuch2", "kianfar", "multiple_opt") and k > 4:
        spec = "columba"
    if spec == "multiple_opt":
        st = op.OracleStrategy(sp.MULTIPLE_OPT, "edit", part)
    else:
        st = op.OracleStrategy(sp.BY_NAME[spec], "edit", part)
    reads = synth.sample_reads(g, n_reads, length, seed=int(rng.integers(1 << 30)), n_frac=0.01, edit_choices=(0, 1, k // 2, max(k - 1, 0), k, k, k + 1))
    os.environ.pop("ORC_NARROW_BLOCKS", None)
    t = time.time()
    try:
        a = op.match_batch(orc[sw], st, k, reads, threads=8)
    except Exception as e:  # (a strategy that does not exist at this k)
        print(f"{spec} {part} k={k}: skipped ({e})", flush=True)
        continue
    os.environ["ORC_NARROW_BLOCKS"] = "32"
    stats = (C.c_uint64 * 2)()
    L.orc_narrow32_stats(stats, 1)
    b = op.match_batch(orc[sw], st, k, reads, threads=8)
    L.orc_narrow32_stats(stats, 1)
    del os.environ["ORC_NARROW_BLOCKS"]
    same = np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    diff = [n for n in a[2] if a[2][n] != b[2][n]]
    rows += a[2]["MATRIX_ROWS"]
    tot[0] += stats[0]; tot[1] += stats[1]
    print(f"{spec} {part} k={k} {length} bp switch {sw}: {len(a[0])} occurrences, {a[2]['NODE_COUNTER']} nodes, {stats[0]} phases on 32 bits, {stats[1]} fell back, "
          f"{'identical' if same and not diff else 'DIFFER ' + str({n: (a[2][n], b[2][n]) for n in diff})} ({time.time() - t:.1f} s)", flush=True)
    bad += (not same) + bool(diff)
print(f"narrow 32-bit blocks: {'OK' if not bad else 'FAILED'} ({n_cfg} configurations, {tot[0]} phases on the 32-bit matrix, {tot[1]} fell back)")
sys.exit(1 if bad else 0)
