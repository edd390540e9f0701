"""CPU soak of the narrow-block experiment (tests/test_narrow_block_matrix.py; no GPU): the oracle's edit-distance search with every part on
ONE 64-bit matrix of 16-row blocks (ORC_NARROW_BLOCKS=1, onlyVerticalGapsLeft answered as the part's own matrix would) against the run on
the reference's matrices (64-bit words up to an upper bound of 10, 128-bit words beyond): occurrences and every counter.
   python3 tools/soak_narrow_blocks.py [configurations [reads per configuration]]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp

op.build()
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rng = np.random.default_rng(77)
g, starts = synth.genome_rep(seed=43, n=600_000, scale=2.0)
ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cpu")
orc = {sw: op.OracleIndex(ix, switch_point=sw) for sw in (0, 4)}
bad = 0
rows = 0
for c in range(n_cfg):
    k = int(rng.integers(1, 14))
    length = int(rng.choice([40, 60, 100, 150, 151, 250, 320, 400, 480]))
    if length < 5 * k:
        length = 150
    part = ["dynamic", "uniform", "static"][int(rng.integers(0, 3))]
    sw = int(rng.choice([0, 4]))
    spec = "columba" if k > 4 or rng.random() < 0.5 else ["kuch1", "minU", "pigeon"][int(rng.integers(0, 3))]
    reads = synth.sample_reads(g, n_reads, length, seed=int(rng.integers(1 << 30)), n_frac=0.01, edit_choices=(0, 1, k // 2, max(k - 1, 0), k, k, k + 1))
    st = op.OracleStrategy(sp.BY_NAME[spec], "edit", part)
    os.environ.pop("ORC_NARROW_BLOCKS", None)
    t = time.time()
    a = op.match_batch(orc[sw], st, k, reads, threads=8)
    os.environ["ORC_NARROW_BLOCKS"] = "1"
    b = op.match_batch(orc[sw], st, k, reads, threads=8)
    del os.environ["ORC_NARROW_BLOCKS"]
    same = np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    diff = [n for n in a[2] if a[2][n] != b[2][n]]
    rows += a[2]["MATRIX_ROWS"]
    print(f"{spec} {part} k={k} {length} bp switch {sw}: {len(a[0])} occurrences, {a[2]['MATRIX_ROWS']} matrix rows, "
          f"{'identical' if same and not diff else 'DIFFER ' + str(diff)} ({time.time() - t:.1f} s)", flush=True)
    bad += (not same) + bool(diff)
print(f"narrow blocks: {'OK' if not bad else 'FAILED'} ({n_cfg} configurations, {rows} matrix rows)")
sys.exit(1 if bad else 0)
