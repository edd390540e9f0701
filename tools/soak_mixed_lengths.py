"""Soak run (GPU box, not part of the test suite): chunks that mix everything a Columba chunk may hold — reads of 1 ... 480
characters, reads not longer than the number of parts (naive backtracking on the device), the empty read, reads with N — on both
index backends, device vs oracle: occurrences and counters.
   python3 tools/soak_mixed_lengths.py [reads per configuration]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import columba_amd as ca
from columba_amd import indexbuild as ib, movebuild, synth
import oracle_py as op, schemes_py as sp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(17)
bad = 0


def chunk(g, n, k, parts, short_frac, seed, lens=(20, 36, 50, 100, 150, 151, 250, 256, 257, 320, 321, 400, 480)):
    r = np.random.default_rng(seed)
    reads = []
    for ln in lens:
        reads += synth.sample_reads(g, max(1, n // len(lens)), ln, seed=int(r.integers(1 << 30)), n_frac=0.02,
                                    edit_choices=(0, 1, 2, max(k - 1, 0), k, k + 1))
    for _ in range(int(n * short_frac)):
        L = int(r.integers(0, parts + 2))
        p = int(r.integers(0, len(g) - 10))
        reads.append(g[p:p + L].tobytes())
    order = r.permutation(len(reads))
    return [reads[i] for i in order]


def compare(tag, o, d, k, names):
    global bad
    (o_occ, o_off, o_cnt), (d_occ, d_off, d_cnt) = o, d
    same = np.array_equal(o_off, d_off) and all(np.array_equal(o_occ[f].astype(np.uint64), d_occ[f].astype(np.uint64))
                                                for f in ("begin", "end", "distance"))
    if k == 0 and np.array_equal(o_off, d_off):
        key = lambda occ, off: [sorted(map(tuple, np.stack([occ["begin"], occ["end"]], 1)[int(off[i]):int(off[i + 1])].tolist()))
                                for i in range(len(off) - 1)]
        same = key(o_occ, o_off) == key(d_occ, d_off)
    dup = o_cnt.get("SURVIVING_DUP_ROWS", 0)
    cn = [n for n in names if o_cnt[n] - (dup if n in ("TOTAL_REPORTED_POSITIONS", "LOCATED_ROWS") else 0) != d_cnt[n]]
    print(f"{tag}: {len(o_off) - 1} reads, {len(o_occ)} occurrences, occurrences {'identical' if same else 'DIFFER'}, "
          f"counters {'identical' if not cn else 'DIFFER ' + str(cn)}", flush=True)
    bad += (not same) + bool(cn)


# ---- FM-index: 300 kbp repeat-rich text (short reads match all over it)
g, starts = synth.genome_rep(seed=31, n=300_000, scale=1.5)
ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
FM = (("multiple_opt", "edit", "dynamic", 4, 5), ("columba", "edit", "dynamic", 7, 8), ("columba", "edit", "uniform", 1, 2),
      ("kuch1", "edit", "static", 2, 3), ("kuch2", "edit", "static", 3, 5), ("pigeon", "hamming", "dynamic", 3, 4),
      ("kianfar", "edit", "dynamic", 4, 5), ("minU", "hamming", "uniform", 5, 6), ("columba", "edit", "dynamic", 0, 1),
      ("naive", "edit", "dynamic", 2, 1), ("naive", "hamming", "dynamic", 3, 1), ("01*0", "edit", "static", 2, 4),
      # beyond 7 errors: the greedy schemes on the wide device tables, the in-text matrix with the wide left margin (no reads below 60
      # characters: at 10 errors those match all over the text)
      ("columba", "edit", "dynamic", 8, 10), ("columba", "edit", "uniform", 10, 12), ("columba", "edit", "static", 9, 11),
      ("columba", "hamming", "dynamic", 9, 11), ("columba", "hamming", "static", 13, 15),
      ("columba", "edit", "dynamic", 11, 13), ("columba", "edit", "static", 12, 14), ("columba", "edit", "uniform", 13, 15))
for spec, metric, part, k, P in FM:
    small = spec in ("kuch2", "01*0")
    dev, orc = ca.Index(ix, kmer_size=4 if small else 10), op.OracleIndex(ix, kmer_size=4 if small else 10)
    if spec == "naive":
        reads = [g[p:p + int(rng.integers(8, 26))].tobytes() for p in rng.integers(0, len(g) - 30, N // 6)] + [b"ACGTN", b""][:1 + (metric == "edit")]
    elif k >= 8:
        reads = chunk(g, max(N // 8, 100), k, P, 0.0, seed=int(rng.integers(1 << 30)), lens=(60, 100, 150, 151, 250, 257, 321, 400, 480) if k <= 10 else (100, 150, 151, 250, 257, 321, 400, 480))
    else:
        reads = chunk(g, N, k, P, 0.0 if k == 0 else 0.02, seed=int(rng.integers(1 << 30)))
    if True:
        if metric == "hamming":
            reads = [r for r in reads if len(r) > 0]
    t = time.time()
    o = op.match_batch(orc, op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=16)
    t1 = time.time()
    d = ca.match_batch(dev, ca.SearchStrategy(spec, metric, part), k, reads)
    names = ["NODE_COUNTER", "IN_TEXT_STARTED", "MATRIX_ROWS", "EXPANSIONS", "SEARCH_STARTED", "LOCATED_ROWS"]
    if k > 0:  # (k = 0: the reference subtracts the size of the whole vector, indexinterface.cpp:942; tests/test_gpu_parity.py)
        names += ["ABORTED_IN_TEXT_VERIF", "TOTAL_REPORTED_POSITIONS"]
    compare(f"FM-index {spec} {metric} {part} k={k} (oracle {t1 - t:.1f}s device {time.time() - t1:.2f}s)", o, d, k, names)
    dev.close()

# ---- b-move index: a small pan-genome
from numpy.random import default_rng
base = np.frombuffer(b"ACGT", np.uint8)[default_rng(3).integers(0, 4, 12_000)]
hap = []
for h in range(8):
    x = base.copy()
    m = default_rng(100 + h).random(len(x)) < 0.01
    x[m] = np.frombuffer(b"ACGT", np.uint8)[default_rng(200 + h).integers(0, 4, int(m.sum()))]
    hap.append(x)
pg = np.concatenate(hap + [np.frombuffer(b"ACGT", np.uint8)[default_rng(9).integers(0, 4, 20_000)]])
mv = movebuild.build_move(pg.tobytes(), device="cuda")
mdev, morc = ca.MoveIndex(mv), op.OracleMoveIndex(mv)
MV = (("multiple_opt", "edit", "dynamic", 4, 5), ("columba", "edit", "dynamic", 6, 7), ("kuch1", "edit", "uniform", 1, 2),
      ("pigeon", "hamming", "dynamic", 2, 3), ("columba", "hamming", "static", 4, 5), ("naive", "edit", "dynamic", 2, 1),
      ("kuch1", "edit", "dynamic", 0, 1),
      # beyond 7 errors: the wide tables, the wide record geometries of the frontier
      ("columba", "edit", "dynamic", 9, 11), ("columba", "edit", "static", 12, 14), ("columba", "hamming", "uniform", 13, 15))
for spec, metric, part, k, P in MV:
    if spec == "naive":
        reads = [pg[p:p + int(rng.integers(8, 22))].tobytes() for p in rng.integers(0, len(pg) - 30, N // 10)]
    elif k >= 8:   # (no reads below 100 characters: at these distances those match all over the text)
        reads = chunk(pg, max(N // 16, 100), k, P, 0.0, seed=int(rng.integers(1 << 30)), lens=(100, 150, 151, 250, 257, 321, 400, 480))
    else:
        reads = chunk(pg, N // 2, k, P, 0.0 if k == 0 else 0.01, seed=int(rng.integers(1 << 30)))
        if metric == "hamming":
            reads = [r for r in reads if len(r) > 0]
    t = time.time()
    o = morc.match_batch(op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=16, word_size=4)
    t1 = time.time()
    d = mdev.match_batch(ca.SearchStrategy(spec, metric, part), k, reads, kmer_size=4)
    names = ["NODE_COUNTER", "EXPANSIONS", "TOTAL_REPORTED_POSITIONS", "LOCATED_ROWS"] + (["SEARCH_STARTED", "MATRIX_ROWS"] if k > 0 and metric == "edit" else [])
    compare(f"b-move {spec} {metric} {part} k={k} (oracle {t1 - t:.1f}s device {time.time() - t1:.2f}s)", o, d, k, names)
print("soak:", "OK" if not bad else f"{bad} mismatches")
sys.exit(1 if bad else 0)
