"""Soak run, part 2 (GPU box, not part of the test suite): alignments / SAM / BEST mode of the device against the oracle on a
larger reference and on configurations the test suite does not run (k = 7, mixed read lengths).  Reuses the checks of
tests/test_gpu_parity.py.   python3 tools/soak_outputs.py"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp
import test_gpu_parity as T

g, starts = synth.genome_rep(seed=77, n=6_000_000, scale=1.5)
ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
world = {"genome": g, "ix": ix, "dev": ca.Index(ix), "orc": op.OracleIndex(ix), "op": op}
oracle_dir = os.path.join(ROOT, "oracle")
bad = 0
def run(name, f, *a):
    global bad
    t = time.time()
    try:
        f(*a)
        print(f"{name}{a[1:] if f is not best_mixed else a}: ok ({time.time() - t:.1f} s)", flush=True)
    except Exception:
        bad += 1
        print(f"{name}{a}: FAILED\n{traceback.format_exc()[-1500:]}", flush=True)

def best_mixed(spec, metric, x, ident):
    rng = np.random.default_rng(11 + x)
    reads = []
    for ln in (40, 60, 100, 150, 151, 200, 256):
        reads += synth.sample_reads(g, 3000, ln, seed=int(rng.integers(1 << 30)), n_frac=0.01, edit_choices=(0, 0, 1, 2, 3, 5, 7, 9))
    st = np.asarray(ix.seq_starts, dtype=np.int64)
    for s in st[1:-1][:12]:
        reads += [g[int(s) - 75:int(s) + 75].tobytes(), g[int(s) - 3:int(s) + 147].tobytes(), g[int(s) - 147:int(s) + 3].tobytes()]
    tab = sp.BY_NAME[spec]
    ms = 0
    while (ms + 1) in tab["schemes"]:
        ms += 1
    ms = min(ms, 13)   # (MAX_K)
    o = op.match_best(world["orc"], op.OracleStrategy(tab, metric, "dynamic"), reads, x=x, min_identity=ident, max_supported=ms, threads=64)
    d = ca.match_best(world["dev"], ca.SearchStrategy(spec, metric, "dynamic"), reads, x=x, min_identity=ident)
    o_occ, o_sid, o_sb, o_cig, o_off, o_best, o_hits, o_cnt = o
    d_occ, d_aln, d_ops, d_off, d_best, d_hits, d_cnt = d
    assert np.array_equal(o_best, d_best) and np.array_equal(o_hits, d_hits) and np.array_equal(o_off, d_off)
    for f in ("begin", "end", "distance", "strand"):
        assert np.array_equal(o_occ[f], d_occ[f]), f
    assert np.array_equal(o_sid, d_aln["seq_id"]) and np.array_equal(o_sb, d_aln["seq_begin"])
    for j in range(len(d_occ)):
        a = d_aln[j]
        assert ca.cigar_string(d_ops[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])]) == o_cig[j], j

for spec, metric, k in (("columba", "edit", 7), ("minU", "edit", 5), ("multiple_opt", "edit", 6), ("columba", "edit", 3), ("pigeon", "edit", 4)):
    run("alignments", T.test_alignments_cigar_and_sequence, world, oracle_dir, spec, metric, k)
for spec, metric, k, xa in (("columba", "edit", 7, False), ("columba", "edit", 7, True), ("minU", "edit", 5, True), ("multiple_opt", "edit", 6, False)):
    run("sam", T.test_sam_records_of_a_chunk, world, spec, metric, k, xa)
for spec, metric, x, ident in (("columba", "edit", 0, 95), ("columba", "edit", 1, 95), ("columba", "edit", 2, 96), ("minU", "edit", 0, 95),
                               ("kuch1", "hamming", 0, 97), ("columba", "hamming", 1, 96)):
    run("best 150 bp", T.test_best_mode, world, spec, metric, x, ident)
    run("best mixed lengths", best_mixed, spec, metric, x, ident)
print("soak outputs:", "OK" if not bad else f"{bad} failures")
sys.exit(1 if bad else 0)
