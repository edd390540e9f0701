"""Soak run, part 5 (GPU box): tiny references (50 ... 5000 characters, one or several sequences, homopolymers) — reads longer
than the text, windows clamped at both ends of the text, ranges covering the whole suffix array.   python3 tools/soak_tiny_texts.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp

rng = np.random.default_rng(3)
bad = 0
def text_of(kind, n):
    if kind == "random":
        return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tolist())
    if kind == "polyA":
        return b"A" * n
    if kind == "tandem":
        return (b"ACGTTGCA" * (n // 8 + 1))[:n]
    return (b"AC" * (n // 2 + 1))[:n]
for kind in ("random", "polyA", "tandem", "dinuc"):
    for n in (50, 200, 1000, 5000):
        t = text_of(kind, n)
        starts = np.array([0, n // 3, n], np.uint32) if n >= 200 else np.array([0, n], np.uint32)
        for sparse in (1, 4, 16):
            ix = ib.build_index(t, sparseness=sparse, seq_starts=starts, device="cuda")
            dev, orc = ca.Index(ix, kmer_size=4 if n < 1000 else 10), op.OracleIndex(ix, kmer_size=4 if n < 1000 else 10)
            g = np.frombuffer(t, np.uint8)
            for spec, metric, part, k in (("multiple_opt", "edit", "dynamic", 4), ("columba", "edit", "dynamic", 7), ("kuch1", "hamming", "dynamic", 2),
                                          ("pigeon", "edit", "uniform", 1), ("multiple_opt", "edit", "dynamic", 0)):
                reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), ln).tolist()) for ln in (36, 60, 150) for _ in range(20)]
                for ln in (36, 60, 150):
                    if n > ln + 8:
                        reads += synth.sample_reads(g, 60, ln, seed=int(rng.integers(1 << 30)), edit_choices=(0, 1, 2, k, k + 1))
                    else:   # reads longer than the text: built from its repetitions
                        reads += [(t * (ln // n + 2))[j:j + ln] for j in range(0, 40)]
                try:
                    o_occ, o_off, o_cnt = op.match_batch(orc, op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=16)
                    d_occ, d_off, d_cnt = ca.match_batch(dev, ca.SearchStrategy(spec, metric, part), k, reads)
                except Exception as e:
                    print(f"{kind} {n} sparse {sparse} {spec} k={k}: {type(e).__name__} {str(e)[:140]}", flush=True)
                    bad += 1
                    continue
                same = np.array_equal(o_off, d_off)
                if same:
                    key = lambda occ, off: [sorted(map(tuple, occ[["begin", "end", "distance"]][int(off[i]):int(off[i + 1])].tolist())) for i in range(len(off) - 1)]
                    same = key(o_occ, o_off) == key(d_occ, d_off) if k == 0 else all(np.array_equal(o_occ[f], d_occ[f]) for f in ("begin", "end", "distance"))
                names = ["NODE_COUNTER", "IN_TEXT_STARTED", "MATRIX_ROWS", "CIGARS_IN_TEXT_VERIFICATION", "EXPANSIONS"] + (["ABORTED_IN_TEXT_VERIF"] if k else [])
                cn = [c for c in names if o_cnt[c] != d_cnt[c]]
                if not same or cn:
                    bad += 1
                    print(f"{kind} {n} sparse {sparse} {spec} k={k}: occurrences {'identical' if same else 'DIFFER'} ({len(o_occ)} vs {len(d_occ)}), counters {cn}", flush=True)
    print(kind, "done", flush=True)
print("soak tiny texts:", "OK" if not bad else f"{bad} problems")
sys.exit(1 if bad else 0)
