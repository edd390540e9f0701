import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp
g, starts = synth.genome_rep(seed=77, n=6_000_000, scale=1.5)
ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
dev, orc = ca.Index(ix), op.OracleIndex(ix)
bad = 0
for spec, metric, part, k in (("multiple_opt", "edit", "dynamic", 4), ("columba", "edit", "dynamic", 7), ("columba", "edit", "dynamic", 1),
                              ("kuch1", "hamming", "dynamic", 3), ("columba", "edit", "uniform", 6), ("multiple_opt", "edit", "static", 2), ("multiple_opt", "edit", "dynamic", 0)):
    reads = []
    for ln in (257, 280, 300, 301, 320, 150, 36):
        reads += synth.sample_reads(g, 3000, ln, seed=ln + k, n_frac=0.02, edit_choices=(0, 1, 2, 3, max(k - 1, 0), k, k, k + 1))
    try:
        o_occ, o_off, o_cnt = op.match_batch(orc, op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=64)
        d_occ, d_off, d_cnt = ca.match_batch(dev, ca.SearchStrategy(spec, metric, part), k, reads)
        same = np.array_equal(o_off, d_off) and (k == 0 or all(np.array_equal(o_occ[f], d_occ[f]) for f in ("begin", "end", "distance")))
        cn = [n for n in ("NODE_COUNTER", "IN_TEXT_STARTED", "MATRIX_ROWS", "CIGARS_IN_TEXT_VERIFICATION", "EXPANSIONS") if o_cnt[n] != d_cnt[n]]
        print(spec, metric, part, k, len(o_occ), "identical" if same and not cn else f"DIFFER {cn}", flush=True)
        bad += (not same) + bool(cn)
    except Exception as e:
        print(spec, metric, part, k, "ERROR", str(e)[:200], flush=True)
        bad += 1
# alignments + best mode on long reads
reads = synth.sample_reads(g, 4000, 300, seed=5, n_frac=0.01, edit_choices=(0, 1, 3, 5, 7, 9))
tab = sp.BY_NAME["columba"]
o = op.match_best(orc, op.OracleStrategy(tab, "edit", "dynamic"), reads, x=0, min_identity=97, max_supported=13, threads=64)
d = ca.match_best(dev, ca.SearchStrategy("columba", "edit", "dynamic"), reads, x=0, min_identity=97)
ok = np.array_equal(o[5], d[4]) and np.array_equal(o[4], d[3]) and all(np.array_equal(o[0][f], d[0][f]) for f in ("begin", "end", "distance", "strand"))
cig = all(ca.cigar_string(d[2][int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])]) == o[3][j] for j, a in enumerate(d[1]))
print("best mode, 300 bp:", "identical" if ok and cig else "DIFFER", len(d[0]), flush=True)
bad += not (ok and cig)
try:
    ca.match_batch(dev, ca.SearchStrategy("multiple_opt"), 4, [b"ACGT" * 121])
    print("484-character read accepted?!"); bad += 1
except ca.CmbError as e:
    print("484 characters (the limit is 480):", str(e)[:80])
print("long reads:", "OK" if not bad else f"{bad} problems")
