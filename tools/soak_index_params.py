"""Soak run, part 3 (GPU box): index parameters the test suite keeps at their defaults — suffix-array sparseness, in-text
switch point, k-mer size of the seed table — device vs oracle.   python3 tools/soak_index_params.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp

g, starts = synth.genome_rep(seed=78, n=4_000_000, scale=1.5)
rng = np.random.default_rng(9)
bad = 0
for sparse in (1, 2, 4, 8, 16, 32):
    ix = ib.build_index(g.tobytes(), sparseness=sparse, seq_starts=starts, device="cuda")
    for sw, km in ((4, 10), (1, 10), (0, 10), (10, 10), (50, 8), (4, 4), (4, 12), (7, 6)) if sparse in (4, 16) else ((4, 10),):
        dev, orc = ca.Index(ix, in_text_switch=sw, kmer_size=km), op.OracleIndex(ix, switch_point=sw, kmer_size=km)
        for spec, metric, part, k in (("multiple_opt", "edit", "dynamic", 4), ("columba", "edit", "dynamic", 6), ("kuch1", "hamming", "dynamic", 3),
                                      ("pigeon", "edit", "uniform", 2)):
            reads = []
            for ln in (50, 100, 150, 250):
                reads += synth.sample_reads(g, 2500, ln, seed=int(rng.integers(1 << 30)), n_frac=0.02, edit_choices=(0, 1, 2, 3, k, k + 1))
            try:
                o_occ, o_off, o_cnt = op.match_batch(orc, op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=64)
                d_occ, d_off, d_cnt = ca.match_batch(dev, ca.SearchStrategy(spec, metric, part), k, reads)
            except Exception as e:
                # (k-mer tables larger than the seeds of a strategy allow are refused by both, with the same meaning)
                print(f"sparseness {sparse}, switch {sw}, k-mer {km}, {spec} {metric} {part} k={k}: {type(e).__name__} {str(e)[:90]}", flush=True)
                continue
            same = np.array_equal(o_off, d_off) and all(np.array_equal(o_occ[f], d_occ[f]) for f in ("begin", "end", "distance"))
            cn = [n for n in ("NODE_COUNTER", "IN_TEXT_STARTED", "MATRIX_ROWS", "ABORTED_IN_TEXT_VERIF", "CIGARS_IN_TEXT_VERIFICATION",
                              "EXPANSIONS", "SEARCH_STARTED", "IMMEDIATE_SWITCH") if o_cnt[n] != d_cnt[n]]
            if not same or cn:
                bad += 1
                print(f"sparseness {sparse}, switch {sw}, k-mer {km}, {spec} {metric} {part} k={k}: occurrences "
                      f"{'identical' if same else 'DIFFER'}, counters {'identical' if not cn else 'DIFFER ' + str(cn)}", flush=True)
    print(f"sparseness {sparse}: done", flush=True)
print("soak index parameters:", "OK" if not bad else f"{bad} mismatches")
sys.exit(1 if bad else 0)
