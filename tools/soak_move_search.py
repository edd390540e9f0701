#!/usr/bin/env python3
"""Soak run of the b-move search against the oracle (GPU box; not part of the test suite): many configurations x thousands of reads
of mixed lengths on pan-genome-like texts of different repetitiveness, occurrences and counters compared.
usage: python tools/soak_move_search.py [reads per configuration]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import columba_amd as ca  # noqa: E402
from columba_amd import movebuild, synth  # noqa: E402
import oracle_py as op  # noqa: E402
import schemes_py as sp  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rng = np.random.default_rng(2)
texts = {"32 x 30 kb, 0.3 % SNPs": movebuild.pangenome(30_000, 32, 0.003, seed=5),
         "8 x 150 kb, 2 % SNPs + repeats": np.concatenate([movebuild.pangenome(150_000, 8, 0.02, seed=6), synth.genome_rep(seed=9, n=300_000, scale=3.0)[0]])}
configs = [("multiple_opt", "edit", "dynamic", 6), ("multiple_opt", "edit", "dynamic", 4), ("multiple_opt", "edit", "uniform", 2),
           ("columba", "edit", "dynamic", 7), ("columba", "edit", "dynamic", 5), ("columba", "edit", "static", 3), ("columba", "edit", "dynamic", 1),
           ("kuch1", "edit", "dynamic", 4), ("kuch1", "edit", "static", 2), ("minU", "edit", "dynamic", 6), ("pigeon", "edit", "dynamic", 3),
           ("kianfar", "edit", "dynamic", 2), ("kuch1", "hamming", "dynamic", 3), ("multiple_opt", "hamming", "dynamic", 6),
           ("columba", "hamming", "uniform", 4), ("kuch1", "edit", "dynamic", 0)]
bad = 0
for tname, g in texts.items():
    mv = movebuild.build_move(g.tobytes(), device="cuda")
    dev, orc = ca.MoveIndex(mv), op.OracleMoveIndex(mv)
    for ws in (8, 5):
        orc.prepare(ws)
        for spec, metric, part, k in configs:
            reads = []
            per = max(50, n_reads // 8 // (8 if (k >= 6 or spec == "kianfar") else 1))
            for ln in (40, 60, 75, 100, 125, 151, 200, 256):
                if ln <= 8 * 0 + (k + 2):
                    continue
                reads += synth.sample_reads(g, per, ln, seed=int(rng.integers(1 << 30)), n_frac=0.03,
                                            edit_choices=(0, 1, 2, max(k - 1, 0), k, k, k + 1))
            t0 = time.time()
            d_occ, d_off, d_cnt = dev.match_batch(ca.SearchStrategy(spec, metric, part), k, reads, kmer_size=ws)
            t1 = time.time()
            o_occ, o_off, o_cnt = orc.match_batch(op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=16, word_size=ws)
            same = np.array_equal(d_off, o_off) and all(np.array_equal(d_occ[f], o_occ[f].astype(d_occ[f].dtype)) for f in ("begin", "end", "distance"))
            names = ["NODE_COUNTER", "EXPANSIONS"] + (["SEARCH_STARTED", "MATRIX_ROWS"] if metric == "edit" and k else [])
            cnt_ok = all(d_cnt[n] == o_cnt[n] for n in names) and d_cnt["TOTAL_REPORTED_POSITIONS"] == o_cnt["TOTAL_REPORTED_POSITIONS"] - o_cnt["SURVIVING_DUP_ROWS"]
            bad += not (same and cnt_ok)
            print(f"{tname} | k-mer {ws} | {spec} {metric} {part} k={k}: {len(reads)} reads, {len(d_occ)} occurrences, device {t1 - t0:.2f} s, "
                  f"oracle {time.time() - t1:.2f} s: {'identical' if same and cnt_ok else 'DIFFERENT (occurrences %s, counters %s)' % (same, cnt_ok)}", flush=True)
print("soak:", "all identical" if bad == 0 else f"{bad} configurations differ")
sys.exit(1 if bad else 0)
