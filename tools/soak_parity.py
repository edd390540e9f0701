"""Soak run (GPU box, not part of the test suite): device vs oracle on larger random inputs than tests/ use.
   python3 tools/soak_parity.py [reads per configuration]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
g, starts = synth.genome_rep(seed=77, n=6_000_000, scale=1.5)
ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
dev, orc = ca.Index(ix), op.OracleIndex(ix)
rng = np.random.default_rng(5)
bad = 0
CONFIGS = (("columba", "edit", "dynamic", 7), ("minU", "edit", "dynamic", 7), ("columba", "edit", "dynamic", 5),
           ("multiple_opt", "edit", "dynamic", 6), ("columba", "edit", "uniform", 7), ("minU", "edit", "static", 6),
           ("multiple_opt", "edit", "dynamic", 4), ("pigeon", "edit", "dynamic", 4), ("kuch1", "hamming", "dynamic", 4),
           ("columba", "edit", "dynamic", 1), ("columba", "edit", "dynamic", 2), ("columba", "edit", "dynamic", 3),
           ("kuch1", "edit", "static", 3), ("kianfar", "edit", "dynamic", 4), ("pigeon", "edit", "uniform", 3),
           ("multiple_opt", "edit", "static", 2), ("minU", "hamming", "dynamic", 7), ("columba", "hamming", "uniform", 5),
           ("multiple_opt", "edit", "dynamic", 0), ("columba", "edit", "static", 6), ("minU", "edit", "uniform", 4))
for spec, metric, part, k in CONFIGS:
    reads = []
    for ln in (40, 60, 100, 150, 151, 200, 256):
        reads += synth.sample_reads(g, N // 7, ln, seed=int(rng.integers(1 << 30)), n_frac=0.02,
                                    edit_choices=(0, 1, 2, 3, max(k - 1, 0), k, k, k + 1))
    t = time.time()
    o_occ, o_off, o_cnt = op.match_batch(orc, op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=64)
    t1 = time.time()
    d_occ, d_off, d_cnt = ca.match_batch(dev, ca.SearchStrategy(spec, metric, part), k, reads)
    t2 = time.time()
    same_off = np.array_equal(o_off, d_off)
    if k == 0 and same_off:
        # exact matches leave the reference in suffix-array order, the device returns them sorted (tests/test_gpu_parity.py)
        key = lambda occ, off: [sorted(map(tuple, occ[["begin", "end", "distance"]][int(off[i]):int(off[i + 1])].tolist()))
                                for i in range(len(off) - 1)]
        same = key(o_occ, o_off) == key(d_occ, d_off)
    else:
        same = same_off and all(np.array_equal(o_occ[f], d_occ[f]) for f in ("begin", "end", "distance"))
    names = ["NODE_COUNTER", "IN_TEXT_STARTED", "MATRIX_ROWS", "CIGARS_IN_TEXT_VERIFICATION", "EXPANSIONS", "SEARCH_STARTED"]
    if k > 0:
        names.append("ABORTED_IN_TEXT_VERIF")  # (k = 0: the reference subtracts the size of the whole vector, indexinterface.cpp:942)
    cn = [n for n in names if o_cnt[n] != d_cnt[n]]
    print(f"{spec} {metric} {part} k={k}: {len(reads)} reads, {len(o_occ)} occurrences, oracle {t1 - t:.1f}s device {t2 - t1:.2f}s, "
          f"occurrences {'identical' if same else 'DIFFER'}, counters {'identical' if not cn else 'DIFFER ' + str(cn)}", flush=True)
    bad += (not same) + bool(cn)
# second data set: human-like repeats (the benchmark's generator), the headline configuration and its neighbours at scale
import torch
g2, starts2 = synth.genome_human_like(40_000_000, seed=2025, device="cuda")
ix2 = ib.build_index(g2, seq_starts=starts2, device="cuda")
dev2, orc2 = ca.Index(ix2), op.OracleIndex(ix2)
g2h = g2.cpu().numpy() if hasattr(g2, "cpu") else g2
for spec, metric, part, k, n, ln in (("multiple_opt", "edit", "dynamic", 4, 200000, 150), ("columba", "edit", "dynamic", 7, 30000, 150),
                                     ("multiple_opt", "edit", "dynamic", 6, 60000, 250), ("columba", "edit", "dynamic", 4, 100000, 100),
                                     ("kuch1", "hamming", "dynamic", 3, 200000, 150), ("minU", "edit", "static", 5, 60000, 76),
                                     # beyond 7 errors: wide record geometries, the in-text matrices with wide left margins
                                     ("columba", "edit", "dynamic", 9, 10000, 150), ("columba", "edit", "dynamic", 11, 6000, 150),
                                     ("columba", "edit", "uniform", 12, 4000, 250), ("columba", "edit", "static", 13, 4000, 150),
                                     ("columba", "hamming", "dynamic", 13, 20000, 150)):
    reads = synth.sample_reads(g2h, n, ln, seed=int(rng.integers(1 << 30)), n_frac=0.01, edit_choices=(0, 0, 1, 2, 3, k, k + 1))
    t = time.time()
    o_occ, o_off, o_cnt = op.match_batch(orc2, op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=128)
    t1 = time.time()
    d_occ, d_off, d_cnt = ca.match_batch(dev2, ca.SearchStrategy(spec, metric, part), k, reads)
    t2 = time.time()
    same = np.array_equal(o_off, d_off) and all(np.array_equal(o_occ[f], d_occ[f]) for f in ("begin", "end", "distance"))
    cn = [c for c in ("NODE_COUNTER", "IN_TEXT_STARTED", "MATRIX_ROWS", "ABORTED_IN_TEXT_VERIF", "CIGARS_IN_TEXT_VERIFICATION",
                      "EXPANSIONS", "SEARCH_STARTED") if o_cnt[c] != d_cnt[c]]
    print(f"human-like 40 Mbp, {spec} {metric} {part} k={k}: {n} x {ln} bp, {len(o_occ)} occurrences, oracle {t1 - t:.1f}s device {t2 - t1:.2f}s, "
          f"occurrences {'identical' if same else 'DIFFER'}, counters {'identical' if not cn else 'DIFFER ' + str(cn)}", flush=True)
    bad += (not same) + bool(cn)
print("soak:", "OK" if not bad else f"{bad} mismatches")
sys.exit(1 if bad else 0)
