import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp
g2, starts2 = synth.genome_human_like(100_000_000, seed=2025, device="cuda")
ix2 = ib.build_index(g2, seq_starts=starts2, device="cuda")
dev2, orc2 = ca.Index(ix2), op.OracleIndex(ix2)
bad = 0
for spec, metric, part, k, n, ln in (("columba", "edit", "dynamic", 7, 400000, 150), ("minU", "edit", "dynamic", 7, 300000, 100),
                                     ("multiple_opt", "edit", "dynamic", 6, 400000, 150), ("columba", "edit", "dynamic", 5, 400000, 60),
                                     ("multiple_opt", "edit", "dynamic", 4, 1000000, 150)):
    buf, offs = synth.sample_reads_fast(g2, n, ln, seed=k + ln, device="cuda", edit_choices=(0, 1, 2, 3, k, k, k + 1))
    buf = np.asarray(buf); offs = np.asarray(offs)
    reads = [buf[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]
    t = time.time()
    o_occ, o_off, o_cnt = op.match_batch(orc2, op.OracleStrategy(sp.BY_NAME[spec], metric, part), k, reads, threads=200)
    t1 = time.time()
    d_occ, d_off, d_cnt = ca.match_batch(dev2, ca.SearchStrategy(spec, metric, part), k, reads)
    t2 = time.time()
    same = np.array_equal(o_off, d_off) and all(np.array_equal(o_occ[f], d_occ[f]) for f in ("begin", "end", "distance"))
    cn = [c for c in ("NODE_COUNTER", "IN_TEXT_STARTED", "MATRIX_ROWS", "ABORTED_IN_TEXT_VERIF", "CIGARS_IN_TEXT_VERIFICATION", "EXPANSIONS") if o_cnt[c] != d_cnt[c]]
    print(f"100 Mbp human-like, {spec} k={k}: {n} x {ln}, {len(o_occ)} occ, oracle {t1-t:.1f}s device {t2-t1:.2f}s: occurrences {'identical' if same else 'DIFFER'}, counters {'identical' if not cn else cn}", flush=True)
    bad += (not same) + bool(cn)
print("stress:", "OK" if not bad else f"{bad} mismatches")
