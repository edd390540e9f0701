"""Development helper: run one parity config and print per-read / counter differences vs the oracle."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np, torch
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
import oracle_py as op, schemes_py as sp
spec, metric, part, k = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
g, starts = synth.genome_rep(seed=11, n=2_000_000, scale=1.5)
ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
dev = ca.Index(ix); orc = op.OracleIndex(ix)
reads = synth.sample_reads(g, 3000, 150, seed=100 + k, n_frac=0.02)
only = [int(x) for x in sys.argv[5:]]
if only: reads = [reads[i] for i in only]
ost = op.OracleStrategy(sp.BY_NAME[spec], metric, part)
o_occ, o_off, o_cnt = op.match_batch(orc, ost, k, reads, threads=8)
d_occ, d_off, d_cnt = ca.match_batch(dev, ca.SearchStrategy(spec, metric, part), k, reads)
tup = lambda occ, off, i: [(int(o["begin"]), int(o["end"]), int(o["distance"])) for o in occ[int(off[i]):int(off[i+1])]]
bad = 0
for i in range(len(reads)):
    a, b = tup(o_occ, o_off, i), tup(d_occ, d_off, i)
    if a != b:
        bad += 1
        if bad <= 5:
            print("read", i, "oracle-only", sorted(set(a) - set(b))[:6], "gpu-only", sorted(set(b) - set(a))[:6], "n", len(a), len(b))
print("differing reads:", bad)
for n in o_cnt:
    if n in d_cnt and o_cnt[n] != d_cnt[n]: print("counter", n, o_cnt[n], d_cnt[n])
