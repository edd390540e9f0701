"""Development helper: run the same batch repeatedly and check that the results never change (queue orders differ
from run to run; the occurrence lists and counters must not)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
n, nreads, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g, starts = synth.genome_human_like(n, seed=2025, device="cuda")
ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
del g
torch.cuda.empty_cache()
dev = ca.Index(ix)
buf, offs = synth.sample_reads_fast(ix.text[:-1], nreads, 150, seed=3, device="cuda")
torch.cuda.empty_cache()
for spec, metric, part, k in (("multiple_opt", "edit", "dynamic", 4), ("kuch1", "hamming", "uniform", 2)):
    b = ca.Batch(dev, ca.SearchStrategy(spec, metric, part), k, packed=(buf, offs))
    ref = None
    for it in range(iters):
        b.run()
        occ, occ_offs, cnt = b.results()
        h = hashlib.sha1(occ.tobytes() + occ_offs.tobytes()).hexdigest()
        key = (h, tuple(sorted(cnt.items())))
        if ref is None:
            ref = key
        assert key == ref, (spec, it, "results changed between runs")
    print(spec, metric, k, iters, "runs identical;", len(occ), "occurrences", flush=True)
    b.close()
