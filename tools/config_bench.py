"""Development helper: throughput of the other BASELINE.json configurations on the bench reference.
    python tools/config_bench.py <genome bp> <reads> <spec> <metric> <partition> <k> [...more 4-tuples]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
n, nreads = int(sys.argv[1]), int(sys.argv[2])
g, starts = synth.genome_human_like(n, seed=2025, device="cuda")
ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
del g
torch.cuda.empty_cache()
dev = ca.Index(ix)
buf, offs = synth.sample_reads_fast(ix.text[:-1], nreads, 150, seed=3, device="cuda")
torch.cuda.empty_cache()
a = sys.argv[3:]
for spec, metric, part, k in zip(a[0::4], a[1::4], a[2::4], a[3::4]):
    b = ca.Batch(dev, ca.SearchStrategy(spec, metric, part), int(k), packed=(buf, offs))
    for it in range(3):
        t = time.time(); b.run(); dt = time.time() - t
    occ, occ_offs, cnt = b.results()
    print(spec, metric, part, k, "reads/s", round(nreads / dt), "ms", round(dt * 1e3, 1), "occ", len(occ),
          {k_: round(v, 2) for k_, v in b.timings().items()}, flush=True)
    b.close()
