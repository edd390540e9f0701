"""Dependent-extension latency: k = 0 exact matching of a few reads = a chain of ~L dependent extends per lane."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import columba_amd as ca
from columba_amd import indexbuild as ib, synth
for n in [int(float(x)) for x in sys.argv[1:]]:
    g, starts = synth.genome_human_like(n, seed=2025, device="cuda")
    ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
    del g
    dev = ca.Index(ix, in_text_switch=0)
    st = ca.SearchStrategy("kuch1", "edit", "dynamic")
    for nreads in (64, 4096, 262144, 1048576):
        buf, offs = synth.sample_reads_fast(ix.text[:-1], nreads, 150, seed=3, device="cuda", edit_choices=(0,), rc_frac=0.0)
        b = ca.Batch(dev, st, 0, packed=(buf, offs))
        b.run(); b.run()
        t = b.timings()["k_partition"]; occ, oo, cnt = b.results()
        print(f"n={n/1e6:.0f}Mbp reads={nreads}: k_partition {t:.3f} ms; ext {cnt['EXPANSIONS']}; "
              f"{t*1e3/150:.2f} us per dependent step (if all lanes run 150); {cnt['EXPANSIONS']/t/1e6:.2f} G ext/s")
        b.close()
    del dev, ix
    torch.cuda.empty_cache()
