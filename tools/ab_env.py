"""A/B of run-time knobs (environment variables read per batch run) on one index and one read set
(development helper).  usage: ab_env.py GENOME_BP READS "A=1 B=2" "A=2" ...   ("-" = no variable)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import columba_amd as ca
from columba_amd import indexbuild as ib, synth

n, nreads = int(sys.argv[1]), int(sys.argv[2])
settings = sys.argv[3:] or ["-"]
g, starts = synth.genome_human_like(n, seed=2025, device="cuda")
ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
del g
torch.cuda.empty_cache()
dev = ca.Index(ix)
buf, offs = synth.sample_reads_fast(ix.text[:-1], nreads, 150, seed=3, device="cuda")
torch.cuda.empty_cache()
st = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
ref = None
for rnd in range(2):
    for sset in settings:
        kv = [x.split("=") for x in sset.split()] if sset != "-" else []
        for k_, v in kv: os.environ[k_] = v
        b = ca.Batch(dev, st, 4, packed=(buf, offs))  # (some knobs are read when the batch is created)
        b.run()
        ts = []
        for it in range(3):
            t = time.time(); b.run(); ts.append(time.time() - t)
        occ, occ_offs, cnt = b.results()
        sig = (len(occ), int(occ["begin"].astype(np.uint64).sum()), cnt["MATRIX_ROWS"], cnt["NODE_COUNTER"])
        if ref is None:
            ref = sig
            d = np.diff(occ_offs.astype(np.int64))
            print("occurrences per read after the filter: max", int(d.max()), "reads with > 24:", int((d > 24).sum()),
                  "> 1000:", int((d > 1000).sum()), flush=True)
        print(f"{sset:40s} ms {[round(1000 * x, 1) for x in ts]} same={sig == ref}",
              {k_: round(v, 1) for k_, v in b.timings().items()}, flush=True)
        b.close()
        for k_, v in kv: del os.environ[k_]
