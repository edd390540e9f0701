#!/usr/bin/env python3
"""The other configurations of BASELINE.json (parity-test cases, not bench lines) as the device runs them on the 3 Gbp
human-like index: reads/s of cmb_batch_run with the reads resident (second of two runs), occurrences, kernel groups.
usage: python tools/config_table.py [genome Mbp [only the configurations whose name contains this]]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import columba_amd as ca  # noqa: E402
from columba_amd import indexbuild as ib, synth  # noqa: E402

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3000
g, starts = synth.genome_human_like(int(mbp * 1e6), seed=2025, device="cuda")
ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
del g
torch.cuda.empty_cache()
dev = ca.Index(ix)
rows = []
for name, spec, metric, part, k, n_reads, length in (
        ("configs[0]-like: k = 0, pigeon, 100 bp", "pigeon", "edit", "uniform", 0, 1_000_000, 100),
        ("configs[1]: k = 2 Hamming, kuch_k+1, 150 bp", "kuch1", "hamming", "dynamic", 2, 1_000_000, 150),
        ("configs[2] (the bench line): k = 4 edit, multiple_opt, 150 bp", "multiple_opt", "edit", "dynamic", 4, 10_000_000, 150),
        ("configs[4]'s reads on the FM-index: k = 6 edit, multiple_opt, 250 bp", "multiple_opt", "edit", "dynamic", 6, 2_000_000, 250),
        ("k = 7 edit, columba strategy, 150 bp", "columba", "edit", "dynamic", 7, 1_000_000, 150),
        # new in round 3: the greedy schemes beyond 7 errors (wide device tables; in-text verification by k_wide_filter + k_verify_wide), long reads
        ("k = 9 edit, columba strategy (greedy scheme, wide in-text matrix), 150 bp", "columba", "edit", "dynamic", 9, 100_000, 150),
        ("k = 12 edit, columba strategy (greedy scheme, the matrices with narrow blocks), 150 bp", "columba", "edit", "dynamic", 12, 20_000, 150),
        ("k = 12 Hamming, columba strategy (greedy scheme), 150 bp", "columba", "hamming", "dynamic", 12, 200_000, 150),
        ("k = 4 edit, multiple_opt, 400 bp", "multiple_opt", "edit", "dynamic", 4, 2_000_000, 400)):
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    buf, offs = synth.sample_reads_fast(ix.text[:-1], n_reads, length, seed=3, device="cuda")
    torch.cuda.empty_cache()
    b = ca.Batch(dev, ca.SearchStrategy(spec, metric, part), k, packed=(buf, offs))
    for _ in range(2):
        t = time.time()
        b.run()
        dt = time.time() - t
    occ, _, cnt = b.results()
    rows.append({"config": name, "reads": n_reads, "ms": round(dt * 1e3, 1), "M_reads_per_s": round(n_reads / dt / 1e6, 2), "occurrences": int(len(occ)),
                 "nodes": int(cnt["NODE_COUNTER"]), "kernel_ms": {k_: round(v, 1) for k_, v in b.timings().items()}})
    print(json.dumps(rows[-1]), flush=True)
    b.close()
