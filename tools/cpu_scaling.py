#!/usr/bin/env python3
"""Thread scaling of the CPU baseline (oracle/, the C++ restatement of the reference's path) on the host this runs on:
reads/s at 1 / 8 / 64 / all hardware threads on the bench workload (150 bp, k = 4 edit, multiple_opt, dynamic partitioning),
reads packed before the clock.   usage (GPU box): python tools/cpu_scaling.py [genome Mbp] > gpurun_out/cpu_scaling.txt
BASELINE.md §2 quotes the survey's probe of real Columba: 21 k reads/s per thread on a cache-resident 16 Mbp index."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from columba_amd import indexbuild as ib, synth  # noqa: E402
import oracle_py as op  # noqa: E402
import schemes_py as sp  # noqa: E402

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
n = int(mbp * 1e6)
dev = "cuda" if torch.cuda.is_available() else "cpu"
g, starts = synth.genome_human_like(n, seed=2025, device=dev)
ix = ib.build_index(g, seq_starts=starts, device=dev)
L = 150
buf, _ = synth.sample_reads_fast(torch.from_numpy(ix.text[:-1]).to(dev), 400_000, L, seed=3, device=dev)
oidx = op.OracleIndex(ix)
ost = op.OracleStrategy(sp.MULTIPLE_OPT, "edit", "dynamic")
cores = os.cpu_count() or 1
try:
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    share = "unlimited" if quota == "max" else f"{int(quota) / int(period):.1f} CPUs"
except Exception:
    share = "unknown"
print(f"cgroup CPU share of this job: {share}")
print(f"host: {cores} hardware threads; synthetic human-like reference {mbp:.0f} Mbp ({ix.nbytes() / 1e9:.2f} GB of index arrays), "
      f"{L} bp reads, k = 4 edit distance, multiple_opt, dynamic partitioning, ALL mode")
print("threads  reads  seconds  reads/s  reads/s per thread")
for threads, ns in ((1, 4_000), (8, 32_000), (16, 64_000), (64, 200_000), (cores, 400_000)):
    packed = (np.ascontiguousarray(buf[:ns * L]), np.arange(ns + 1, dtype=np.uint64) * np.uint64(L))
    t0 = time.perf_counter()
    occ, offs, cnt = op.match_batch(oidx, ost, 4, threads=threads, packed=packed)
    dt = time.perf_counter() - t0
    print(f"{threads:7d} {ns:6d} {dt:8.2f} {ns / dt:8.0f} {ns / dt / threads:8.0f}", flush=True)
