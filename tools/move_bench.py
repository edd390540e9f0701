#!/usr/bin/env python3
"""Microbenchmark + soak of the b-move extension kernel (first stage).

Builds the move tables of a pan-genome-like text (copies of one sequence with SNPs), collects ranges by a breadth-first walk
over the index, then
  * checks a sample of them against the oracle (all four children, all modes), and
  * times `k_move_extend` on the whole set (HIP events on the launch stream, cmb_move_extend_bench) next to the oracle's
    restatement of the reference's per-character walks on one host core.
usage: python tools/move_bench.py [--base-mbp 2] [--copies 32] [--snp 0.001] [--ranges 2000000]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import columba_amd as ca  # noqa: E402
from columba_amd import movebuild  # noqa: E402


def plcp_gpu(text, mv):
    """PLCP values of a (repetitive) text on the GPU: PLCP[SA[i]] = LCP of the suffixes SA[i - 1] and SA[i], found by doubling
    comparisons of 8-byte words (the harness' stand-in for the reference's Kasai loop, bmove/plcp.h:56-80)."""
    n = mv.n
    t = torch.from_numpy(np.concatenate([mv.text, np.zeros(16, np.uint8)])).cuda()
    sa = torch.from_numpy(mv.sa.astype(np.int64)).cuda()
    cur, prv = sa[1:], sa[:-1]
    lcp = torch.zeros(n - 1, dtype=torch.int64, device="cuda")
    active = torch.arange(n - 1, device="cuda")
    while active.numel():
        a = (cur[active] + lcp[active]).clamp_(max=n)
        b = (prv[active] + lcp[active]).clamp_(max=n + 1)
        eq = t[a] == t[b]
        eq &= (a < n) & (b < n)
        idx = active[eq]
        lcp[idx] += 1
        active = idx
    out = torch.zeros(n, dtype=torch.int64, device="cuda")
    out[cur] = lcp
    return out.cpu().numpy().astype(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base-mbp", type=float, default=2.0)
    ap.add_argument("--copies", type=int, default=32)
    ap.add_argument("--snp", type=float, default=0.001)
    ap.add_argument("--ranges", type=int, default=2_000_000)
    ap.add_argument("--check", type=int, default=100_000)
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads of the exact-matching leg (250 bp, BASELINE config 5's length)")
    args = ap.parse_args()
    rng = np.random.default_rng(1)
    base_len = int(args.base_mbp * 1e6)
    base = rng.integers(0, 4, base_len, dtype=np.uint8)
    parts = []
    for _ in range(args.copies):
        s = base.copy()
        m = rng.random(base_len) < args.snp
        s[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
        parts.append(s)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(parts)]
    if (text.shape[0] + 1) & text.shape[0] == 0:
        text = text[:-1]
    t0 = time.time()
    mv = movebuild.build_move(text, device="cuda", with_locate=False)
    if args.reads:  # locate arrays for the exact-matching leg: PLCP of a repetitive text through the suffix array on the GPU
        mv.plcp = plcp_gpu(text, mv)
    print(f"text {mv.n / 1e6:.1f} Mbp, {mv.runs_fwd} / {mv.runs_rev} runs (n/r = {mv.n / mv.runs_fwd:.1f}), built in {time.time() - t0:.0f} s", flush=True)
    dev = ca.MoveIndex(mv, with_locate=bool(args.reads))
    print(f"index in HBM: {dev.device_bytes() / 1e6:.1f} MB", flush=True)
    # ranges: a breadth-first walk that alternates direction, a slice of every level
    frontier = dev.complete_range()
    pool = []
    level = 0
    while sum(p.shape[0] for p in pool) < args.ranges and frontier.shape[0]:
        mode = (1, 0)[level % 2] if level % 3 else 1
        ch, ok = dev.extend(mode, frontier)
        nxt = ch.reshape(-1)[ok.reshape(-1) == 1]
        if nxt.shape[0] > 600_000:
            nxt = nxt[np.sort(rng.choice(nxt.shape[0], 600_000, replace=False))]
        frontier = nxt
        if level >= 8:
            pool.append(frontier)
        level += 1
    parents = np.concatenate(pool)[:args.ranges]
    w = parents["end"] - parents["begin"]
    nruns = parents["end_run"] - parents["begin_run"] + 1
    print(f"{parents.shape[0]} ranges from levels 9..{level}: width median {int(np.median(w))}, p90 {int(np.percentile(w, 90))}; "
          f"runs spanned median {int(np.median(nruns))}, p90 {int(np.percentile(nruns, 90))}", flush=True)
    # soak: device vs oracle
    import oracle_py as op
    orc = op.OracleMoveIndex(mv, with_locate=bool(args.reads))
    sample = parents[rng.choice(parents.shape[0], min(args.check, parents.shape[0]), replace=False)]
    fields = [f for f in ca.MOVE_RANGE_DTYPE.names if f != "reserved"]
    for mode in (0, 1, 2):
        d_ch, d_ok = dev.extend(mode, sample)
        for c in range(4):
            o_ch, o_ok, _ = orc.extend(mode, sample, np.full(sample.shape[0], c + 1, dtype=np.uint8))
            assert np.array_equal(d_ok[:, c], o_ok), (mode, c)
            for f in fields:
                assert np.array_equal(d_ch[:, c][f], o_ch[f]), (mode, c, f)
    print(f"soak: {sample.shape[0]} parents x 3 modes x 4 children identical to the oracle", flush=True)
    # device timing
    L = ca.lib()
    d_par = torch.from_numpy(parents.view(np.uint8).reshape(-1).copy()).cuda()
    d_ch = torch.empty(parents.shape[0] * 4 * 80, dtype=torch.uint8, device="cuda")
    d_ok = torch.empty(parents.shape[0] * 4, dtype=torch.uint8, device="cuda")
    res = {}
    for mode, name in ((1, "backward"), (0, "forward"), (2, "backward_uni")):
        ms = C.c_float()
        ca._chk(L.cmb_move_extend_bench(dev.h, mode, d_par.data_ptr(), parents.shape[0], d_ch.data_ptr(), d_ok.data_ptr(), args.iters, C.byref(ms)))
        res[name] = {"ms": round(ms.value, 3), "M_parents_per_s": round(parents.shape[0] / ms.value / 1e3, 1)}
    # the reference's walks on one core (oracle restatement): four extensions per parent
    ns = min(args.cpu_sample, parents.shape[0])
    cs = parents[:ns]
    t0 = time.time()
    steps = 0
    for c in range(4):
        _, _, st = orc.extend(1, cs, np.full(ns, c + 1, dtype=np.uint8))
        steps += st
    cpu_s = time.time() - t0
    row_bytes = (mv.lfbp_fwd.shape[0] - 24) // (mv.runs_fwd + 1)
    out = {"text_mbp": round(mv.n / 1e6, 1), "runs": mv.runs_fwd, "ranges": int(parents.shape[0]), "device": res,
           "io_bytes_per_parent": 80 + 4 * 80 + 4,
           "cpu_port_1core_M_parents_per_s": round(ns / cpu_s / 1e6, 3), "cpu_sample": ns,
           "reference_rows_stepped_per_parent": round(steps / ns, 2), "reference_row_bytes": int(row_bytes)}
    if args.reads:
        # exact matching end to end: 250 bp reads sampled from the text, half of them reverse-complemented, 10 % with one
        # substitution (no occurrence as a rule)
        comp = np.zeros(256, dtype=np.uint8)
        for a, b in zip(b"ACGT", b"TGCA"):
            comp[a] = b
        L = 250
        starts = rng.integers(0, mv.n - 1 - L, args.reads)
        reads = []
        for i, p0 in enumerate(starts):
            r = text[p0:p0 + L].copy()
            if i % 10 == 0:
                r[L // 2] = b"ACGT"[(b"ACGT".index(r[L // 2]) + 1) % 4]
            if i % 2:
                r = comp[r][::-1]
            reads.append(r.tobytes())
        t0 = time.time()
        occ, offs, cnt = dev.match_exact(reads)
        wall = time.time() - t0
        ms = dev.last_ms
        ns = min(2000, len(reads))
        t0 = time.time()
        o_occ, o_offs, o_cnt = orc.match_exact(reads[:ns])
        cpu_s = time.time() - t0
        assert np.array_equal(offs[:ns + 1], o_offs) and np.array_equal(occ["begin"][:int(o_offs[-1])], o_occ[:, 0])
        dev_ms = ms["extend"] + ms["scan"] + ms["locate"]
        out["exact_250bp"] = {"reads": len(reads), "occurrences": int(occ.shape[0]), "nodes": cnt["NODE_COUNTER"],
                              "device_ms": {k: round(v, 3) for k, v in ms.items()},
                              "M_reads_per_s_device": round(len(reads) / dev_ms / 1e3, 2),
                              "M_reads_per_s_with_transfers": round(len(reads) / wall / 1e6, 3),
                              "cpu_port_1core_reads_per_s": round(ns / cpu_s, 1), "cpu_sample": ns,
                              "checked_against_oracle": ns}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
