"""Seeded synthetic references and reads (SURVEY.md §8d "Synthetic stand-ins").

There is no network: neither GRCh38 nor the reference's Zenodo example data is
available, so tests and bench.py use these generators.  All generators are
deterministic functions of their seed.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for a, b in zip(b"ACGT", b"TGCA"):
    _COMP[a] = b


def revcomp(s: bytes) -> bytes:
    return _COMP[np.frombuffer(s, dtype=np.uint8)][::-1].tobytes()


def _mutate(rng, seg: np.ndarray, div: float) -> np.ndarray:
    seg = seg.copy()
    m = rng.random(seg.shape[0]) < div
    seg[m] = ACGT[rng.integers(0, 4, int(m.sum()))]
    return seg


def genome_small(seed: int = 1, n: int = 1_000_000) -> Tuple[np.ndarray, np.ndarray]:
    """S-small: uniform ACGT, one 50 kb segment duplicated once, 2 sequences."""
    rng = np.random.default_rng(seed)
    g = ACGT[rng.integers(0, 4, n)]
    dup = min(50_000, n // 8)
    src = n // 10
    dst = n // 2 + n // 7
    g[dst:dst + dup] = g[src:src + dup]
    starts = np.array([0, n // 2, n], dtype=np.uint32)
    return g, starts


def genome_mid(seed: int = 7, n: int = 16_000_000, n_rep: int = 200, rep_len: int = 5000,
               div: float = 0.01, n_seqs: int = 4) -> Tuple[np.ndarray, np.ndarray]:
    """S-mid: uniform genome + n_rep copies of repeats at `div` divergence."""
    rng = np.random.default_rng(seed)
    g = ACGT[rng.integers(0, 4, n)]
    rep_len = min(rep_len, max(16, n // (4 * max(n_rep, 1))))
    nfam = max(1, n_rep // 10)
    fams = [ACGT[rng.integers(0, 4, rep_len)] for _ in range(nfam)]
    for i in range(n_rep):
        pos = int(rng.integers(0, n - rep_len))
        g[pos:pos + rep_len] = _mutate(rng, fams[i % nfam], div)
    starts = np.linspace(0, n, n_seqs + 1).astype(np.uint32)
    return g, starts


def genome_rep(seed: int = 11, n: int = 16_000_000, scale: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """S-rep: repeat-rich genome (Alu-like 300 bp x 12000 @12 %, L1-like <=6 kb x 300 @5 %,
    2000 tandem repeats), scaled with n."""
    rng = np.random.default_rng(seed)
    g = ACGT[rng.integers(0, 4, n)]
    f = n / 16_000_000 * scale
    alu = ACGT[rng.integers(0, 4, 300)]
    for _ in range(int(12000 * f)):
        pos = int(rng.integers(0, n - 300))
        g[pos:pos + 300] = _mutate(rng, alu, 0.12)
    l1 = ACGT[rng.integers(0, 4, 6000)]
    for _ in range(int(300 * f)):
        ln = int(rng.integers(500, 6001))
        off = int(rng.integers(0, 6000 - ln + 1))
        pos = int(rng.integers(0, n - ln))
        g[pos:pos + ln] = _mutate(rng, l1[off:off + ln], 0.05)
    for _ in range(int(2000 * f)):
        unit = ACGT[rng.integers(0, 4, int(rng.integers(2, 30)))]
        copies = int(rng.integers(5, 40))
        tr = np.tile(unit, copies)
        pos = int(rng.integers(0, n - tr.shape[0]))
        g[pos:pos + tr.shape[0]] = tr
    starts = np.linspace(0, n, 5).astype(np.uint32)
    return g, starts


def sample_reads(genome: np.ndarray, n_reads: int, read_len: int, seed: int = 3,
                 edit_choices=(0, 0, 1, 1, 2, 3, 4), p_sub: float = 0.7, p_ins: float = 0.15,
                 rc_frac: float = 0.5, n_frac: float = 0.0) -> List[bytes]:
    """Reads sampled uniformly; edits drawn from `edit_choices` (70 % subst / 15 % ins /
    15 % del by default); `rc_frac` reverse-complemented; optional reads with an N."""
    rng = np.random.default_rng(seed)
    n = genome.shape[0]
    out = []
    for i in range(n_reads):
        pos = int(rng.integers(0, n - read_len - 8))
        r = list(genome[pos:pos + read_len + 8].tobytes())
        ne = int(edit_choices[int(rng.integers(0, len(edit_choices)))])
        for _ in range(ne):
            p = int(rng.integers(1, read_len - 1))
            u = rng.random()
            if u < p_sub:
                r[p] = int(ACGT[(int(np.searchsorted(ACGT, r[p])) + int(rng.integers(1, 4))) % 4])
            elif u < p_sub + p_ins:
                r.insert(p, int(ACGT[int(rng.integers(0, 4))]))
            else:
                del r[p]
        s = bytes(r[:read_len])
        if n_frac > 0 and rng.random() < n_frac:
            p = int(rng.integers(0, read_len))
            s = s[:p] + b"N" + s[p + 1:]
        if rng.random() < rc_frac:
            s = revcomp(s)
        out.append(s)
    return out
