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


_CODE4 = np.zeros(256, dtype=np.int64)
for _i, _c in enumerate(b"ACGT"):
    _CODE4[_c] = _i


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


# --------------------------------------------------------------------------------------------
# large, human-like references and vectorised read sampling (bench.py)
# --------------------------------------------------------------------------------------------
def genome_human_like(n: int, seed: int = 2025, device="cpu"):
    """S-human-like: uniform background with injected repeat families so that the
    SA-range width distribution has a heavy tail like a mammalian genome:
      * Alu-like: 300 bp consensus, n/3000 copies (~10 % of the genome), 10-15 % divergence
      * L1-like: 6 kb consensus, truncated copies covering ~8 %, 5 % divergence
      * segmental duplications: 20 kb segments copied at 1 % divergence (~2 %)
      * tandem repeats: 2-30 bp units x 5-40 copies
    Returns (uint8 ASCII torch tensor on `device`, seq_starts numpy).  Deterministic in (`seed`, device type).
    """
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)

    def rnd(shape):
        return torch.randint(0, 4, shape, generator=gen, device=dev, dtype=torch.uint8)

    g = torch.empty(n, dtype=torch.uint8, device=dev)
    for o in range(0, n, 1 << 28):
        m = min(1 << 28, n - o)
        g[o:o + m] = rnd((m,))

    def put(idx, vals):
        # overlapping copies write some positions twice: on the GPU a plain indexed store lets either value win,
        # differently from run to run; the deterministic kernel (sorted indices) makes the genome reproducible
        prev = torch.are_deterministic_algorithms_enabled()
        torch.use_deterministic_algorithms(True)
        try:
            g.index_put_((idx,), vals)
        finally:
            torch.use_deterministic_algorithms(prev)

    def inject(consensus, copies, div, min_len=None):
        ln = consensus.numel()
        chunk = max(1, (1 << 26) // ln)
        for c0 in range(0, copies, chunk):
            c = min(chunk, copies - c0)
            pos = torch.randint(0, n - ln, (c,), generator=gen, device=dev)
            vals = consensus[None, :].repeat(c, 1)
            mut = torch.rand((c, ln), generator=gen, device=dev) < div
            vals = torch.where(mut, rnd((c, ln)), vals)
            idx = pos[:, None] + torch.arange(ln, device=dev)[None, :]
            if min_len is not None:  # truncated copies: keep a random suffix
                keep = torch.randint(min_len, ln + 1, (c,), generator=gen, device=dev)
                mask = torch.arange(ln, device=dev)[None, :] >= (ln - keep)[:, None]
                put(idx[mask], vals[mask])
            else:
                put(idx.reshape(-1), vals.reshape(-1))

    if n >= 100_000:
        inject(rnd((300,)), n // 3000, 0.12)
        inject(rnd((6000,)), max(1, n // 130_000), 0.05, min_len=500)
        nseg = max(1, n // 1_000_000)
        seg_len = min(20_000, n // 50)
        src = torch.randint(0, n - seg_len, (nseg,), generator=gen, device=dev)
        for s0 in src.tolist():
            inject(g[s0:s0 + seg_len].clone(), 1, 0.01)
        ntr = max(1, n // 8000)
        units = torch.randint(2, 31, (ntr,), generator=gen, device=dev)
        copies = torch.randint(5, 41, (ntr,), generator=gen, device=dev)
        pos = torch.randint(0, n - 1300, (ntr,), generator=gen, device=dev)
        maxl = 30 * 40
        ar = torch.arange(maxl, device=dev)[None, :]
        unit_seq = rnd((ntr, 30))
        tr = torch.gather(unit_seq, 1, (ar % units[:, None]).long())
        mask = ar < (units * copies)[:, None]
        idx = pos[:, None] + ar
        put(idx[mask], tr[mask])
    text = acgt[g.long()] if n < (1 << 28) else torch.cat([acgt[g[o:o + (1 << 28)].long()] for o in range(0, n, 1 << 28)])
    starts = np.linspace(0, n, 25).astype(np.uint32)  # 24 "chromosomes"
    return text, starts


def sample_reads_fast(genome, n_reads: int, read_len: int, seed: int = 3,
                      edit_choices=(0, 0, 1, 1, 2, 3, 4), p_sub: float = 0.7, p_ins: float = 0.15,
                      rc_frac: float = 0.5, device="cpu"):
    """Vectorised (torch) variant of `sample_reads`: same error model — edits drawn from
    `edit_choices`, 70 % substitutions / 15 % insertions / 15 % deletions, 50 % reverse-complemented.
    `genome`: uint8 ASCII numpy array or torch tensor.  Returns (buffer uint8 numpy
    [n_reads*read_len], offsets uint64[n_reads+1]).  Deterministic in (seed, device type)."""
    import torch
    dev = torch.device(device)
    g = genome if isinstance(genome, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(genome))
    g = g.to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    n = int(g.numel())
    W = read_len + 8
    L = read_len
    out_all = []
    chunk = 1 << 18
    choices = torch.tensor(list(edit_choices), device=dev)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    code = torch.zeros(256, dtype=torch.int64, device=dev)
    code[acgt.long()] = torch.arange(4, device=dev)
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=dev)
    comp[acgt.long()] = torch.tensor([84, 71, 67, 65], dtype=torch.uint8, device=dev)
    cols = torch.arange(L, device=dev)[None, :]
    for c0 in range(0, n_reads, chunk):
        m = min(chunk, n_reads - c0)
        pos = torch.randint(0, n - W - 1, (m,), generator=gen, device=dev)
        ne = choices[torch.randint(0, len(edit_choices), (m,), generator=gen, device=dev)]
        shift = torch.zeros((m, L), dtype=torch.int64, device=dev)
        ins_mask = torch.zeros((m, L), dtype=torch.bool, device=dev)
        sub_mask = torch.zeros((m, L), dtype=torch.bool, device=dev)
        for e in range(int(max(edit_choices))):
            act = ne > e
            p = torch.randint(1, L - 1, (m,), generator=gen, device=dev)
            u = torch.rand((m,), generator=gen, device=dev)
            is_sub = act & (u < p_sub)
            is_ins = act & (u >= p_sub) & (u < p_sub + p_ins)
            is_del = act & (u >= p_sub + p_ins)
            at = cols == p[:, None]
            sub_mask |= at & is_sub[:, None]
            ins_mask |= at & is_ins[:, None]
            shift += (is_del[:, None] & (cols >= p[:, None])).long()
            shift -= (is_ins[:, None] & (cols > p[:, None])).long()
        src = (cols + shift).clamp_(0, W - 1) + pos[:, None]
        out = g[src]
        subst = acgt[(code[out.long()] + torch.randint(1, 4, (m, L), generator=gen, device=dev)) % 4]
        out = torch.where(sub_mask, subst, out)
        out = torch.where(ins_mask, acgt[torch.randint(0, 4, (m, L), generator=gen, device=dev)], out)
        rc = torch.rand((m,), generator=gen, device=dev) < rc_frac
        out = torch.where(rc[:, None], torch.flip(comp[out.long()], dims=[1]), out)
        out_all.append(out.cpu())
    buf = torch.cat(out_all).reshape(-1).numpy() if out_all else np.zeros(0, np.uint8)
    offs = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    return np.ascontiguousarray(buf), offs
