"""Harness-side builder of the run-length compressed (b-move) index parts (test/bench tooling, not the hot path).

Produces, for a text over ACGT, what `columba_build` of the RUN_LENGTH_COMPRESSION flavour derives from the suffix
arrays of the text and of the reversed text (reference buildindex.cpp:1606-1686, 64-bit length_t, without PHI_MOVE):

* the move tables as the bytes of the reference's `.LFBP` / `.rev.LFBP` files (bmove/moverepr.cpp:145-181: three
  length_t header words, then nrOfRuns + 1 bit-packed rows) — the format `cmb_move_create` takes;
* the suffix array samples at run boundaries (buildindex.cpp:942-953), forward and reverse;
* the predecessor positions with their run mapping for phi / phi^-1 (buildindex.cpp:990-1013, :1044-1066) and the PLCP
  array (bmove/plcp.h:56-80, values; the reference stores them in sdsl sparse bit-vectors, whose file format is sdsl's
  and is not reproduced here).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from .indexbuild import CODE, suffix_array


@dataclass
class MoveArrays:
    n: int                      # text length including '$'
    lfbp_fwd: np.ndarray        # uint8: bytes of <base>.LFBP
    lfbp_rev: np.ndarray        # uint8: bytes of <base>.rev.LFBP
    smpf: np.ndarray            # uint64 samplesFirst
    smpl: np.ndarray            # uint64 samplesLast
    rev_smpf: np.ndarray
    rev_smpl: np.ndarray
    pred_first: np.ndarray      # uint64, sorted marked positions of predFirst
    first_to_run: np.ndarray    # uint64
    pred_last: np.ndarray
    last_to_run: np.ndarray
    plcp: np.ndarray            # uint32 PLCP values by text position
    sa: np.ndarray              # uint64 suffix array (kept for tests)
    rev_sa: np.ndarray
    text: np.ndarray            # uint8 ASCII with the final '$'

    @property
    def runs_fwd(self) -> int:
        return int(np.frombuffer(self.lfbp_fwd[8:16].tobytes(), dtype=np.uint64)[0])

    @property
    def runs_rev(self) -> int:
        return int(np.frombuffer(self.lfbp_rev[8:16].tobytes(), dtype=np.uint64)[0])


def _bits(v: int) -> int:
    return int(math.ceil(math.log2(v)))  # moverepr.h:44-46


def pack_lfbp(bwt: np.ndarray, cum: np.ndarray, length_bits: int = 64) -> np.ndarray:
    """The bytes of a .LFBP file for BWT codes `bwt` (0..4) — buildindex.cpp:826-915 + moverepr.cpp:145-181."""
    n = int(bwt.shape[0])
    starts = np.flatnonzero(np.concatenate([[True], bwt[1:] != bwt[:-1]])).astype(np.uint64)
    head = bwt[starts.astype(np.int64)].astype(np.uint64)
    r = int(starts.shape[0])
    # LF of a run start = C[c] + number of c's before it
    out = np.zeros(r, dtype=np.uint64)
    for c in range(5):
        m = bwt == c
        before = np.cumsum(m) - m  # occurrences of c before position i
        sel = head == c
        out[sel] = np.uint64(int(cum[c])) + before[starts[sel].astype(np.int64)].astype(np.uint64)
    out_run = (np.searchsorted(starts, out, side="right") - 1).astype(np.uint64)
    zero_pos = int(np.flatnonzero(bwt == 0)[0])
    bits_n, bits_r, bits_c = _bits(n), _bits(r), 3
    total_bits = bits_c + 2 * bits_n + bits_r
    total_bytes = (total_bits + 7) // 8
    mask_n = np.uint64((1 << bits_n) - 1)
    mask_r = np.uint64((1 << bits_r) - 1)
    # rows 0..r-1 and the terminating row (0, n, n, r), every value cut to its field like setRowValue does
    h = np.concatenate([head, [np.uint64(0)]])
    a = np.concatenate([starts, [np.uint64(n)]]) & mask_n
    b = np.concatenate([out, [np.uint64(n)]]) & mask_n
    d = np.concatenate([out_run, [np.uint64(r)]]) & mask_r
    lo = np.zeros(r + 1, dtype=np.uint64)
    hi = np.zeros(r + 1, dtype=np.uint64)

    def place(val, off):
        nonlocal lo, hi
        if off < 64:
            lo |= val << np.uint64(off)
            if off > 0:
                hi |= val >> np.uint64(64 - off)
        else:
            hi |= val << np.uint64(off - 64)

    place(h, 0)
    place(a, bits_c)
    place(b, bits_c + bits_n)
    place(d, bits_c + 2 * bits_n)
    rows = np.stack([lo, hi], axis=1).view(np.uint8).reshape(r + 1, 16)[:, :total_bytes]
    header = np.array([n, r, zero_pos], dtype=np.uint64 if length_bits == 64 else np.uint32).view(np.uint8)
    return np.concatenate([header, rows.reshape(-1)])


def _samples(sa: np.ndarray, bwt: np.ndarray):
    chg = np.flatnonzero(bwt[1:] != bwt[:-1])
    first = np.concatenate([[sa[0]], sa[chg + 1]]).astype(np.uint64)
    last = np.concatenate([sa[chg], [sa[-1]]]).astype(np.uint64)
    return first, last


def _plcp(tc: np.ndarray, sa: np.ndarray) -> np.ndarray:
    """PLCP values (Kasai; bmove/plcp.h:56-80) — vectorised by prefix doubling on ranks is overkill for the harness sizes:
    LCP of neighbouring suffixes by chunked comparison."""
    n = int(sa.shape[0])
    isa = np.empty(n, dtype=np.int64)
    isa[sa.astype(np.int64)] = np.arange(n)
    plcp = np.zeros(n, dtype=np.uint32)
    pad = np.concatenate([tc, np.array([254, 253], dtype=np.uint8)])  # beyond the text: never equal
    cur = sa[1:].astype(np.int64)
    prv = sa[:-1].astype(np.int64)
    l = np.zeros(n - 1, dtype=np.int64)
    active = np.arange(n - 1)
    while active.size:
        a = cur[active] + l[active]
        b = prv[active] + l[active]
        a = np.minimum(a, n)
        b = np.minimum(b, n + 1)
        eq = pad[a] == pad[b]
        l[active[eq]] += 1
        active = active[eq]
    plcp[cur] = l.astype(np.uint32)
    return plcp


def build_move(text, device: str | torch.device = "cpu", with_locate: bool = True) -> MoveArrays:
    t = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.asarray(text, dtype=np.uint8)
    if t.size == 0 or t[-1] != ord("$"):
        t = np.concatenate([t, np.array([ord("$")], dtype=np.uint8)])
    tc = CODE[t]
    if (tc[:-1] == 0).any() or (tc == 255).any():
        raise ValueError("text must consist of A,C,G,T followed by one final '$'")
    n = int(tc.shape[0])
    if n & (n - 1) == 0:
        raise ValueError("a text size that is a power of two cannot be packed (moverepr.cpp:75-77: the terminating row's "
                         "start position does not fit ceil(log2(n)) bits)")
    dev = torch.device(device)
    tt = torch.from_numpy(tc.copy()).to(dev)
    sa = suffix_array(tt).cpu().numpy().astype(np.int64)
    bwt = tc[np.where(sa > 0, sa - 1, n - 1)]
    rsa = suffix_array(torch.flip(tt, dims=[0])).cpu().numpy().astype(np.int64)
    rbwt = tc[np.where(rsa > 0, n - rsa, 0)]
    cnt = np.bincount(tc, minlength=5)
    cum = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.uint64)
    smpf, smpl = _samples(sa, bwt)
    rsmpf, rsmpl = _samples(rsa, rbwt)

    def pred(samples):  # buildindex.cpp:990-1013, :1044-1066 (stable order of equal keys cannot occur: keys are distinct)
        key = np.where(samples > 0, samples - np.uint64(1), np.uint64(n - 1))
        order = np.argsort(key, kind="stable")
        return key[order].astype(np.uint64), order.astype(np.uint64)

    pf, ftr = pred(smpf)
    pl, ltr = pred(smpl)
    return MoveArrays(n=n, lfbp_fwd=pack_lfbp(bwt, cum), lfbp_rev=pack_lfbp(rbwt, cum), smpf=smpf, smpl=smpl,
                      rev_smpf=rsmpf, rev_smpl=rsmpl, pred_first=pf, first_to_run=ftr, pred_last=pl, last_to_run=ltr,
                      plcp=_plcp(tc, sa) if with_locate else np.zeros(0, np.uint32), sa=sa.astype(np.uint64), rev_sa=rsa.astype(np.uint64), text=t)


def save_move(mv: MoveArrays, base: str) -> None:
    """Write the index parts as include/columba_amd_bmove.hpp (BMove) reads them: the two .LFBP files in the reference's
    format, everything the reference keeps in sdsl containers as plain little-endian 64-bit arrays."""
    from . import plcp_runs
    mv.lfbp_fwd.tofile(base + ".LFBP")
    mv.lfbp_rev.tofile(base + ".rev.LFBP")
    pos, sm = plcp_runs(mv.plcp)
    for ext, a in ((".smpf", mv.smpf), (".smpl", mv.smpl), (".rev.smpf", mv.rev_smpf), (".rev.smpl", mv.rev_smpl),
                   (".prdf", mv.pred_first), (".ftr", mv.first_to_run), (".prdl", mv.pred_last), (".ltr", mv.last_to_run),
                   (".plcp.pos", pos), (".plcp.sum", sm)):
        np.ascontiguousarray(a, dtype="<u8").tofile(base + ext + ".u64")


def plcp_gpu(mv: MoveArrays) -> np.ndarray:
    """PLCP values of a (repetitive) text on the GPU: PLCP[SA[i]] = LCP of the suffixes SA[i - 1] and SA[i], found by
    comparing one more character per round for the pairs still equal (the harness' stand-in for the reference's Kasai loop,
    bmove/plcp.h:56-80, which is what `build_move(with_locate=True)` runs on the CPU)."""
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


def pangenome(base_len: int, copies: int, snp: float, seed: int = 1) -> np.ndarray:
    """`copies` haplotypes of one random sequence of `base_len` characters, each with its own substitutions at rate `snp`
    (ASCII, no '$'): the repetitiveness (n / r of the BWT) of a collection of genomes of one species."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 4, base_len, dtype=np.uint8)
    parts = []
    for _ in range(copies):
        s = base.copy()
        m = rng.random(base_len) < snp
        s[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
        parts.append(s)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(parts)]
    if (text.shape[0] + 1) & text.shape[0] == 0:  # (a text size that is a power of two cannot be packed)
        text = text[:-1]
    return text
