"""Harness-side index construction and index-file I/O (NOT on the hot path).

The hot path consumes Columba's Vanilla index arrays (SURVEY.md §5 "On-disk
index formats").  There is no network and no pre-built index on the GPU box, so
tests and bench.py build a synthetic index with the routines below.  They
produce *the reference's own array layouts* (so a real ``columba_build`` index
loads through the same code path, see ``load_index``):

* BWT / reverse BWT              — reference: src/buildindex.cpp:706-711, :575-585
* ``BitvecIntl<4>`` bits+counts  — reference: src/bitvec.h:247-281, :329-349,
                                    src/fmindex/bwtrepr.h:56-72
* sparse SA + rank9 ``Bitvec``   — reference: src/fmindex/suffixArray.h:150-164,
                                    src/bitvec.h:134-149
* 3-bit ``EncodedText`` BWT      — reference: src/fmindex/encodedtext.h:93-118
* file formats                   — reference: src/buildindex.cpp:341-386, :432,
                                    :591, :688; src/bitvec.h:176-195, :378-394

Suffix arrays are built by prefix doubling on torch tensors (CPU here, the GPU
on the bench box); the SA of a text is unique, so any correct builder yields
byte-identical index files (the reference uses libsais, buildindex.cpp:479).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"$ACGT"):
    CODE[_c] = _i


@dataclass
class IndexArrays:
    """Host copy of a Vanilla Columba index in the reference's array layouts."""

    text: np.ndarray            # uint8 ASCII, text[-1] == ord('$')
    counts: np.ndarray          # uint64[5] cumulative counts of $,A,C,G,T
    dollar_pos_fwd: int
    bv_fwd: np.ndarray          # uint64, BitvecIntl<4> bits
    cnt_fwd: np.ndarray         # uint64, BitvecIntl<4> counts
    dollar_pos_rev: int
    bv_rev: np.ndarray
    cnt_rev: np.ndarray
    bwt_words: np.ndarray       # uint64, EncodedText<5>
    sa_bv: np.ndarray           # uint64, Bitvec bits of the sampled rows
    sa_bv_counts: np.ndarray    # uint64, rank9 counts
    sa_samples: np.ndarray      # uint32 samples in SA-row order
    sparseness: int = 4
    seq_starts: np.ndarray = field(default_factory=lambda: np.zeros(1, np.uint32))
    seq_names: List[str] = field(default_factory=lambda: ["seq0"])

    @property
    def n(self) -> int:
        return int(self.text.shape[0])

    def nbytes(self) -> int:
        return sum(int(getattr(self, f).nbytes) for f in (
            "text", "bv_fwd", "cnt_fwd", "bv_rev", "cnt_rev", "bwt_words",
            "sa_bv", "sa_bv_counts", "sa_samples"))


# --------------------------------------------------------------------------
# suffix array by prefix doubling (torch; works on cpu and cuda)
# --------------------------------------------------------------------------
_SORT_LIM = (1 << 31) - 1024  # torch.sort refuses dimensions beyond INT_MAX
_SEL_CHUNK = 1 << 29


def _sort_big(key: torch.Tensor):
    """(sorted keys, permutation) like torch.sort, for any length: keys beyond torch.sort's INT_MAX limit are
    split at a sampled median into two independently sorted halves (equal keys stay on one side)."""
    n = int(key.numel())
    if n <= _SORT_LIM:
        return torch.sort(key)
    dev = key.device
    pivot = key[torch.randint(0, n, (1 << 20,), device=dev)].median()
    n_lo = 0
    for o in range(0, n, _SEL_CHUNK):
        n_lo += int((key[o:o + _SEL_CHUNK] < pivot).sum().item())
    if n_lo == 0 or n_lo == n:
        raise ValueError("degenerate keys: cannot split for sorting")
    k2 = torch.empty(n, dtype=key.dtype, device=dev)
    i2 = torch.empty(n, dtype=torch.int64, device=dev)
    a, b = 0, n_lo
    for o in range(0, n, _SEL_CHUNK):
        k = key[o:o + _SEL_CHUNK]
        m = k < pivot
        idx = torch.arange(o, o + int(k.numel()), dtype=torch.int64, device=dev)
        c = int(m.sum().item())
        k2[a:a + c] = k[m]
        i2[a:a + c] = idx[m]
        a += c
        c = int(k.numel()) - c
        m = ~m
        k2[b:b + c] = k[m]
        i2[b:b + c] = idx[m]
        b += c
    del key
    for lo, hi in ((0, n_lo), (n_lo, n)):
        sk, p = _sort_big(k2[lo:hi].clone())
        k2[lo:hi] = sk
        i2[lo:hi] = i2[lo:hi][p]
        del sk, p
    return k2, i2


def suffix_array(codes: torch.Tensor) -> torch.Tensor:
    """SA of the sequence ``codes`` (uint8, values 0..4), where a suffix that is a proper prefix of
    another sorts first (end-of-string smallest).  Prefix doubling with radix sorts (torch.sort).

    Returns an int64 tensor of length n on ``codes.device``.
    """
    dev = codes.device
    n = int(codes.numel())
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    # doubling key = (rank[i] - n/2) * (n+1) + rank[i+h]: centred so that texts up to 2^32 fit an int64
    half = n // 2
    assert (half + 2) * (n + 2) < 2 ** 63, "text too long for single-key prefix doubling"
    rdt = torch.int32 if n < 2 ** 31 - 2 else torch.int64
    h = 20  # initial key: first 20 symbols, 3 bits each (values 1..5, 0 = beyond the end)
    cp = torch.zeros(n + h, dtype=torch.uint8, device=dev)
    cp[:n] = codes.to(torch.uint8) + 1
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for j in range(h):
        key <<= 3
        key |= cp[j:j + n]
    del cp
    skey, sa = _sort_big(key)
    del key
    while True:
        flag = torch.ones(n, dtype=torch.bool, device=dev)
        flag[1:] = skey[1:] != skey[:-1]
        del skey
        srank = torch.cumsum(flag, 0, dtype=rdt)  # 1-based dense rank in SA order
        del flag
        if int(srank[-1].item()) == n:
            return sa
        rank = torch.empty(n, dtype=rdt, device=dev)
        rank[sa] = srank
        del srank, sa
        key = rank.to(torch.int64, copy=True)
        key -= half
        key *= (n + 1)
        if h < n:
            key[:n - h] += rank[h:]
        del rank
        skey, sa = _sort_big(key)
        del key
        h *= 2


_CHUNK = 1 << 27  # positions per chunk when packing bits (bounds temporary memory)


def _pack_bits_le(bits: torch.Tensor) -> torch.Tensor:
    """bits: bool tensor [W*64] -> int64 words (bit b of word w = bits[64w+b])."""
    sh = torch.arange(64, dtype=torch.int64, device=bits.device)
    out = []
    for o in range(0, bits.numel(), _CHUNK):
        w = bits[o:o + _CHUNK].reshape(-1, 64).to(torch.int64)
        out.append((w << sh).sum(dim=1))
    return torch.cat(out) if len(out) != 1 else out[0]


def _popcount_words(bits: torch.Tensor) -> torch.Tensor:
    out = []
    for o in range(0, bits.numel(), _CHUNK):
        out.append(bits[o:o + _CHUNK].reshape(-1, 64).sum(dim=1, dtype=torch.int64))
    return torch.cat(out) if len(out) != 1 else out[0]


def _rank_counts(pc: torch.Tensor, nblk: int, n_words: int):
    """L1 (absolute) and packed L2 (seven 9-bit partial sums) per 8-word block."""
    within = torch.cumsum(pc.reshape(nblk, 8), dim=1)
    blocktot = within[:, 7]
    l1 = torch.cumsum(blocktot, 0) - blocktot
    l2 = torch.zeros(nblk, dtype=torch.int64, device=pc.device)
    for j in range(1, 8):
        l2 |= within[:, j - 1] << (9 * (j - 1))
    # words past the end of the bitvector are never visited by index(): their L2 stay 0
    # (bitvec.h:335 / :138 loop over existing words only)
    last_words = n_words - (nblk - 1) * 8
    if last_words < 8:
        l2[-1] &= (1 << (9 * max(last_words - 1, 0))) - 1
    return l1, l2


def build_bitvec_intl(bwt_codes: torch.Tensor):
    """BWTRepresentation<5> arrays for a BWT given as codes 0..4 (0 = '$').

    Follows bwtrepr.h:56-72 (cumulative encoding, N = n+1 bits per symbol) and
    BitvecIntl<4>::index (bitvec.h:329-349).  Returns (bv, counts, dollarPos)
    as numpy uint64 arrays.
    """
    dev = bwt_codes.device
    n = int(bwt_codes.numel())
    N = n + 1
    nw = (N + 63) // 64
    nblk = (N + 511) // 512
    dollar_pos = n
    for o in range(0, n, _SEL_CHUNK):  # (chunked: nonzero / masked selects are limited to INT_MAX elements)
        z = torch.nonzero(bwt_codes[o:o + _SEL_CHUNK] == 0).flatten()
        if z.numel():
            dollar_pos = o + int(z[0].item())
            break
    codes = torch.zeros(nblk * 512, dtype=torch.uint8, device=dev)
    codes[:n] = bwt_codes.to(torch.uint8)
    bv = torch.zeros((nw, 4), dtype=torch.int64, device=dev)
    counts = torch.zeros((nblk, 4, 2), dtype=torch.int64, device=dev)
    for c in range(1, 5):
        bits = (codes != 0) & (codes <= c)
        bv[:, c - 1] = _pack_bits_le(bits)[:nw]
        l1, l2 = _rank_counts(_popcount_words(bits), nblk, nw)
        del bits
        counts[:, c - 1, 0] = l1
        counts[:, c - 1, 1] = l2
    bv_np = bv.reshape(-1).cpu().numpy().view(np.uint64)
    cnt_np = counts.reshape(-1).cpu().numpy().view(np.uint64)
    return bv_np, cnt_np, dollar_pos


def build_sparse_sa(sa: torch.Tensor, sparseness: int):
    """SparseSuffixArray (suffixArray.h:150-164) + Bitvec::index (bitvec.h:134-149)."""
    dev = sa.device
    n = int(sa.numel())
    mark = (sa % sparseness) == 0
    samples = np.concatenate([sa[o:o + _SEL_CHUNK][mark[o:o + _SEL_CHUNK]].to(torch.int64).cpu().numpy().astype(np.uint32)
                              for o in range(0, n, _SEL_CHUNK)]) if n else np.zeros(0, np.uint32)
    nw = (n + 63) // 64
    nblk = (nw + 7) // 8
    bits = torch.zeros(nblk * 512, dtype=torch.bool, device=dev)
    bits[:n] = mark
    del mark
    words = _pack_bits_le(bits)
    l1, l2 = _rank_counts(_popcount_words(bits), nblk, nw)
    cw = (nw + 7) // 4
    counts = np.zeros(cw, dtype=np.uint64)
    inter = torch.stack([l1, l2], dim=1).reshape(-1).cpu().numpy().view(np.uint64)
    counts[:min(cw, inter.shape[0])] = inter[:cw]
    return words[:nw].cpu().numpy().view(np.uint64), counts, samples


def encode_bwt(bwt_codes) -> np.ndarray:
    """EncodedText<5> (encodedtext.h:93-118): 3 bits per symbol, MSB first; 64 symbols = 3 words."""
    t = bwt_codes if isinstance(bwt_codes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(bwt_codes))
    dev = t.device
    n = int(t.numel())
    nw = (n * 3) // 64 + 1
    ngrp = (n + 63) // 64
    wsh = torch.arange(63, -1, -1, dtype=torch.int64, device=dev)
    out = []
    grp_chunk = _CHUNK // 64
    for g0 in range(0, ngrp, grp_chunk):
        g1 = min(ngrp, g0 + grp_chunk)
        sym = torch.zeros((g1 - g0) * 64, dtype=torch.uint8, device=dev)
        seg = t[g0 * 64:min(n, g1 * 64)].to(torch.uint8)
        sym[:seg.numel()] = seg
        bits = torch.stack([(sym >> 2) & 1, (sym >> 1) & 1, sym & 1], dim=1)  # MSB first
        words = (bits.reshape(-1, 3, 64).to(torch.int64) << wsh).sum(dim=2)
        out.append(words.reshape(-1))
    allw = torch.cat(out) if len(out) != 1 else out[0]
    res = np.zeros(nw, dtype=np.uint64)
    got = allw[:nw].cpu().numpy().view(np.uint64)
    res[:got.shape[0]] = got
    return res


def build_index(text, sparseness: int = 4, seq_starts: Optional[np.ndarray] = None,
                seq_names: Optional[List[str]] = None, device: str | torch.device = "cpu",
                with_bwt: bool = True) -> IndexArrays:
    """Build all Vanilla index arrays for ``text`` (ACGT only; '$' appended if absent).

    ``text``: bytes / numpy uint8 ASCII, or a torch uint8 tensor of ASCII codes (may live on the GPU).
    """
    dev = torch.device(device)
    if isinstance(text, torch.Tensor):
        tt = text.to(dev)
    else:
        t = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.asarray(text, dtype=np.uint8)
        tt = torch.from_numpy(np.array(t, dtype=np.uint8, copy=True)).to(dev)
    if tt.numel() == 0 or int(tt[-1].item()) != ord("$"):
        tt = torch.cat([tt, torch.tensor([ord("$")], dtype=torch.uint8, device=dev)])
    lut = torch.from_numpy(CODE).to(dev)
    tc = lut[tt.long()] if tt.numel() < (1 << 28) else torch.cat(
        [lut[tt[o:o + (1 << 28)].long()] for o in range(0, tt.numel(), 1 << 28)])
    if bool((tc[:-1] == 0).any()) or bool((tc == 255).any()):
        raise ValueError("text must consist of A,C,G,T followed by one final '$'")
    n = int(tt.numel())
    # forward
    sa = suffix_array(tc)
    prev = sa - 1
    prev = torch.where(prev < 0, n - 1, prev)
    bwt = tc[prev]
    del prev
    bv_fwd, cnt_fwd, dpos_f = build_bitvec_intl(bwt)
    sa_bv, sa_cnt, samples = build_sparse_sa(sa, sparseness)
    del sa
    bwt_words = encode_bwt(bwt) if with_bwt else np.zeros(1, np.uint64)
    del bwt
    # reverse text (buildindex.cpp:575-585, createRevSAWithSanityCheck :750)
    rtc = torch.flip(tc, dims=[0])
    rsa = suffix_array(rtc)
    del rtc
    idx = n - rsa
    idx = torch.where(rsa == 0, 0, idx)
    del rsa
    rbwt = tc[idx]
    del idx
    bv_rev, cnt_rev, dpos_r = build_bitvec_intl(rbwt)
    del rbwt
    cc = torch.bincount(tc.long() if n < (1 << 28) else tc[:1].long(), minlength=5) if n < (1 << 28) else None
    if cc is None:
        cc = torch.zeros(5, dtype=torch.int64, device=dev)
        for o in range(0, n, 1 << 28):
            cc += torch.bincount(tc[o:o + (1 << 28)].long(), minlength=5)
    cc = cc.cpu().numpy().astype(np.uint64)
    counts = np.zeros(5, dtype=np.uint64)
    counts[1:] = np.cumsum(cc)[:-1]
    if seq_starts is None:
        seq_starts = np.array([0, n - 1], dtype=np.uint32)
    if seq_names is None:
        seq_names = [f"seq{i}" for i in range(len(seq_starts) - 1)]
    return IndexArrays(text=tt.cpu().numpy(), counts=counts, dollar_pos_fwd=dpos_f, bv_fwd=bv_fwd,
                       cnt_fwd=cnt_fwd, dollar_pos_rev=dpos_r, bv_rev=bv_rev, cnt_rev=cnt_rev,
                       bwt_words=bwt_words, sa_bv=sa_bv, sa_bv_counts=sa_cnt, sa_samples=samples,
                       sparseness=sparseness, seq_starts=np.asarray(seq_starts, dtype=np.uint32),
                       seq_names=list(seq_names))


# --------------------------------------------------------------------------
# file I/O in the reference's on-disk formats
# --------------------------------------------------------------------------
def save_index(ix: IndexArrays, base: str) -> None:
    n = ix.n
    with open(base + ".meta", "w") as f:           # buildindex.cpp:688
        f.write("21\n4\nVANILLA\n")
    cct = np.zeros(256, dtype=np.uint32)           # buildindex.cpp:432
    for ch, cnt in zip(*np.unique(ix.text, return_counts=True)):
        cct[ch] = cnt
    cct.tofile(base + ".cct")
    with open(base + ".txt.bin", "wb") as f:       # buildindex.cpp:591
        f.write(struct.pack("<I", n))
        f.write(ix.text.tobytes())
    with open(base + ".bwt", "wb") as f:           # encodedtext.h:275
        f.write(struct.pack("<QQ", n, ix.bwt_words.shape[0]))
        f.write(ix.bwt_words.tobytes())
    for ext, dp, bv, cnt in ((".brt", ix.dollar_pos_fwd, ix.bv_fwd, ix.cnt_fwd),
                             (".rev.brt", ix.dollar_pos_rev, ix.bv_rev, ix.cnt_rev)):
        with open(base + ext, "wb") as f:          # bwtrepr.h:113 + bitvec.h:378
            f.write(struct.pack("<QQ", dp, n + 1))
            f.write(bv.tobytes())
            f.write(cnt.tobytes())
    s = ix.sparseness
    with open(f"{base}.sa.bv.{s}", "wb") as f:     # bitvec.h:176
        f.write(struct.pack("<Q", n))
        f.write(ix.sa_bv.tobytes())
        f.write(ix.sa_bv_counts.tobytes())
    ix.sa_samples.astype(np.uint32).tofile(f"{base}.sa.{s}")   # suffixArray.h:229
    ix.seq_starts.astype(np.uint32).tofile(base + ".pos")      # buildindex.cpp:341
    with open(base + ".sna", "wb") as f:
        for name in ix.seq_names:
            b = name.encode()
            f.write(struct.pack("<Q", len(b)))
            f.write(b)
    np.zeros(1, dtype=np.uint32).tofile(base + ".fsid")
    with open(base + ".headerSN.bin", "wb") as f:               # buildindex.cpp:341-353
        for i, name in enumerate(ix.seq_names):
            f.write(f"@SQ\tSN:{name}\tLN:{int(ix.seq_starts[i + 1]) - int(ix.seq_starts[i])}\n".encode())


def read_sparse_sa(base: str, sparseness: int):
    """``SparseSuffixArray(basename, sparseness)`` (suffixArray.h:176-207): the rank9 Bitvec of the sampled rows
    (bitvec.h:187: N, words, counts) and the raw 32-bit samples.  Returns (N, words, counts, samples)."""
    name = f"{base}.sa.bv.{sparseness}"
    if not os.path.exists(name):
        raise RuntimeError("Cannot open file: " + name + ". Did you set an incorrect suffix array sparseness "
                           "factor using the -s flag or move your index files?")
    with open(name, "rb") as f:
        N = struct.unpack("<Q", f.read(8))[0]
        nw = (N + 63) // 64
        sa_bv = np.frombuffer(f.read(nw * 8), dtype=np.uint64).copy()
        sa_cnt = np.frombuffer(f.read(((nw + 7) // 4) * 8), dtype=np.uint64).copy()
    if not os.path.exists(f"{base}.sa.{sparseness}"):
        raise RuntimeError(f"Problem reading file: {base}.sa.{sparseness}")
    samples = np.fromfile(f"{base}.sa.{sparseness}", dtype=np.uint32)
    return N, sa_bv, sa_cnt, samples


def load_index(base: str, sparseness: int = 4) -> IndexArrays:
    """Load a Vanilla index written by ``columba_build`` (32-bit length_t) or ``save_index``.

    Mirrors the loaders of the reference: indexinterface.cpp:77-204,
    fmindex.cpp:75-131, suffixArray.h:176, bitvec.h:187,388.
    """
    if os.path.exists(base + ".meta"):
        with open(base + ".meta") as f:
            toks = f.read().split()
        if len(toks) >= 2 and int(toks[1]) != 4:
            raise RuntimeError("The index was built with a compiled version that uses "
                               f"{int(toks[1]) * 8}-bit numbers; only 32-bit length_t is supported")
        if len(toks) >= 3 and toks[2] != "VANILLA":
            raise RuntimeError("The index was built with a different flavor of Columba")
    if not os.path.exists(base + ".cct"):
        raise RuntimeError("Cannot open file: " + base + ".cct")
    cct = np.fromfile(base + ".cct", dtype=np.uint32)
    nz = [int(c) for c in cct if c != 0]
    counts = np.zeros(5, dtype=np.uint64)
    counts[:len(nz)] = np.concatenate([[0], np.cumsum(nz)[:-1]])
    with open(base + ".txt.bin", "rb") as f:
        n = struct.unpack("<I", f.read(4))[0]
        text = np.frombuffer(f.read(n), dtype=np.uint8).copy()
    with open(base + ".bwt", "rb") as f:
        tsize, nwords = struct.unpack("<QQ", f.read(16))
        bwt_words = np.frombuffer(f.read(nwords * 8), dtype=np.uint64).copy()

    def rd_brt(path):
        with open(path, "rb") as f:
            dp, N = struct.unpack("<QQ", f.read(16))
            bvw = 4 * ((N + 63) // 64)
            cw = 8 * ((N + 511) // 512)
            bv = np.frombuffer(f.read(bvw * 8), dtype=np.uint64).copy()
            cnt = np.frombuffer(f.read(cw * 8), dtype=np.uint64).copy()
        return dp, bv, cnt

    dpf, bvf, cf = rd_brt(base + ".brt")
    dpr, bvr, cr = rd_brt(base + ".rev.brt")
    _, sa_bv, sa_cnt, samples = read_sparse_sa(base, sparseness)
    pos = np.fromfile(base + ".pos", dtype=np.uint32)
    names = []
    if os.path.exists(base + ".sna"):
        with open(base + ".sna", "rb") as f:
            data = f.read()
        o = 0
        while o + 8 <= len(data):
            ln = struct.unpack_from("<Q", data, o)[0]
            o += 8
            names.append(data[o:o + ln].decode(errors="replace"))
            o += ln
    return IndexArrays(text=text, counts=counts, dollar_pos_fwd=dpf, bv_fwd=bvf, cnt_fwd=cf,
                       dollar_pos_rev=dpr, bv_rev=bvr, cnt_rev=cr, bwt_words=bwt_words,
                       sa_bv=sa_bv, sa_bv_counts=sa_cnt, sa_samples=samples,
                       sparseness=sparseness, seq_starts=pos, seq_names=names)
