"""GPU parity of the b-move backend's first stage through the C-ABI: the move tables in HBM, character
extension with toeholds (all four children from one scan of the parent's runs) and locate, against the oracle's restatement
of bmove/moverepr.cpp (pinned to the reference) and bmove/bmove.cpp (checked by brute force in tests/test_move_oracle.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

pytestmark = pytest.mark.gpu

FIELDS = ["begin", "end", "begin_run", "end_run", "rev_begin", "rev_end", "rev_begin_run", "rev_end_run", "toehold",
          "original_depth", "runs_valid", "rev_runs_valid", "toehold_represents_end"]


def _pangenome(rng, base_len, copies, rate):
    base = rng.integers(0, 4, base_len)
    parts = []
    for _ in range(copies):
        s = base.copy()
        m = rng.random(base_len) < rate
        s[m] = rng.integers(0, 4, int(m.sum()))
        parts.append(s)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(parts)]


@pytest.fixture(scope="module")
def mworld(oracle_built):
    import columba_amd as ca
    from columba_amd import movebuild
    import oracle_py as op
    rng = np.random.default_rng(5)
    text = np.concatenate([_pangenome(rng, 20_000, 24, 0.004), np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 30_000)],
                           np.frombuffer(b"A" * 300 + b"AC" * 200 + b"G", np.uint8)])
    mv = movebuild.build_move(text.tobytes(), device="cuda")
    return {"mv": mv, "dev": ca.MoveIndex(mv), "orc": op.OracleMoveIndex(mv), "ca": ca, "op": op, "rng": rng}


def _same(a, b, what):
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), (what, f, np.flatnonzero(a[f] != b[f])[:5])


def test_tables_in_hbm_are_the_file_rows(mworld):
    dev, orc, mv = mworld["dev"], mworld["orc"], mworld["mv"]
    assert (dev.n, dev.runs, dev.rev_runs) == (mv.n, mv.runs_fwd, mv.runs_rev)
    assert mv.runs_fwd * 8 < mv.n  # a repetitive text: long runs
    for rev in (0, 1):
        assert np.array_equal(dev.rows(rev), orc.rows(rev))
    a, b = dev.complete_range(), orc.complete_range()
    _same(a, b, "complete range")
    assert dev.device_bytes() >= 16 * (mv.runs_fwd + mv.runs_rev)


def _oracle_children(orc, mode, parents):
    n = parents.shape[0]
    ch = np.zeros((n, 4), dtype=parents.dtype)
    ok = np.zeros((n, 4), dtype=np.uint8)
    for c in range(4):
        ch[:, c], ok[:, c], _ = orc.extend(mode, parents, np.full(n, c + 1, dtype=np.uint8))
    return ch, ok


def test_extension_all_modes_against_the_oracle(mworld):
    """a breadth-first walk over the index that switches direction from level to level (so that ranges with stale run
    indices are extended too), every child compared field by field"""
    dev, orc, rng = mworld["dev"], mworld["orc"], mworld["rng"]
    stats = {"children": 0, "empty": 0, "narrower": 0, "stale_parents": 0}
    for modes in ((1, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 1), (0, 0, 1, 0, 1, 1, 0, 1, 0, 0, 1, 0), (2,) * 12, (1,) * 14, (0,) * 14):
        frontier = dev.complete_range()
        for level, mode in enumerate(modes):
            d_ch, d_ok = dev.extend(mode, frontier)
            o_ch, o_ok = _oracle_children(orc, mode, frontier)
            assert np.array_equal(d_ok, o_ok), (modes, level)
            _same(d_ch.reshape(-1), o_ch.reshape(-1), (modes, level))
            stats["children"] += int(d_ok.sum())
            stats["empty"] += int((d_ok == 0).sum())
            tr = frontier["rev_runs_valid" if mode == 0 else "runs_valid"] == 0
            stats["stale_parents"] += int(tr.sum())
            w_parent = (frontier["end"] - frontier["begin"])[:, None]
            stats["narrower"] += int(((d_ch["end"] - d_ch["begin"] < w_parent) & (d_ok == 1)).sum())
            nxt = d_ch.reshape(-1)[d_ok.reshape(-1) == 1]
            if nxt.shape[0] > 6000:
                nxt = nxt[np.sort(rng.choice(nxt.shape[0], 6000, replace=False))]
            frontier = nxt
            if frontier.shape[0] == 0:
                break
    assert stats["children"] > 50_000 and stats["empty"] > 5_000 and stats["narrower"] > 20_000 and stats["stale_parents"] > 2_000, stats


def test_locate_against_the_oracle(mworld):
    dev, orc, mv, rng = mworld["dev"], mworld["orc"], mworld["mv"], mworld["rng"]
    ranges = []
    frontier = dev.complete_range()
    for level, mode in enumerate((1, 0, 1, 1, 0, 1, 0, 0, 1, 1, 0, 1, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0)):
        ch, ok = dev.extend(mode, frontier)
        nxt = ch.reshape(-1)[ok.reshape(-1) == 1]
        if nxt.shape[0] > 3000:
            nxt = nxt[np.sort(rng.choice(nxt.shape[0], 3000, replace=False))]
        frontier = nxt
        w = frontier["end"] - frontier["begin"]
        pick = frontier[w <= 3000]
        if pick.shape[0] > 400:
            pick = pick[rng.choice(pick.shape[0], 400, replace=False)]
        ranges.append(pick)
    ranges = np.concatenate(ranges)
    assert ranges.shape[0] > 3000
    pos, offs = dev.locate(ranges)
    multi = 0
    for i in range(0, ranges.shape[0], 7):
        want = orc.locate(ranges[i:i + 1])
        got = pos[int(offs[i]):int(offs[i + 1])]
        assert np.array_equal(got, want), i
        assert np.array_equal(np.sort(got), np.sort(mv.sa[int(ranges["begin"][i]):int(ranges["end"][i])]))
        multi += got.shape[0] > 1
    assert multi > 100
    # all of them against the suffix array
    for i in range(ranges.shape[0]):
        b, e = int(ranges["begin"][i]), int(ranges["end"][i])
        assert np.array_equal(np.sort(pos[int(offs[i]):int(offs[i + 1])]), np.sort(mv.sa[b:e])), i


def test_files_of_the_32_bit_build_load(mworld):
    ca, mv = mworld["ca"], mworld["mv"]
    from columba_amd import movebuild
    import copy
    m32 = copy.copy(mv)
    for name in ("lfbp_fwd", "lfbp_rev"):
        f = getattr(mv, name)
        hdr = np.frombuffer(f[:24].tobytes(), dtype=np.uint64).astype(np.uint32).view(np.uint8)
        setattr(m32, name, np.concatenate([hdr, f[24:]]))
    d32 = ca.MoveIndex(m32, with_locate=False, length_bits=32)
    for rev in (0, 1):
        assert np.array_equal(d32.rows(rev), mworld["dev"].rows(rev))
    d32.close()


def test_malformed_tables_and_ranges_are_refused(mworld):
    ca, mv, dev = mworld["ca"], mworld["mv"], mworld["dev"]
    import copy
    # a row whose start position breaks the order of the runs
    bad = copy.copy(mv)
    f = mv.lfbp_fwd.copy()
    row_bytes = (f.shape[0] - 24) // (mv.runs_fwd + 1)
    f[24 + 5 * row_bytes: 24 + 6 * row_bytes] = f[24 + 3 * row_bytes: 24 + 4 * row_bytes]
    bad.lfbp_fwd = f
    with pytest.raises(ca.CmbError) as e:
        ca.MoveIndex(bad)
    assert e.value.code == ca.CMB_ERR_INVALID and "inconsistent move table" in str(e.value)
    # truncated file
    bad = copy.copy(mv)
    bad.lfbp_rev = mv.lfbp_rev[:-40]
    with pytest.raises(ca.CmbError) as e:
        ca.MoveIndex(bad)
    assert e.value.code == ca.CMB_ERR_INVALID and "truncated" in str(e.value)
    # locate positions out of order
    bad = copy.copy(mv)
    bad.pred_first = mv.pred_first[::-1].copy()
    with pytest.raises(ca.CmbError) as e:
        ca.MoveIndex(bad)
    assert e.value.code == ca.CMB_ERR_INVALID and "locate" in str(e.value)
    # parents that are not ranges of the index
    cr = dev.complete_range()
    for field, val in (("end", mv.n + 5), ("end_run", mv.runs_fwd), ("begin", mv.n)):
        p = cr.copy()
        p[field] = val
        with pytest.raises(ca.CmbError) as e:
            dev.extend(1, p)
        assert e.value.code == ca.CMB_ERR_INVALID
    ch, ok = dev.extend(1, cr)
    p = ch[0, 1:2].copy()
    p["begin_run"] += 1  # "valid" run indices that do not hold the range's ends
    with pytest.raises(ca.CmbError):
        dev.extend(1, p)
    # a range whose toehold / depth do not describe its interval
    p = ch[0, 1:2].copy()
    p["original_depth"] = 40
    with pytest.raises(ca.CmbError) as e:
        dev.locate(p)
    assert e.value.code == ca.CMB_ERR_INVALID
    # an index without the locate arrays
    d2 = ca.MoveIndex(mv, with_locate=False)
    with pytest.raises(ca.CmbError):
        d2.locate(ch[0, 1:2])
    d2.close()


def test_exact_matching_end_to_end(mworld):
    """k = 0 on the b-move index: reads in, occurrences out — the oracle's order (forward strand, then reverse complement,
    each as collectTextPositions walks), its counters, and a naive scan of the text"""
    dev, orc, mv, rng = mworld["dev"], mworld["orc"], mworld["mv"], mworld["rng"]
    t = mv.text.tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = [b"", b"A", b"N", b"ACGTN", b"a" * 20, b"AC" * 30]
    for i in range(3000):
        L = int(rng.choice([1, 2, 5, 12, 20, 36, 100, 150, 250]))
        p0 = int(rng.integers(0, len(t) - 1 - L))
        r = bytearray(t[p0:p0 + L])
        u = rng.random()
        if u < 0.15:
            r[int(rng.integers(0, L))] = b"ACGT"[int(rng.integers(0, 4))]
        elif u < 0.2:
            r[int(rng.integers(0, L))] = ord("N")
        elif u < 0.5:
            r = bytearray(bytes(r).translate(comp)[::-1])
        elif u < 0.55:
            r = bytearray(bytes(r).lower())
        reads.append(bytes(r))
    d_occ, d_off, d_cnt = dev.match_exact(reads)
    o_occ, o_off, o_cnt = orc.match_exact(reads)
    assert np.array_equal(d_off, o_off) and d_cnt == o_cnt
    for j, f in enumerate(("begin", "end", "distance", "strand")):
        assert np.array_equal(d_occ[f].astype(np.uint64), o_occ[:, j]), f
    assert d_occ.shape[0] > 20_000 and (d_occ["strand"] == 1).sum() > 3000
    # naive scan for a slice of the reads
    for i in range(6, 300, 5):
        ru = reads[i].upper()
        want = []
        if b"N" not in ru:
            for strand, pat in ((0, ru), (1, ru.translate(comp)[::-1])):
                k = t.find(pat)
                while k >= 0:
                    want.append((k, k + len(pat), strand))
                    k = t.find(pat, k + 1)
        got = [(int(o["begin"]), int(o["end"]), int(o["strand"])) for o in d_occ[int(d_off[i]):int(d_off[i + 1])]]
        assert sorted(got) == sorted(want), i


def test_kmer_table(mworld):
    """populateTable of the RLC flavour on the device: all 4^k entries equal to the oracle's, field by field"""
    dev, orc = mworld["dev"], mworld["orc"]
    for k in (1, 3, 6, 8):
        d, o = dev.kmer_table(k), orc.kmer_table(k)
        _same(d, o, ("kmer table", k))
        assert (d["end"] > d["begin"]).sum() > min(4 ** k, 3000) * 0.9
