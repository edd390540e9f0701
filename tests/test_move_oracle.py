"""The restated b-move backend (oracle/oracle_move.hpp) and the harness builder (columba_amd/movebuild.py) on the CPU.

The move table itself is pinned to the reference's moverepr.cpp by tests/test_oracle_golden.py.  bmove.cpp (extension with
toeholds, phi, locate) cannot be built here (sdsl-lite is absent), so that layer is checked against brute force on the
suffix arrays: the three extension variants must yield the suffix-array intervals of the extended pattern, a toehold that
is an occurrence of it, and locate must return exactly the interval's suffix array values.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

from columba_amd import movebuild  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def _pangenome(rng, base_len, copies, rate):
    base = rng.integers(0, 4, base_len)
    parts = []
    for _ in range(copies):
        s = base.copy()
        m = rng.random(base_len) < rate
        s[m] = rng.integers(0, 4, int(m.sum()))
        parts.append(s)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(parts)]


def test_builder_writes_the_reference_file_bytes():
    """pack_lfbp = the bytes the reference's own writer produced (64-bit vectors of the fixture)."""
    cmds = open(os.path.join(GOLD, "ref_vectors_rlc.cmds")).read().splitlines()
    outs = open(os.path.join(GOLD, "ref_vectors_rlc.out")).read().splitlines()
    n = 0
    for c, o in zip(cmds, outs):
        t = c.split()
        if t[1] != "64":
            continue
        mv = movebuild.build_move(t[2].encode())
        got = (mv.lfbp_rev if t[3] == "1" else mv.lfbp_fwd).tobytes().hex()
        assert got == o.split(" | ")[0], t[2][:40]
        n += 1
    assert n >= 50


def _interval(sa_sorted_suffixes, text, pat):
    """[lo, hi) of suffixes starting with pat, by brute force"""
    import bisect
    lo = bisect.bisect_left(sa_sorted_suffixes, pat)
    hi = bisect.bisect_left(sa_sorted_suffixes, pat + b"\xff")
    return lo, hi


@pytest.fixture(scope="module")
def small_world(oracle_built):
    import oracle_py as op
    rng = np.random.default_rng(77)
    text = np.concatenate([_pangenome(rng, 300, 12, 0.01), np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 500)]])
    mv = movebuild.build_move(text.tobytes())
    t = mv.text.tobytes()
    rt = t[::-1]
    suf = [t[i:] for i in mv.sa.astype(np.int64)]
    rsuf = [rt[i:] for i in mv.rev_sa.astype(np.int64)]
    return {"mv": mv, "orc": op.OracleMoveIndex(mv), "op": op, "t": t, "suf": suf, "rsuf": rsuf, "rng": rng}


def _check_node(w, node, pat):
    mv, t = w["mv"], w["t"]
    lo, hi = _interval(w["suf"], t, pat)
    assert (int(node["begin"][0]), int(node["end"][0])) == (lo, hi), pat
    assert int(node["original_depth"][0]) == len(pat)
    # the toehold is an occurrence of the pattern (its end if it represents the end)
    start = int(node["toehold"][0]) - (len(pat) - 1 if node["toehold_represents_end"][0] else 0)
    assert t[start:start + len(pat)] == pat
    got = np.sort(w["orc"].locate(node))
    assert np.array_equal(got, np.sort(mv.sa[lo:hi])), pat


def test_extension_and_locate_against_the_suffix_array(small_world):
    w = small_world
    orc, t, rng = w["orc"], w["t"], w["rng"]
    code = {65: 1, 67: 2, 71: 3, 84: 4}
    checked = dead = 0
    for trial in range(300):
        L = int(rng.integers(2, 40))
        p0 = int(rng.integers(0, len(t) - 1 - L))
        pat = bytearray(t[p0:p0 + L])
        if rng.random() < 0.3:
            pat[int(rng.integers(0, L))] = b"ACGT"[int(rng.integers(0, 4))]
        pat = bytes(pat)
        # grow the pattern from a random pivot, switching direction at random (as a search scheme does)
        piv = int(rng.integers(0, L))
        lo_i, hi_i = piv, piv  # matched pat[lo_i:hi_i]
        node = orc.complete_range()
        uni = rng.random() < 0.25  # right-to-left only, with the unidirectional variant
        if uni:
            lo_i = hi_i = L
        alive = True
        while alive and (lo_i > 0 or hi_i < L):
            go_back = uni or hi_i == L or (lo_i > 0 and rng.random() < 0.5)
            if go_back:
                ch = pat[lo_i - 1]
                child, ok, _ = orc.extend(2 if uni else 1, node, np.array([code[ch]], dtype=np.uint8))
                lo_i -= 1
            else:
                ch = pat[hi_i]
                child, ok, _ = orc.extend(0, node, np.array([code[ch]], dtype=np.uint8))
                hi_i += 1
            cur = pat[lo_i:hi_i]
            lo, hi = _interval(w["suf"], t, cur)
            if not ok[0]:
                assert lo == hi
                assert int(child["begin"][0]) == 0 and int(child["end"][0]) == 0 and not child["runs_valid"][0]
                alive = False
                dead += 1
                break
            node = child
            _check_node(w, node, cur)
            if not uni:
                rlo, rhi = _interval(w["rsuf"], t[::-1], cur[::-1])
                assert (int(node["rev_begin"][0]), int(node["rev_end"][0])) == (rlo, rhi)
            checked += 1
    assert checked > 2000 and dead > 20


def test_run_indices_are_exact(small_world):
    """begin_run / end_run of every child = the runs that contain begin and end - 1 (whenever they are marked valid)"""
    w = small_world
    orc, mv, rng = w["orc"], w["mv"], w["rng"]
    rows = [orc.rows(0), orc.rows(1)]
    assert rows[0].shape[0] == mv.runs_fwd + 1 and rows[1].shape[0] == mv.runs_rev + 1
    node = orc.complete_range()
    frontier = [node]
    seen = 0
    for depth in range(9):
        nxt = []
        for nd in frontier:
            for mode in (0, 1):
                for c in range(1, 5):
                    child, ok, _ = orc.extend(mode, nd, np.array([c], dtype=np.uint8))
                    if not ok[0]:
                        continue
                    for pre, tab in (("", rows[0]), ("rev_", rows[1])):
                        if child[pre + "runs_valid"][0]:
                            starts = tab[:-1, 1]
                            b, e = int(child[pre + "begin"][0]), int(child[pre + "end"][0])
                            assert int(child[pre + "begin_run"][0]) == np.searchsorted(starts, b, side="right") - 1
                            assert int(child[pre + "end_run"][0]) == np.searchsorted(starts, e - 1, side="right") - 1
                            seen += 1
                    if rng.random() < 0.6:
                        nxt.append(child)
        frontier = nxt[:150]
    assert seen > 500


def test_exact_matching_against_naive_search(small_world):
    """exactMatchesOutput of the RLC flavour (both strands) = every occurrence a naive scan of the text finds"""
    w = small_world
    orc, t, rng = w["orc"], w["t"], w["rng"]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for _ in range(150):
        L = int(rng.integers(1, 60))
        p0 = int(rng.integers(0, len(t) - 1 - L))
        r = bytearray(t[p0:p0 + L])
        u = rng.random()
        if u < 0.2:
            r[int(rng.integers(0, L))] = b"ACGT"[int(rng.integers(0, 4))]
        elif u < 0.3:
            r[int(rng.integers(0, L))] = ord("N")
        elif u < 0.5:
            r = bytearray(bytes(r).translate(comp)[::-1])
        elif u < 0.55:
            r = bytearray(bytes(r).lower())
        reads.append(bytes(r))
    occ, offs, cnt = orc.match_exact(reads)
    assert offs[-1] == occ.shape[0] == cnt["TOTAL_REPORTED_POSITIONS"]

    def find_all(pat):
        out, i = [], t.find(pat)
        while i >= 0:
            out.append(i)
            i = t.find(pat, i + 1)
        return out

    hits = 0
    for i, r in enumerate(reads):
        ru = r.upper()
        want = []
        if b"N" not in ru:
            want = [(p, p + len(ru), 0, 0) for p in find_all(ru)] + [(p, p + len(ru), 0, 1) for p in find_all(ru.translate(comp)[::-1])]
        got = [tuple(int(v) for v in row) for row in occ[int(offs[i]):int(offs[i + 1])]]
        assert sorted(got) == sorted(want), (i, r)
        # forward strand first (searchstrategy.cpp:499-510)
        assert [g[3] for g in got] == sorted(g[3] for g in got)
        hits += len(got) > 0
    assert hits > 80 and cnt["NODE_COUNTER"] > 1000


def test_kmer_table_against_the_suffix_array(small_world):
    """populateTable of the RLC flavour: every entry = the suffix-array intervals of its k-mer in both texts, exact run
    indices of the SA range, a toehold that is an occurrence; absent k-mers are SARangePair()"""
    w = small_world
    orc, t, mv = w["orc"], w["t"], w["mv"]
    starts = orc.rows(0)[:-1, 1]
    for k in (1, 2, 4):
        tab = orc.kmer_table(k)
        present = 0
        for key in range(4 ** k):
            word = bytes(b"ACGT"[(key >> (2 * i)) & 3] for i in range(k - 1, -1, -1))
            lo, hi = _interval(w["suf"], t, word)
            e = tab[key]
            if lo == hi:
                assert int(e["begin"]) == 0 and int(e["end"]) == 0 and e["runs_valid"] == 1 and int(e["original_depth"]) == 0
                continue
            present += 1
            assert (int(e["begin"]), int(e["end"])) == (lo, hi) and int(e["original_depth"]) == k
            rlo, rhi = _interval(w["rsuf"], t[::-1], word[::-1])
            assert (int(e["rev_begin"]), int(e["rev_end"])) == (rlo, rhi)
            assert e["runs_valid"] == 1
            assert int(e["begin_run"]) == np.searchsorted(starts, lo, side="right") - 1
            assert int(e["end_run"]) == np.searchsorted(starts, hi - 1, side="right") - 1
            start = int(e["toehold"]) - (k - 1 if e["toehold_represents_end"] else 0)
            assert t[start:start + k] == word
        assert present >= min(4 ** k, 200)
