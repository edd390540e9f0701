"""The approximate search of the run-length compressed flavour as restated in oracle/ (MatcherT over the b-move index) on the
CPU.  Neither bmove.cpp (sdsl-lite) nor indexinterface.cpp (parallel_hashmap) of the reference can be built here, so this
restatement is PARITY UNPINNED; what stands in for a pin:

  * the two flavours of the reference are ONE search layer compiled twice; with in-text verification switched off (switch
    point 0) the FM-index flavour walks exactly the tree the b-move flavour walks — same suffix-array intervals, same matrix
    rows, same cluster centres — so occurrences, NODE_COUNTER, EXPANSIONS, SEARCH_STARTED and MATRIX_ROWS must be equal between
    the FM-index restatement (which the GPU tests tie to the HIP path and the golden vectors tie to the reference's
    function-level code) and the b-move restatement;
  * ground truth (oracle/groundtruth.c): soundness and completeness of the b-move flavour's lists by plain dynamic programming.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

from columba_amd import indexbuild as ib, movebuild, synth  # noqa: E402
from test_ground_truth import check_completeness, check_soundness, gt  # noqa: E402,F401
from test_move_oracle import _pangenome  # noqa: E402


@pytest.fixture(scope="module")
def world(oracle_built):
    import oracle_py as op
    rng = np.random.default_rng(5)
    # pan-genome-like: 12 copies of a 6 kb sequence with 1 % SNPs, plus a repeat-rich stretch and a random tail
    g = np.concatenate([_pangenome(rng, 6000, 12, 0.01), synth.genome_rep(seed=3, n=60_000, scale=4.0)[0],
                        np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 4000)]])
    text = g.tobytes()
    mv = movebuild.build_move(text)
    ix = ib.build_index(text, device="cpu")
    return {"g": g, "text": text, "mv": mv, "move": op.OracleMoveIndex(mv), "fm0": op.OracleIndex(ix, switch_point=0, kmer_size=6),
            "fm0_4": op.OracleIndex(ix, switch_point=0, kmer_size=4), "op": op}


CONFIGS = [
    ("multiple_opt", "edit", "dynamic", 4, 100),
    ("multiple_opt", "edit", "dynamic", 6, 250),
    ("multiple_opt", "edit", "uniform", 2, 60),
    ("kuch1", "edit", "static", 3, 100),
    ("kuch1", "edit", "dynamic", 1, 150),
    ("columba", "edit", "dynamic", 5, 150),
    ("columba", "edit", "dynamic", 7, 150),
    ("minU", "edit", "uniform", 3, 80),
    ("pigeon", "edit", "dynamic", 2, 100),
    ("kianfar", "edit", "dynamic", 3, 100),
    ("kuch2", "edit", "dynamic", 3, 100),
    ("01*0", "edit", "static", 2, 100),
    ("kuch1", "hamming", "dynamic", 2, 100),
    ("multiple_opt", "hamming", "uniform", 4, 150),
    ("columba", "hamming", "dynamic", 3, 100),
    ("kuch1", "edit", "dynamic", 0, 100),
]


@pytest.mark.parametrize("spec,metric,partition,k,length", CONFIGS)
def test_bmove_search_equals_fm_search_without_in_text_verification(world, gt, spec, metric, partition, k, length):
    import schemes_py as sp
    op = world["op"]
    n = 30 if spec == "kianfar" or k >= 6 else 80
    reads = synth.sample_reads(world["g"], n, length, seed=40 + k + length, n_frac=0.05,
                               edit_choices=(0, 1, 2, max(k - 1, 0), k, k, k + 1))
    reads += [b"N" * 40, world["text"][:length], world["text"][-length:], b"ACGT" * 12, b"acgtn" * 9 + world["text"][100:160].lower()]
    small = spec in ("kuch2", "01*0")
    ws = 4 if small else 6
    st = op.OracleStrategy(sp.BY_NAME[spec], metric, partition)
    m_occ, m_off, m_cnt = world["move"].match_batch(st, k, reads, threads=4, word_size=ws)
    f_occ, f_off, f_cnt = op.match_batch(world["fm0_4" if small else "fm0"], st, k, reads, threads=4)
    assert len(m_occ) > 0
    assert np.array_equal(m_off, f_off)
    for i in range(len(reads)):
        a = [tuple(int(x) for x in o) for o in m_occ[int(m_off[i]):int(m_off[i + 1])]]
        b = [tuple(int(x) for x in o) for o in f_occ[int(f_off[i]):int(f_off[i + 1])]]
        if k == 0:  # (exact matches stay in the order they were located: suffix-array order there, phi order here)
            a, b = sorted(a), sorted(b)
        if a != b:  # only the strand label of an occurrence found on both strands may differ (unstable sort in the reference)
            assert [t[:3] for t in a] == [t[:3] for t in b], (i, reads[i], a, b)
    for name in ("NODE_COUNTER", "EXPANSIONS", "SEARCH_STARTED", "MATRIX_ROWS"):
        assert m_cnt[name] == f_cnt[name], (name, m_cnt[name], f_cnt[name])
    for name in ("IN_TEXT_STARTED", "IMMEDIATE_SWITCH", "ABORTED_IN_TEXT_VERIF", "TEXT_BYTES"):
        assert m_cnt[name] == 0 and f_cnt[name] == 0, name
    # in-index occurrences are equal under FMOcc::== only if toehold and run indices agree as well (indexhelpers.h:1226-1233):
    # the b-move flavour merges fewer of them before locating
    assert m_cnt["TOTAL_REPORTED_POSITIONS"] >= f_cnt["TOTAL_REPORTED_POSITIONS"] - f_cnt["SURVIVING_DUP_ROWS"]
    assert m_cnt["ROW_STEPS"] > 0 and m_cnt["LOCATED_ROWS"] > 0
    if k > 0:
        checked, _ = check_soundness(gt, world["text"], reads, m_occ, m_off, k, metric)
        assert checked > 20
    if k in (2, 3) and length <= 100:
        hits, chain = check_completeness(gt, world["text"], reads[:25], m_occ, m_off, k, metric)
        assert hits > 10 and chain * 20 <= hits


@pytest.mark.parametrize("spec,metric,x,min_identity", [("columba", "edit", 0, 96), ("columba", "edit", 1, 95), ("kuch1", "hamming", 0, 97),
                                                        ("minU", "edit", 2, 97)])
def test_bmove_best_mode_equals_fm_best_mode_without_in_text_verification(world, spec, metric, x, min_identity):
    """BEST (+x strata) mode is ONE function over both flavours (matchApproxBestPlusX, searchstrategy.cpp:623-746); what differs
    below it is where the CIGAR's reference string comes from (the matched string instead of the text, indexinterface.h:966-971) and
    how an occurrence over a sequence end is trimmed (checkTrimmedMatch on the matched string, indexinterface.cpp:722-796, instead
    of inTextVerificationOneString) — both read text[begin, end): same alignments, sequences, CIGARs, best distances and hit counts
    as the FM-index restatement with switch point 0 on the same text."""
    import schemes_py as sp
    op = world["op"]
    g = world["g"]
    n = len(world["text"])
    starts = np.array([0, 30_000, 71_000, 100_000, n], dtype=np.int64)
    ix = ib.build_index(world["text"], seq_starts=starts, device="cpu")
    fm = op.OracleIndex(ix, switch_point=0, kmer_size=6)
    world["move"].attach_text(world["text"], np.asarray(ix.seq_starts, dtype=np.uint32), word_size=6)
    reads = synth.sample_reads(g, 300, 120, seed=70 + x, n_frac=0.02, edit_choices=(0, 0, 1, 2, 3, 5, 8))
    for s in starts[1:-1]:   # reads across sequence ends: trimmed or dropped
        reads += [g[int(s) - 60:int(s) + 60].tobytes(), g[int(s) - 3:int(s) + 117].tobytes(), g[int(s) - 117:int(s) + 3].tobytes()]
    reads += [b"ACGT" * 30, b"N" * 120]
    tab = sp.BY_NAME[spec]
    st = op.OracleStrategy(tab, metric, "dynamic")
    a = op.match_best(world["move"], st, reads, x=x, min_identity=min_identity, max_supported=7, threads=4, word_size=6)
    b = op.match_best(fm, st, reads, x=x, min_identity=min_identity, max_supported=7, threads=4)
    assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6]) and np.array_equal(a[4], b[4])
    assert (a[5] != 0xFFFFFFFF).sum() > 150 and (a[5] == 0xFFFFFFFF).sum() > 0
    for f in ("begin", "end", "distance"):
        assert np.array_equal(a[0][f], b[0][f]), f
    assert (a[0]["strand"] != b[0]["strand"]).sum() <= 2   # (palindromic hits: the strand label of a tie)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    same = a[0]["strand"] == b[0]["strand"]
    assert all(ca == cb for ca, cb, s2 in zip(a[3], b[3], same) if s2)
    for c in ("NODE_COUNTER", "EXPANSIONS", "SEARCH_STARTED"):
        assert a[7][c] == b[7][c], c
