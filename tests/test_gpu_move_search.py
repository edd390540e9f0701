"""GPU parity of the approximate search on the b-move backend (BASELINE configs[4]) through the C-ABI:
cmb_move_match_batch against the oracle's restatement of the RUN_LENGTH_COMPRESSION flavour (oracle_move_search.hpp, itself
tied to the FM-index restatement and to plain DP by tests/test_move_search_oracle.py), and against ground truth directly."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

from columba_amd import synth  # noqa: E402
from test_ground_truth import check_completeness, check_soundness, gt  # noqa: E402,F401
from test_gpu_move import _pangenome  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sworld(oracle_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import columba_amd as ca
    from columba_amd import movebuild
    import oracle_py as op
    rng = np.random.default_rng(15)
    # pan-genome-like: 16 haplotypes of a 40 kb sequence with 0.5 % SNPs, a repeat-rich stretch, a random tail
    g = np.concatenate([_pangenome(rng, 40_000, 16, 0.005), synth.genome_rep(seed=3, n=150_000, scale=4.0)[0],
                        np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20_000)]])
    mv = movebuild.build_move(g.tobytes(), device="cuda")
    return {"g": g, "text": g.tobytes(), "mv": mv, "dev": ca.MoveIndex(mv), "orc": op.OracleMoveIndex(mv), "ca": ca, "op": op}


def _tuples(occ, offs, i):
    return [(int(o["begin"]), int(o["end"]), int(o["distance"]), int(o["strand"])) for o in occ[int(offs[i]):int(offs[i + 1])]]


def _compare(w, spec, partition, k, reads, kmer_size=8, metric="edit"):
    import schemes_py as sp
    ca, op = w["ca"], w["op"]
    o_occ, o_off, o_cnt = w["orc"].match_batch(op.OracleStrategy(sp.BY_NAME[spec], metric, partition), k, reads, threads=8, word_size=kmer_size)
    d_occ, d_off, d_cnt = w["dev"].match_batch(ca.SearchStrategy(spec, metric, partition), k, reads, kmer_size=kmer_size)
    assert len(o_occ) > 0
    assert np.array_equal(o_off, d_off)
    strand_only = 0
    for i in range(len(reads)):
        a, b = _tuples(o_occ, o_off, i), _tuples(d_occ, d_off, i)
        if k == 0:
            a, b = sorted(a), sorted(b)
        if a != b:  # (only the strand label of an occurrence found on both strands: unstable sort in the reference)
            assert [t[:3] for t in a] == [t[:3] for t in b], (i, reads[i], a, b)
            strand_only += 1
    assert strand_only <= max(1, len(o_occ) // 500)
    names = ["NODE_COUNTER", "EXPANSIONS", "TOTAL_REPORTED_POSITIONS", "LOCATED_ROWS"]
    if k > 0 and metric == "edit":
        names += ["SEARCH_STARTED", "MATRIX_ROWS"]
    # in-index occurrences that the reference's sort + adjacent-unique leaves in twice are located twice there
    # (Occurrences::eraseDoublesFM, indexhelpers.h:2135-2146: operator< ignores fields operator== compares); the device
    # removes every duplicate, the oracle counts the repeated work
    surplus = {"TOTAL_REPORTED_POSITIONS": o_cnt["SURVIVING_DUP_ROWS"], "LOCATED_ROWS": o_cnt["SURVIVING_DUP_ROWS"]}
    for n in names:
        assert o_cnt[n] - surplus.get(n, 0) == d_cnt[n], (n, o_cnt[n], surplus.get(n, 0), d_cnt[n])
    assert o_cnt["SURVIVING_DUP_ROWS"] * 20 <= max(o_cnt["LOCATED_ROWS"], 1)
    for n in ("IN_TEXT_STARTED", "IMMEDIATE_SWITCH", "ABORTED_IN_TEXT_VERIF", "TEXT_BYTES"):
        assert d_cnt[n] == 0, n
    if k > 0:
        assert 0 < d_cnt["TABLE_ROWS"] <= o_cnt["ROW_STEPS"] * 4  # one scan per parent instead of the reference's walks per character
    return d_occ, d_off, d_cnt, o_cnt


def _reads(g, k, n, length, seed):
    rd = synth.sample_reads(g, n, length, seed=seed, n_frac=0.03, edit_choices=(0, 1, 2, max(k - 1, 0), k, k, k + 1))
    return rd + [b"N" * 60, g[:length].tobytes(), g[-length:].tobytes(), b"ACGT" * 15, b"acgtn" * 8 + g[1000:1100].tobytes().lower()]


@pytest.mark.parametrize("spec,partition,k,length", [
    ("multiple_opt", "dynamic", 6, 250),   # BASELINE configs[4]: 250 bp, k = 6, multiple_opt
    ("multiple_opt", "dynamic", 4, 150),
    ("multiple_opt", "uniform", 2, 100),
    ("multiple_opt", "static", 4, 100),
    ("kuch1", "dynamic", 1, 100),
    ("kuch1", "static", 3, 150),
    ("kuch1", "dynamic", 4, 250),
    ("pigeon", "uniform", 2, 100),
    ("columba", "dynamic", 5, 150),
    ("columba", "dynamic", 3, 100),
    ("minU", "dynamic", 5, 250),
    ("minU", "uniform", 7, 150),
    ("kianfar", "dynamic", 3, 100),        # searches that start with errors allowed in the first part
    ("kuch1", "dynamic", 0, 100),
    ("multiple_opt", "dynamic", 4, 400),   # reads beyond 320 characters: contexts with the match words of 16 row blocks
    ("kuch1", "dynamic", 1, 480),
])
def test_bmove_search_parity(sworld, gt, spec, partition, k, length):
    n = 300 if (spec == "kianfar" or k >= 6) else 1200
    reads = _reads(sworld["g"], k, n, length, seed=70 + k + length)
    occ, offs, cnt, _ = _compare(sworld, spec, partition, k, reads)
    if k > 0:
        checked, _ = check_soundness(gt, sworld["text"], reads[:400], occ, offs, k, "edit")
        assert checked > 100


@pytest.mark.parametrize("spec,partition,k,length", [
    ("kuch1", "dynamic", 2, 100),
    ("kuch1", "static", 3, 150),
    ("multiple_opt", "uniform", 4, 150),
    ("multiple_opt", "dynamic", 6, 250),
    ("columba", "dynamic", 3, 100),
    ("pigeon", "uniform", 1, 100),
    ("kianfar", "dynamic", 4, 100),
    # beyond 8 parts: the greedy schemes of the columba strategy on the wide tables (k_mvs_parts / k_mvs_exact / k_mvs_hbfs<.., MAXP_WIDE>)
    ("columba", "dynamic", 8, 150),
    ("columba", "uniform", 10, 150),
    ("columba", "static", 13, 250),
    ("columba", "dynamic", 13, 150),
])
def test_bmove_hamming_parity(sworld, gt, spec, partition, k, length):
    """recApproxMatchHamming on the b-move index (indexinterface.cpp:1211-1304, RLC branches) + getTextOccHamming"""
    reads = _reads(sworld["g"], k, 1500 if k <= 7 else 600, length, seed=170 + k + length)
    # (substitutions only would be the typical Hamming input; reads with indels simply have fewer occurrences)
    reads += synth.sample_reads(sworld["g"], 500, length, seed=5 + k, p_sub=1.0, p_ins=0.0, edit_choices=(0, 1, k, k))
    occ, offs, cnt, _ = _compare(sworld, spec, partition, k, reads, metric="hamming")
    checked, _ = check_soundness(gt, sworld["text"], reads[:300], occ, offs, k, "hamming")
    assert checked > 50


def test_bmove_search_small_kmer_tables_and_directions(sworld):
    """schemes whose seeds need k-mers of at most four characters (kuch2, 01*0), k-mer sizes 4 and 10, ragged read lengths"""
    g = sworld["g"]
    reads = []
    for ln in (36, 50, 75, 100, 101, 151, 200, 250, 256):
        reads += synth.sample_reads(g, 60, ln, seed=ln, edit_choices=(0, 1, 2, 3))
    _compare(sworld, "kuch2", "dynamic", 3, reads, kmer_size=4)
    _compare(sworld, "01*0", "static", 2, reads, kmer_size=4)
    _compare(sworld, "multiple_opt", "dynamic", 4, reads, kmer_size=10)
    _compare(sworld, "columba", "dynamic", 2, reads, kmer_size=4)


def test_bmove_search_is_complete(sworld, gt):
    """ground truth directly on the device's lists: every end position within k edits is covered"""
    ca = sworld["ca"]
    g = sworld["g"]
    for spec, k in (("multiple_opt", 4), ("kuch1", 2)):
        reads = _reads(g, k, 60, 100, seed=300 + k)
        occ, offs, _ = sworld["dev"].match_batch(ca.SearchStrategy(spec, "edit", "dynamic"), k, reads, kmer_size=8)
        # (the checker works on 32-bit style arrays of begin / end / distance: the records have the same field names)
        hits, chain = check_completeness(gt, sworld["text"], reads, occ, offs, k, "edit")
        assert hits > 60 and chain * 20 <= hits


@pytest.fixture(scope="module")
def tinyworld(oracle_built):
    """a small pan-genome for the naive-backtracking tests: a read of three characters with two errors matches everywhere"""
    import columba_amd as ca
    from columba_amd import movebuild
    import oracle_py as op
    rng = np.random.default_rng(21)
    g = np.concatenate([_pangenome(rng, 3_000, 6, 0.01), np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 4_000)]])
    mv = movebuild.build_move(g.tobytes(), device="cuda")
    return {"g": g, "text": g.tobytes(), "mv": mv, "dev": ca.MoveIndex(mv), "orc": op.OracleMoveIndex(mv), "ca": ca, "op": op}


@pytest.mark.parametrize("spec,metric,partition,k", [("kuch1", "edit", "dynamic", 1), ("multiple_opt", "edit", "uniform", 2),
                                                     ("pigeon", "edit", "static", 3), ("columba", "edit", "dynamic", 3),
                                                     ("kuch1", "hamming", "dynamic", 2), ("columba", "hamming", "uniform", 3)])
def test_bmove_reads_not_longer_than_the_number_of_parts(tinyworld, spec, metric, partition, k):
    """searchstrategy.cpp:148-152, :442-459: such reads are matched by naive backtracking (k_mvs_naive) inside the chunk"""
    g = tinyworld["g"]
    rng = np.random.default_rng(10 * k + len(spec))
    reads = []
    for i in range(40):
        p = int(rng.integers(0, len(g) - 60))
        reads.append(g[p:p + 50].tobytes())
        L = int(rng.integers(1, k + 2))
        p = int(rng.integers(0, len(g) - 10))
        reads.append(g[p:p + L].tobytes())
    if metric == "edit":
        reads.append(b"")
    _compare(tinyworld, spec, partition, k, reads, kmer_size=4, metric=metric)


@pytest.mark.parametrize("metric,k,length", [("edit", 1, 18), ("edit", 2, 20), ("edit", 3, 14), ("hamming", 2, 20), ("hamming", 3, 26)])
def test_bmove_naive_strategy(tinyworld, metric, k, length):
    """`-S naive` (NaiveBackTrackingStrategy, searchstrategy.h:2785-2820): every read by backtracking over the whole pattern"""
    g = tinyworld["g"]
    rng = np.random.default_rng(5 * k + length)
    reads = []
    for _ in range(60):
        L = int(rng.integers(max(2, length - 5), length + 1))
        p = int(rng.integers(0, len(g) - L - 1))
        r = bytearray(g[p:p + L].tobytes())
        for _ in range(int(rng.integers(0, k + 1))):
            r[int(rng.integers(0, L))] = b"ACGT"[int(rng.integers(0, 4))]
        reads.append(bytes(r))
    reads += [b"ACGTN", b"acgtacgtac"]
    os.environ["CMB_TEST_SMALL_POOLS"] = "1"   # (the node queue of the naive search starts small and is grown)
    try:
        _compare(tinyworld, "naive", "dynamic", k, reads, kmer_size=4, metric=metric)
    finally:
        del os.environ["CMB_TEST_SMALL_POOLS"]


@pytest.mark.parametrize("spec,k,length", [("multiple_opt", 4, 150), ("columba", 6, 250), ("kuch1", 2, 100), ("columba", 7, 100),
                                           ("columba", 9, 150), ("columba", 10, 150), ("columba", 12, 200)])   # (beyond 9 errors: k_cigar_wide, round 4)
def test_bmove_alignments(sworld, spec, k, length):
    """CIGAR and sequence of the occurrences on the b-move index (cmb_move_attach_text + cmb_move_batch_alignments: findCIGAR on
    text[begin, end), which is the matched string the reference's search carries along for this flavour): equal to those of the
    FM-index flavour on the same text — the two flavours report the same occurrences when the FM-index never switches to in-text
    verification (tests/test_move_search_oracle.py), and that path's alignments are checked against the oracle — and every CIGAR
    re-derives its distance."""
    from columba_amd import indexbuild as ib
    ca = sworld["ca"]
    g = sworld["g"]
    starts = np.array([0, 250_000, 640_000, len(g)], dtype=np.uint64)
    reads = _reads(g, k, 500, length, seed=40 + k)
    for s0 in (250_000, 640_000):   # reads across sequence boundaries: spans = 1
        reads.append(g[s0 - length // 2:s0 + length - length // 2].tobytes())
    st = ca.SearchStrategy(spec, "edit", "dynamic")
    fm = ca.Index(ib.build_index(sworld["text"], seq_starts=starts.astype(np.int64), device="cuda"), in_text_switch=0, kmer_size=8)
    fb = ca.Batch(fm, st, k, reads)
    fb.want_alignments()
    fb.run()
    f_occ, f_off, _ = fb.results()
    f_aln, f_ops = fb.alignments()
    sworld["dev"].attach_text(sworld["text"], starts)
    mb = ca.MoveBatch(sworld["dev"], st, k, reads=reads, kmer_size=8)
    mb.want_alignments()
    mb.run()
    m_occ, m_off, _ = mb.results()
    m_aln, m_ops = mb.alignments()
    assert len(m_occ) > 400 and np.array_equal(f_off, m_off)
    for f in ("begin", "end", "distance"):
        assert np.array_equal(f_occ[f].astype(np.uint64), m_occ[f].astype(np.uint64)), f
    assert np.array_equal(f_aln["seq_id"], m_aln["seq_id"]) and np.array_equal(f_aln["seq_begin"], m_aln["seq_begin"])
    assert np.array_equal(f_aln["spans"], m_aln["spans"]) and int(m_aln["spans"].sum()) >= 2
    same_strand = f_occ["strand"] == m_occ["strand"]
    assert (~same_strand).sum() <= max(1, len(m_occ) // 500)
    n_edits = 0
    for j in range(len(m_occ)):
        a, b = m_aln[j], f_aln[j]
        cm = ca.cigar_string(m_ops[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])])
        if same_strand[j]:
            assert cm == ca.cigar_string(f_ops[int(b["cigar_off"]):int(b["cigar_off"]) + int(b["cigar_len"])]), (j, m_occ[j])
        # the CIGAR is an alignment of the read with its text window at the reported distance
        import re
        ops = [(int(n), o) for n, o in re.findall(r"(\d+)([MID])", cm)]
        assert sum(n for n, o in ops if o in "MI") == len(reads[int(np.searchsorted(m_off, j, side="right") - 1)])
        assert sum(n for n, o in ops if o in "MD") == int(m_occ[j]["end"]) - int(m_occ[j]["begin"])
        n_edits += int(m_occ[j]["distance"]) > 0
    assert n_edits > 50
    # ... and the SAM records of the chunk (generateOutputSingleEnd through cmb_sam_chunk on the text-only index) are the FM-index flavour's
    ids = [f"r{i} extra" for i in range(len(reads))]
    quals = ["I" * len(r) for r in reads]
    names = ["chrA", "chrB", "chrC"]
    sam_m = mb.sam(ids, quals, names).splitlines()
    sam_f = fb.sam(ids, quals, names).splitlines()
    assert len(sam_m) == len(sam_f) > len(reads)
    diff = [i for i in range(len(sam_m)) if sam_m[i] != sam_f[i]]
    assert len(diff) <= max(2, len(sam_m) // 300), (len(diff), sam_m[diff[0]], sam_f[diff[0]])   # (strand labels of palindromic hits)
    fb.close()
    mb.close()
    fm.close()


@pytest.mark.parametrize("env", [{"CMB_MATRIX64": "1"}, {"CMB_TEST_NARROW_WV": "2"}, {"CMB_MOVE_POS64": "1"}])
def test_bmove_frontier_code_paths(sworld, monkeypatch, env):
    """Round 4: up to 7 errors the frontier of this backend carries the in-index matrix on 32-bit words (GeoN32) and, on indexes below 2^32,
    works on 32-bit positions with the children in slots (mvExpandSlots) — every test above runs on that.  Here the same lists and counters
    on the reference's 64-bit matrix words, through the re-run a phase triggers that does not fit the small matrix (bound lowered by the
    test hook), and on the general 40-bit positions."""
    ca = sworld["ca"]
    reads = _reads(sworld["g"], 5, 600, 150, seed=77)
    st = ca.SearchStrategy("columba", "edit", "dynamic")
    a = sworld["dev"].match_batch(st, 5, reads, kmer_size=8)
    for n, v in env.items():
        monkeypatch.setenv(n, v)
    b = sworld["dev"].match_batch(st, 5, reads, kmer_size=8)
    assert len(a[0]) > 300
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_bmove_pool_growth(sworld):
    """pools that start from almost nothing are grown and the search re-run: same lists"""
    ca = sworld["ca"]
    reads = _reads(sworld["g"], 4, 400, 150, seed=9)
    st = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
    a = sworld["dev"].match_batch(st, 4, reads, kmer_size=8)
    os.environ["CMB_TEST_SMALL_POOLS"] = "1"
    try:
        b = sworld["dev"].match_batch(st, 4, reads, kmer_size=8)
    finally:
        del os.environ["CMB_TEST_SMALL_POOLS"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


@pytest.mark.parametrize("spec,metric,x,min_identity", [("columba", "edit", 0, 96), ("columba", "edit", 1, 96), ("multiple_opt", "edit", 0, 97),
                                                        ("kuch1", "hamming", 0, 98), ("minU", "edit", 2, 97), ("columba", "hamming", 1, 96),
                                                        ("kuch1", "edit", 0, 50), ("columba", "edit", 0, 92), ("columba", "edit", 1, 93)])   # (strata beyond 7 errors: round 4)
def test_bmove_best_mode(sworld, spec, metric, x, min_identity):
    """BEST (+x strata) mode on the b-move index (cmb_move_match_best: matchApproxBestPlusX with b-move batches as strata, every
    strand filtered by itself; CIGARs and trimming from the text beside the index) against the oracle's restatement of the
    RUN_LENGTH_COMPRESSION flavour: best distance and hits per read, alignments in the reference's order, sequences, CIGARs, counters."""
    import schemes_py as sp
    ca, op = sworld["ca"], sworld["op"]
    g = sworld["g"]
    starts = np.array([0, 250_000, 640_000, len(g)], dtype=np.uint32)
    sworld["dev"].attach_text(sworld["text"], starts)
    sworld["orc"].attach_text(sworld["text"], starts, word_size=8)
    deep = min_identity in (92, 93)   # identities that let the strata run beyond 7 errors (150 bp: 12 / 10)
    reads = synth.sample_reads(g, 500 if deep else 1200, 150, seed=300 + x, n_frac=0.01, edit_choices=(0, 1, 3, 6, 8, 9, 10, 11, 12, 13) if deep else (0, 0, 1, 2, 3, 5, 6, 9))
    for s0 in (250_000, 640_000):   # reads across sequence ends: trimmed or dropped (findSeqName)
        reads += [g[s0 - 75:s0 + 75].tobytes(), g[s0 - 3:s0 + 147].tobytes(), g[s0 - 147:s0 + 3].tobytes(), g[s0 - 5:s0 + 145].tobytes()]
    reads += [b"ACGT" * 37 + b"AC", b"N" * 150, g[:150].tobytes(), g[-150:].tobytes()]
    if min_identity == 50:   # strata of reads not longer than the number of parts: naive backtracking inside a stratum's batch
        reads = reads[:400] + reads[1200:] + [b"A", b"AC", b"ACG", b"ACGTA", b"GATTACA", b""]
    tab = sp.BY_NAME[spec]
    max_sup = 0
    while (max_sup + 1) in tab["schemes"]:
        max_sup += 1
    max_sup = min(max_sup, 13)   # (strata up to the reference's MAX_K, as on the FM-index)
    o_occ, o_sid, o_sb, o_cig, o_off, o_best, o_hits, o_cnt = op.match_best(
        sworld["orc"], op.OracleStrategy(tab, metric, "dynamic"), reads, x=x, min_identity=min_identity, max_supported=max_sup, threads=8,
        word_size=8)
    d_occ, d_aln, d_ops, d_off, d_best, d_hits, d_cnt = ca.match_best(
        sworld["dev"], ca.SearchStrategy(spec, metric, "dynamic"), reads, x=x, min_identity=min_identity, kmer_size=8)
    assert np.array_equal(o_best, d_best)
    assert (o_best != 0xFFFFFFFF).sum() > 150 and (o_best == 0xFFFFFFFF).sum() > 0
    if deep:
        assert ((o_best != 0xFFFFFFFF) & (o_best > 7)).sum() > 20   # best alignments beyond 7 errors are among them
    assert np.array_equal(o_hits, d_hits) and np.array_equal(o_off, d_off)
    for f in ("begin", "end", "distance"):
        assert np.array_equal(o_occ[f], d_occ[f]), f
    same = o_occ["strand"] == d_occ["strand"]
    assert (~same).sum() <= max(1, len(d_occ) // (20 if min_identity == 50 else 500))   # (the strand label of an occurrence found on both strands)
    assert np.array_equal(o_sid, d_aln["seq_id"]) and np.array_equal(o_sb, d_aln["seq_begin"])
    for j in range(len(d_occ)):
        a = d_aln[j]
        got = ca.cigar_string(d_ops[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])])
        assert not same[j] or got == o_cig[j], (j, d_occ[j], got, o_cig[j])
    for n in ("NODE_COUNTER", "SEARCH_STARTED", "EXPANSIONS"):
        assert o_cnt[n] == d_cnt[n], (n, o_cnt[n], d_cnt[n])
    for n in ("IN_TEXT_STARTED", "IMMEDIATE_SWITCH", "ABORTED_IN_TEXT_VERIF"):
        assert d_cnt[n] == 0, n
    if (spec, metric, x) == ("columba", "edit", 0) and not deep:   # the reads 3 and 5 characters over a sequence end are found with trimming
        assert d_best[1205] <= 3 and d_best[1207] <= 5 and int(d_aln[int(d_off[1205])]["seq_begin"]) == 0


@pytest.mark.parametrize("partition,k,length", [("dynamic", 8, 150), ("uniform", 10, 150), ("dynamic", 11, 150), ("static", 12, 250), ("dynamic", 13, 150)])
def test_bmove_edit_distance_beyond_seven_errors(sworld, gt, partition, k, length):
    """the greedy schemes on the b-move index under edit distance: the wide record geometries of the frontier (mvExpand / bfsHeavy on
    GeoW, from 11 errors on GeoX with its 16-row blocks) — occurrences and counters against the oracle's RLC flavour, which takes the
    reference's 64- or 128-bit matrix per search part"""
    reads = _reads(sworld["g"], k, 250, length, seed=400 + k)
    occ, offs, _, _ = _compare(sworld, "columba", partition, k, reads)
    # ... and against ground truth: every reported window has its distance by dynamic programming, every window within k errors is
    # covered (tests/test_ground_truth.py)
    checked, _ = check_soundness(gt, sworld["text"], reads[:60], occ, offs, k, "edit")
    hits, chain = check_completeness(gt, sworld["text"], reads[:25], occ, offs, k, "edit")
    assert checked > 20 and hits > 10 and chain * 20 <= hits


def test_bmove_read_pairs_in_best_mode(sworld):
    """Read pairs in BEST mode on the b-move index (ca.pair_chunk_sam_best over MoveBatch: the chunk walks through its strata together,
    cmb_pair_best_*): the records equal those of the FM-index flavour on the same text when that one never switches to in-text
    verification — the two flavours report the same occurrences then — and the simulated fragments come back as proper pairs."""
    from columba_amd import indexbuild as ib
    ca = sworld["ca"]
    g = sworld["g"]
    starts = np.array([0, 250_000, 640_000, len(g)], dtype=np.uint64)
    rng = np.random.default_rng(91)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    n, L = 160, 100
    r1, r2 = [], []
    for i in range(n):
        frag = int(rng.integers(220, 400))
        p0 = int(rng.integers(640_100, len(g) - 1000)) if i % 3 else int(rng.integers(100, 600_000))  # (unique tail / repeated haplotypes)
        a = bytearray(g[p0:p0 + L].tobytes())
        b = bytearray(g[p0 + frag - L:p0 + frag].tobytes().translate(comp)[::-1])
        for m in (a, b):
            for _ in range(int(rng.integers(0, 3))):
                q = int(rng.integers(1, L - 1))
                m[q] = b"ACGT"[(b"ACGT".index(bytes([m[q]])) + 1) % 4]
        if i % 2:
            a, b = b, a
        r1.append(bytes(a))
        r2.append(bytes(b))
    ids1, ids2, quals = [f"@f{i}/1" for i in range(n)], [f"@f{i}/2" for i in range(n)], ["I" * L] * n
    names = ["hapA", "hapB", "tail"]
    st = ca.SearchStrategy("columba", "edit", "dynamic")
    sworld["dev"].attach_text(sworld["text"], starts)
    text, mapped, batches = ca.pair_chunk_sam_best(sworld["dev"], st, r1, r2, ids1, ids2, quals, quals, names, x=0, min_identity=95,
                                                   orientation=ca.ORIENTATION_FR, max_frag=600, min_frag=100, kmer_size=8)
    fm = ca.Index(ib.build_index(sworld["text"], seq_starts=starts.astype(np.int64), device="cuda"), in_text_switch=0, kmer_size=8)
    want, mapped_fm, _ = ca.pair_chunk_sam_best(fm, st, r1, r2, ids1, ids2, quals, quals, names, x=0, min_identity=95,
                                                orientation=ca.ORIENTATION_FR, max_frag=600, min_frag=100)
    assert text == want and mapped == mapped_fm and mapped > 0.85 * n and batches <= 80
    flags = [int(ln.split("\t")[1]) for ln in text.splitlines()]
    assert sum(1 for f in flags if f & 2) >= 2 * 0.8 * n


@pytest.mark.parametrize("halves", [2, 3])
def test_bmove_concurrent_halves(sworld, monkeypatch, halves):
    """A large chunk is matched as concurrent halves, each a batch of its own (from 2^19 reads; CMB_MOVE_SUBBATCHES forces it here): occurrences,
    offsets, counters and alignments are those of the one batch"""
    ca = sworld["ca"]
    g = sworld["g"]
    starts = np.array([0, 250_000, 640_000, len(g)], dtype=np.uint64)
    sworld["dev"].attach_text(sworld["text"], starts)
    reads = _reads(g, 4, 3001, 150, seed=77) + [b"N" * 150, g[:150].tobytes()]
    st = ca.SearchStrategy("columba", "edit", "dynamic")

    def run():
        mb = ca.MoveBatch(sworld["dev"], st, 4, reads=reads, kmer_size=8)
        mb.want_alignments()
        mb.filter_per_strand()
        mb.run()
        occ, offs, cnt = mb.results()
        aln, ops = mb.alignments()
        t = mb.timings()
        mb.close()
        return occ, offs, cnt, aln, ops, t

    monkeypatch.setenv("CMB_MOVE_SUBBATCHES", "1")
    a = run()
    monkeypatch.setenv("CMB_MOVE_SUBBATCHES", str(halves))
    b = run()
    assert len(a[0]) > 3000 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and set(a[5]) == set(b[5])
