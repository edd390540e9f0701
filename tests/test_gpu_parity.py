"""Parity of the HIP path (through the C-ABI) with the CPU oracle on the same seeded inputs.

Bar: bit-exact (all arithmetic on the path is integer).  Run with `pytest -m gpu` on an MI355X.
"""
import numpy as np
import pytest

import columba_amd as ca
from columba_amd import indexbuild as ib
from columba_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(oracle_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import oracle_py as op
    # repeat-rich 2 Mbp genome: enough repeats that the in-index DFS (not only in-text
    # verification) is exercised
    g, starts = synth.genome_rep(seed=11, n=2_000_000, scale=1.5)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
    dev = ca.Index(ix)
    orc = op.OracleIndex(ix)
    # kuch2 and 01*0 place their seeds for k-mers of at most 4 characters (searchstrategy.h:3020, :3204; the
    # reference's CLI lowers the k-mer size of the index to 4 for them, alignparameters.cpp:1275-1278)
    return {"genome": g, "ix": ix, "dev": dev, "orc": orc, "op": op,
            "dev4": ca.Index(ix, kmer_size=4), "orc4": op.OracleIndex(ix, kmer_size=4)}


def _tuples(occs, offs, i):
    return [(int(o["begin"]), int(o["end"]), int(o["distance"]), int(o["strand"]))
            for o in occs[int(offs[i]):int(offs[i + 1])]]


def _compare(world, spec_name, metric, partition, k, reads, counters=True, dups_rare=True):
    import schemes_py as sp
    op = world["op"]
    small = spec_name in ("kuch2", "01*0") and "dev4" in world
    ost = op.OracleStrategy(sp.BY_NAME[spec_name], metric, partition)
    o_occ, o_off, o_cnt = op.match_batch(world["orc4" if small else "orc"], ost, k, reads, threads=8)
    dst = ca.SearchStrategy(spec_name, metric, partition)
    d_occ, d_off, d_cnt = ca.match_batch(world["dev4" if small else "dev"], dst, k, reads)
    assert len(o_occ) > 0
    strand_only = 0
    for i in range(len(reads)):
        a, b = _tuples(o_occ, o_off, i), _tuples(d_occ, d_off, i)
        if k == 0:
            # exact matches come back unsorted from the reference (suffix-array order, forward strand
            # then reverse complement: searchstrategy.cpp:499-510); the C-ABI returns them sorted
            a, b = sorted(a), sorted(b)
        if a != b:
            # equal (begin,end,distance) found on both strands: the reference's own choice is
            # unspecified (unstable sort, indexhelpers.h:2148-2156) — only that may differ
            assert [t[:3] for t in a] == [t[:3] for t in b], (i, reads[i], a, b)
            strand_only += 1
    assert strand_only <= max(1, len(o_occ) // 1000)
    if counters:
        names = ["NODE_COUNTER", "IN_TEXT_STARTED", "IMMEDIATE_SWITCH", "SEARCH_STARTED", "EXPANSIONS",
                 "LF_STEPS", "LOCATED_ROWS", "TEXT_BYTES", "MATRIX_ROWS"]
        if k > 0:
            names += ["ABORTED_IN_TEXT_VERIF", "TOTAL_REPORTED_POSITIONS"]
        if k > 0 and metric == "edit":
            names.append("CIGARS_IN_TEXT_VERIFICATION")
        # in-index occurrences that the reference's sort + adjacent-unique leaves in twice are located twice there
        # (Occurrences::eraseDoublesFM, indexhelpers.h:2135-2146: operator< ignores depth and strand, operator== does
        # not, std::sort is unstable); the device removes every duplicate.  The oracle counts that repeated work.
        surplus = {"LF_STEPS": o_cnt["SURVIVING_DUP_LF"], "LOCATED_ROWS": o_cnt["SURVIVING_DUP_ROWS"],
                   "TOTAL_REPORTED_POSITIONS": o_cnt["SURVIVING_DUP_ROWS"]}
        for n in names:
            assert o_cnt[n] - surplus.get(n, 0) == d_cnt[n], (n, o_cnt[n], surplus.get(n, 0), d_cnt[n])
        if dups_rare:  # (well below 1 % at the default switch point; few rows are located at all when it is 0)
            assert o_cnt["SURVIVING_DUP_ROWS"] * 200 <= max(o_cnt["LOCATED_ROWS"], 1)
    return o_cnt


def test_rank_and_extend(world):
    rng = np.random.default_rng(5)
    n = world["ix"].n
    p = rng.integers(0, n + 1, 20000).astype(np.uint64)
    p[:4] = [0, 1, n, n - 1]
    c = rng.integers(0, 4, p.shape[0]).astype(np.uint32)
    for rev in (0, 1):
        assert np.array_equal(world["dev"].rank(rev, c, p), world["orc"].rank(rev, c, p))
    # extension of ranges reached by real searches + random ranges (incl. empty / full)
    b = rng.integers(0, n, 20000)
    w = np.minimum((2.0 ** rng.uniform(0, np.log2(n), 20000)).astype(np.int64), n - b)
    b2 = rng.integers(0, n, 20000)
    r = np.stack([b, b + w, b2, np.minimum(b2 + w, n)], axis=1).astype(np.uint32)
    r[0] = [0, n, 0, n]
    r[1] = [5, 5, 7, 7]
    for mode in (0, 1, 2):
        do, dk = world["dev"].extend(mode, r)
        oo, ok = world["orc"].extend(mode, r)
        assert np.array_equal(dk, ok)
        assert np.array_equal(do, oo)


def test_kmer_table_and_locate(world):
    assert np.array_equal(world["dev"].kmer_table(), world["orc"].kmer_table())
    rng = np.random.default_rng(6)
    rows = rng.integers(0, world["ix"].n, 50000).astype(np.uint32)
    dp, dlf = world["dev"].locate(rows)
    opos, olf = world["orc"].locate(rows)
    assert np.array_equal(dp, opos) and dlf == olf
    # locate is the inverse of the suffix array: text positions are a permutation
    allrows = np.arange(0, min(world["ix"].n, 200000), dtype=np.uint32)
    pos, _ = world["dev"].locate(allrows)
    assert len(np.unique(pos)) == len(allrows)


def test_in_text_verification_hook(world):
    g = world["genome"]
    rng = np.random.default_rng(8)
    for trial in range(60):
        # (k = 7 with a free start: the reference's 128-bit matrix; 8 ... 10: k_verify_wide, the same band on 64-bit words with a wide left margin)
        k = int(rng.integers(1, 8)) if trial < 30 else int(rng.integers(8, 14))   # (11 ... 13: 8-row blocks, a left margin of 40 bits)
        pos = int(rng.integers(100, len(g) - 400))
        pat = synth.sample_reads(g[pos:pos + 400], 1, int(rng.choice([50, 100, 150, 250])), seed=trial,
                                 edit_choices=(0, 1, 2, k, max(k - 1, 0)), rc_frac=0.0)[0]
        fixed = bool(trial % 2)
        starts = np.concatenate([rng.integers(0, len(g), 20), np.arange(max(0, pos - 30), pos + 60, 3),
                                 [len(g) - 5, len(g), 0]]).astype(np.uint32)
        d, dc = world["dev"].verify(pat, starts, k, int(trial % 3 == 0), fixed)
        o, oc = world["orc"].verify(pat, starts, k, int(trial % 3 == 0), fixed)
        key = lambda a: sorted((int(x["begin"]), int(x["end"]), int(x["distance"])) for x in a)
        assert key(d) == key(o), trial
        for n in ("IN_TEXT_STARTED", "ABORTED_IN_TEXT_VERIF", "CIGARS_IN_TEXT_VERIFICATION", "MATRIX_ROWS"):
            assert dc[n] == oc[n], (trial, n)


def test_production_edit_verification_path(world):
    """The path cmb_batch_run takes for edit-distance candidates — k_verify<keys> -> radix sort + run-length encode ->
    k_verify_stage x stages -> k_traceback — at function level (cmb_verify_batch_staged), WITH duplicate candidates:
    identical candidates are verified once and every counter is scaled by the multiplicity, which must give exactly
    the counters of the reference verifying each of them (FMIndex::inTextVerification, fmindex.cpp:267-310)."""
    g = world["genome"]
    rng = np.random.default_rng(18)
    dup_total = 0
    for trial in range(36):
        k = int(rng.integers(1, 8)) if trial < 24 else 8 + trial % 6   # (from k = 7 with a free start: the reference's 128-bit matrix;
        #                                                                  beyond 7: k_wide_filter + k_verify_wide on the band of that matrix)
        pos = int(rng.integers(100, len(g) - 400))
        pat = synth.sample_reads(g[pos:pos + 400], 1, int(rng.choice([50, 100, 150, 250])), seed=1000 + trial,
                                 edit_choices=(0, 1, 2, k, max(k - 1, 0)), rc_frac=0.0)[0]
        fixed = bool(trial % 2)
        base = np.concatenate([rng.integers(0, len(g), 20), np.arange(max(0, pos - 30), pos + 60, 3),
                               [len(g) - 5, len(g), 0]]).astype(np.uint32)
        # every candidate 1..5 times, shuffled (the parts of a read seeding the same alignment look like this)
        mult = rng.integers(1, 6, base.shape[0])
        starts = np.repeat(base, mult)
        rng.shuffle(starts)
        dup_total += int(starts.shape[0] - base.shape[0])
        min_ed = int(trial % 3 == 0)
        d, dc = world["dev"].verify(pat, starts, k, min_ed, fixed, staged=True)
        o, oc = world["orc"].verify(pat, starts, k, min_ed, fixed)
        key = lambda a: sorted(set((int(x["begin"]), int(x["end"]), int(x["distance"])) for x in a))
        assert key(d) == key(o), trial
        # the direct kernel (one candidate per lane, no de-duplication) must agree as well, duplicates included
        d1, dc1 = world["dev"].verify(pat, starts, k, min_ed, fixed)
        assert sorted((int(x["begin"]), int(x["end"]), int(x["distance"])) for x in d1) == \
            sorted((int(x["begin"]), int(x["end"]), int(x["distance"])) for x in o), trial
        for n in ("IN_TEXT_STARTED", "ABORTED_IN_TEXT_VERIF", "CIGARS_IN_TEXT_VERIFICATION", "MATRIX_ROWS", "TEXT_BYTES"):
            assert dc[n] == oc[n], (trial, n, dc[n], oc[n])
            assert dc1[n] == oc[n], (trial, n, dc1[n], oc[n])
    assert dup_total > 1000


@pytest.mark.parametrize("spec,metric,partition,k", [
    ("multiple_opt", "edit", "dynamic", 4),
    ("multiple_opt", "edit", "dynamic", 2),
    ("multiple_opt", "edit", "uniform", 6),
    ("kuch1", "edit", "dynamic", 4),
    ("kuch1", "edit", "static", 3),
    ("kuch1", "edit", "uniform", 1),
    ("pigeon", "edit", "uniform", 4),
    ("pigeon", "edit", "dynamic", 2),
    ("kuch1", "hamming", "dynamic", 2),
    ("kuch1", "hamming", "static", 3),
    ("pigeon", "hamming", "uniform", 1),
    ("multiple_opt", "hamming", "dynamic", 4),
    ("kuch1", "edit", "dynamic", 0),
    # round 2: the remaining `-S` strategies (alignparameters.cpp:1341-1372)
    ("kuch2", "edit", "dynamic", 3),
    ("kuch2", "edit", "static", 4),
    ("kuch2", "hamming", "dynamic", 2),
    ("kianfar", "edit", "dynamic", 4),     # searches whose FIRST part already allows errors (U[0] > 0)
    ("kianfar", "edit", "static", 3),
    ("kianfar", "hamming", "uniform", 4),
    ("01*0", "edit", "dynamic", 2),
    ("01*0", "edit", "uniform", 4),
    ("minU", "edit", "dynamic", 5),
    ("minU", "edit", "uniform", 3),
    ("minU", "hamming", "dynamic", 6),
    ("columba", "edit", "dynamic", 4),     # the CLI's default strategy
    ("columba", "edit", "dynamic", 6),
    ("columba", "edit", "dynamic", 7),     # in-text verification beyond the reference's 64-bit matrix
    ("minU", "edit", "dynamic", 7),
    ("columba", "edit", "dynamic", 1),
    ("columba", "edit", "uniform", 5),
    ("columba", "hamming", "dynamic", 3),
])
def test_match_batch_parity(world, spec, metric, partition, k):
    # (kianfar's schemes start searches with errors allowed in the first part: ~20 000 nodes per read for the oracle)
    reads = synth.sample_reads(world["genome"], 500 if spec == "kianfar" else 3000, 150, seed=100 + k, n_frac=0.02)
    cnt = _compare(world, spec, metric, partition, k, reads)
    if k >= 2 and metric == "edit":
        assert cnt["SEARCH_STARTED"] > 0 and cnt["IN_TEXT_STARTED"] > 0  # both regimes exercised


@pytest.mark.parametrize("spec,partition,k,env", [
    ("multiple_opt", "dynamic", 4, {"CMB_MATRIX64": "1"}),      # the reference's 64-bit words in the frontier (GeoN)
    ("columba", "dynamic", 7, {"CMB_MATRIX64": "1"}),
    ("kianfar", "static", 3, {"CMB_MATRIX64": "1"}),
    ("columba", "dynamic", 6, {"CMB_TEST_NARROW_WV": "3"}),     # a phase "too wide" for the small matrix: the batch re-runs on GeoN
    ("multiple_opt", "uniform", 4, {"CMB_TEST_NARROW_WV": "0"}),
])
def test_in_index_matrix_geometries(world, monkeypatch, spec, partition, k, env):
    """Round 4: up to 7 errors the frontier carries the in-index matrix on 32-bit words with 8-row blocks (GeoN32) — every parity test above
    runs on it.  Here the same results on the reference's 64-bit words, and through the re-run a phase triggers whose first column does not
    fit the small matrix (the bound lowered by the test hook so that it happens)."""
    for n, v in env.items():
        monkeypatch.setenv(n, v)
    reads = synth.sample_reads(world["genome"], 400 if spec == "kianfar" else 2000, 150, seed=300 + k, n_frac=0.02)
    _compare(world, spec, "edit", partition, k, reads)


@pytest.mark.parametrize("name,dirname,mode,metric,partition,k", [
    ("kuch1", "kuch_k+1", "custom", "edit", "dynamic", 4),
    ("kuch1", "kuch_k+1", "custom", "edit", "static", 3),
    ("kuch2", "kuch_k+2", "custom", "hamming", "dynamic", 2),
    ("01*0", "01star0", "custom", "edit", "static", 3),
    ("kianfar", "kianfar", "custom", "edit", "dynamic", 4),
    ("multiple_opt", "multiple_opt", "multiple", "edit", "dynamic", 4),
    ("multiple_opt", "multiple_opt", "multiple", "edit", "uniform", 6),
])
def test_scheme_directories_equal_builtin_strategies(world, name, dirname, mode, metric, partition, k):
    """`-c <dir>` / `-d <dir>` on the reference's own search_schemes/ data (tests/golden/search_schemes) give the
    results AND counters of the hard-coded strategy of the same name (150 bp reads are beyond every k-mer cut-off:
    100 hard-coded, 50 custom, 20 multiple)."""
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_schemes", dirname)
    reads = synth.sample_reads(world["genome"], 500 if name == "kianfar" else 2000, 150, seed=300 + k, n_frac=0.02)
    dev = world["dev4" if name in ("kuch2", "01*0") else "dev"]
    o1, f1, c1 = ca.match_batch(dev, ca.SearchStrategy(name, metric, partition), k, reads)
    o2, f2, c2 = ca.match_batch(dev, ca.SearchStrategy.from_dir(d, mode, metric, partition), k, reads)
    assert len(o1) > 0 and np.array_equal(f1, f2) and np.array_equal(o1, o2)
    assert c1 == c2


def test_custom_directory_with_dynamic_selection(world):
    """`-c <dir>` as the CLI runs it (DynamicCustomStrategy: scheme + mirror image, dynamic selection) against the
    oracle driven by the same tables"""
    import os
    import schemes_py as sp
    op = world["op"]
    reads = synth.sample_reads(world["genome"], 2000, 150, seed=321, n_frac=0.02)
    for dirname, k in (("kuch_k+1", 4), ("kuch_k+2", 3), ("pigeon", 5)):
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_schemes", dirname)
        spec = sp.load_custom_dir(dirname, dynamic=True)
        o_occ, o_off, o_cnt = op.match_batch(world["orc"], op.OracleStrategy(spec, "edit", "dynamic"), k, reads, threads=8)
        d_occ, d_off, d_cnt = ca.match_batch(world["dev"], ca.SearchStrategy.from_dir(d, "custom_dynamic"), k, reads)
        assert np.array_equal(o_off, d_off)
        for f in ("begin", "end", "distance"):
            assert np.array_equal(o_occ[f], d_occ[f]), (dirname, f)
        for n in ("NODE_COUNTER", "IN_TEXT_STARTED", "SEARCH_STARTED", "EXPANSIONS", "MATRIX_ROWS"):
            assert o_cnt[n] == d_cnt[n], (dirname, n)
        # (net of the duplicates the reference's unstable sort lets through, see _compare)
        assert o_cnt["TOTAL_REPORTED_POSITIONS"] - o_cnt["SURVIVING_DUP_ROWS"] == d_cnt["TOTAL_REPORTED_POSITIONS"], dirname


def test_ragged_and_odd_reads(world):
    g = world["genome"]
    rng = np.random.default_rng(9)
    reads = []
    for ln in (36, 50, 75, 76, 77, 100, 101, 125, 151, 200, 250, 256):
        reads += synth.sample_reads(g, 40, ln, seed=ln, edit_choices=(0, 1, 2, 3))
    reads.append(b"N" * 100)
    reads.append(b"A" * 150)
    reads.append(b"ACGT" * 30)
    reads.append(b"acgtn" * 20 + g[5000:5100].tobytes().lower())
    reads.append(g[-151:-1].tobytes())           # touches the end of the text
    reads.append(g[0:150].tobytes())             # begins at text position 0
    _compare(world, "multiple_opt", "edit", "dynamic", 4, reads)
    _compare(world, "kuch1", "hamming", "dynamic", 3, reads)
    # the longest in-text matrices the device holds: 320 characters at k = 7 (341 rows, band of 29 columns)
    long_reads = [r for r in reads if len(r) >= 100]
    for ln in (250, 256, 257, 300, 320):
        long_reads += synth.sample_reads(g, 60, ln, seed=1000 + ln, edit_choices=(0, 3, 5, 7))
    _compare(world, "columba", "edit", "dynamic", 7, long_reads)
    _compare(world, "columba", "edit", "dynamic", 1, long_reads)   # two parts: phases of up to 300 rows
    _compare(world, "multiple_opt", "edit", "dynamic", 4, long_reads)


def test_reads_of_up_to_480_characters(world):
    """reads beyond 320 characters: contexts with the match words of 16 row blocks (dev_bfs_edit.hpp: ctxU4For), the long-read
    instances of k_parts / k_exact, longer verification windows; mixed with ordinary and with very short reads in one chunk"""
    g = world["genome"]
    reads = []
    for ln in (321, 352, 400, 401, 450, 480):
        reads += synth.sample_reads(g, 40, ln, seed=2000 + ln, n_frac=0.02, edit_choices=(0, 1, 3, 5, 7))
    reads += synth.sample_reads(g, 60, 150, seed=5, edit_choices=(0, 2, 4))
    reads += [g[-481:-1].tobytes(), g[0:480].tobytes(), b"ACGTA", b"ACG"]
    _compare(world, "multiple_opt", "edit", "dynamic", 4, reads, dups_rare=False)
    long_only = [r for r in reads if len(r) > 100]
    _compare(world, "columba", "edit", "dynamic", 7, long_only)
    _compare(world, "columba", "edit", "uniform", 1, long_only)    # two parts: phases of 240 rows
    _compare(world, "kuch1", "edit", "dynamic", 1, long_only)      # (dynamic partitioning: one part may take most of the read)
    _compare(world, "kuch1", "hamming", "static", 3, long_only)
    _compare(world, "pigeon", "edit", "dynamic", 0, long_only)


@pytest.mark.parametrize("partition,k,length", [("dynamic", 8, 150), ("uniform", 10, 150), ("static", 13, 200), ("dynamic", 13, 250),
                                                ("dynamic", 9, 100), ("dynamic", 11, 400)])
def test_hamming_distance_with_eight_to_thirteen_errors(world, partition, k, length):
    """`-S columba` beyond 7 errors: the greedy schemes of ColumbaSearchStrategy (searchstrategy.h:3396-3658; k + 1 parts, up to
    fourteen) on the wide instances of k_parts / k_exact / k_hbfs — Hamming distance (the edit-distance matcher stops at 7); a few
    reads not longer than the number of parts ride along (naive backtracking)"""
    g = world["genome"]
    reads = synth.sample_reads(g, 1200, length, seed=900 + k, n_frac=0.01, edit_choices=(0, 2, 5, 8, k, k, k + 1))
    reads += [g[5000:5000 + k + 1].tobytes(), g[77:80].tobytes(), b"N" * length, g[-length - 1:-1].tobytes()]
    _compare(world, "columba", "hamming", partition, k, reads, dups_rare=False)


@pytest.fixture(scope="module")
def world0(world):
    """the same text under an index that never switches to in-text verification (the reference's -i 0)"""
    op = world["op"]
    return {"genome": world["genome"], "ix": world["ix"], "op": op, "dev": ca.Index(world["ix"], in_text_switch=0),
            "orc": op.OracleIndex(world["ix"], switch_point=0)}


@pytest.mark.parametrize("partition,k,length", [("dynamic", 8, 150), ("uniform", 9, 150), ("static", 10, 200), ("dynamic", 10, 250),
                                                ("dynamic", 8, 100), ("dynamic", 11, 150), ("uniform", 12, 200), ("static", 13, 150),
                                                ("dynamic", 13, 250), ("dynamic", 12, 480)])
def test_edit_distance_with_eight_to_ten_errors_in_the_index(world0, partition, k, length):
    """Edit distance beyond 7 errors on an index with in-text switch point 0: the whole search stays in the index — the greedy
    schemes (9 ... 11 parts), the 64-bit in-index matrix (bitparallelmatrix.h:309-316: up to 10 errors), final-column packs of 32
    cells of 6 bits (GeoW), filter keys with 4 + 5 bits for distance and width.  Occurrences and counters equal to the oracle's."""
    g = world0["genome"]
    reads = synth.sample_reads(g, 500, length, seed=700 + k, n_frac=0.01, edit_choices=(0, 3, 6, 8, k, k, k + 1))
    reads += [b"N" * length, g[-length - 1:-1].tobytes(), g[0:length].tobytes()]
    _compare(world0, "columba", "edit", partition, k, reads, dups_rare=False)


@pytest.mark.parametrize("partition,k,length", [("dynamic", 8, 150), ("uniform", 9, 100), ("static", 10, 150), ("dynamic", 10, 250),
                                                ("dynamic", 11, 150), ("uniform", 12, 150), ("static", 13, 200), ("dynamic", 13, 400)])
def test_edit_distance_with_eight_to_ten_errors(world, partition, k, length):
    """... and on the default index (in-text switch point 4): candidates of up to 41 band columns are verified by k_wide_filter + k_verify_wide
    (the band of the reference's 128-bit matrix on 64-bit words with 16-row blocks and a left margin of 31 bits)"""
    g = world["genome"]
    reads = synth.sample_reads(g, 600, length, seed=300 + k, n_frac=0.01, edit_choices=(0, 3, 6, 8, k, k, k + 1))
    reads += [b"N" * length, g[-length - 1:-1].tobytes(), g[0:length].tobytes()]
    _compare(world, "columba", "edit", partition, k, reads, dups_rare=False)


def test_more_than_thirteen_errors_are_refused_up_front(world):
    with pytest.raises(ca.CmbError) as e:   # (MAX_K, definitions.h:50)
        ca.match_batch(world["dev"], ca.SearchStrategy("columba", "edit", "dynamic"), 14, [b"ACGT" * 40])
    assert e.value.code in (ca.CMB_ERR_UNSUPPORTED, ca.CMB_ERR_INVALID)
    with pytest.raises(ca.CmbError):
        ca.match_batch(world["dev"], ca.SearchStrategy("columba", "hamming", "dynamic"), 14, [b"ACGT" * 40])


@pytest.mark.parametrize("spec,k", [("columba", 7), ("columba", 5), ("multiple_opt", 6)])
def test_short_reads_with_many_errors(world, spec, k):
    """Reads of 40 ... 100 characters at 5 ... 7 errors: short parts, replays of long descendant lists that are interrupted
    in the final column (the events of those replays once carried a wrong context: lost alignments, and an out-of-range
    context index on larger references — found by tools/soak_parity.py, localised by tools/bounds_check.sh)."""
    g = world["genome"]
    reads = []
    for ln in (40, 48, 60, 75, 100):
        reads += synth.sample_reads(g, 1500, ln, seed=7000 + ln + k, n_frac=0.02, edit_choices=(0, 1, 2, 3, k - 1, k, k, k + 1))
    _compare(world, spec, "edit", "dynamic", k, reads)


def _edge_reads(g, n, k, seed):
    """Reads whose k edits are all indels packed at one end: their alignments run along the edges of the
    verification band (the cells the narrow trace rows of `k_traceback` do not store)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        pos = int(rng.integers(1000, len(g) - 1000))
        seg = g[pos:pos + 150 + k].tobytes()
        at = int(rng.integers(0, 6)) if i % 2 == 0 else 150 - k - int(rng.integers(0, 6))
        kind = (i // 2) % 3
        nd = k if kind == 0 else 0 if kind == 1 else k // 2      # deletions, the rest are insertions
        ni = k - nd
        ins = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), ni).tolist())
        r = seg[:at] + ins + seg[at + nd:]
        out.append(r[:150])
    return out


@pytest.mark.parametrize("k", [1, 2, 3, 4, 6, 7])
def test_band_edge_alignments(world, k):
    import os
    reads = _edge_reads(world["genome"], 1500 if k < 7 else 600, k, seed=40 + k)
    spec = "columba" if k == 7 else "multiple_opt" if k % 2 == 0 else "kuch1"
    _compare(world, spec, "edit", "dynamic", k, reads)
    if k <= 4:
        # the wide rows (used for k > 4) check, on every traceback step, the two rules the narrow rows rely on
        os.environ["CMB_TRACE_WIDE"] = "1"
        try:
            _compare(world, spec, "edit", "dynamic", k, reads)
        finally:
            del os.environ["CMB_TRACE_WIDE"]


def test_high_copy_repeat(oracle_built):
    """Reads from a 400-copy element: hundreds of occurrences per read — the long-segment path of the occurrence
    filter (`k_filter_mark`: segments of more than 24 / more than 64 keys) and wide SA ranges in `k_fmocc`."""
    import oracle_py as op
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    elem = rng.choice(acgt, 400)
    parts = []
    for c in range(400):
        parts.append(rng.choice(acgt, int(rng.integers(200, 900))))
        e = elem.copy()
        mut = rng.random(400) < 0.01
        e[mut] = rng.choice(acgt, int(mut.sum()))
        parts.append(e)
    g = np.concatenate(parts + [rng.choice(acgt, 1000)])
    ix = ib.build_index(g.tobytes(), device="cuda")
    w = {"genome": g, "ix": ix, "dev": ca.Index(ix), "orc": op.OracleIndex(ix), "op": op}
    reads = [elem[o:o + 150].tobytes() for o in range(0, 250, 5)]
    reads += synth.sample_reads(g, 200, 150, seed=3, edit_choices=(0, 1, 2))
    cnt = _compare(w, "multiple_opt", "edit", "dynamic", 4, reads)
    assert cnt["TOTAL_REPORTED_POSITIONS"] > 100 * 50
    _compare(w, "kuch1", "hamming", "dynamic", 2, reads)
    _compare(w, "kuch1", "edit", "dynamic", 0, reads)


def test_byte_text_path(world):
    """The matrix kernels read a 2-bit copy of the text; an index whose text holds other characters than ACGT
    before the '$' keeps the one-byte codes (forced here with CMB_TEXT_BYTES=1 at index creation)."""
    import os
    os.environ["CMB_TEXT_BYTES"] = "1"
    try:
        dev2 = ca.Index(world["ix"])
    finally:
        del os.environ["CMB_TEXT_BYTES"]
    reads = synth.sample_reads(world["genome"], 3000, 150, seed=104, n_frac=0.02)
    for spec, k in (("multiple_opt", 4), ("multiple_opt", 6)):
        st = ca.SearchStrategy(spec, "edit", "dynamic")
        o1, f1, c1 = ca.match_batch(world["dev"], st, k, reads)
        o2, f2, c2 = ca.match_batch(dev2, st, k, reads)
        assert np.array_equal(f1, f2) and np.array_equal(o1, o2)
        assert c1 == c2


@pytest.mark.parametrize("sparseness,switch,kmer", [(1, 4, 10), (8, 0, 10), (32, 1, 8), (4, 10, 12), (16, 50, 4), (2, 7, 6), (4, 4, 13)])
def test_index_parameters(world, sparseness, switch, kmer):
    """Suffix-array sparseness, in-text switch point and k-mer size of the seed table away from their defaults (4, 4, 10;
    tools/soak_index_params.py runs the full grid at scale)."""
    import schemes_py as sp
    op = world["op"]
    g = world["genome"][:600_000]
    ix = ib.build_index(g.tobytes(), sparseness=sparseness, device="cuda")
    dev, orc = ca.Index(ix, in_text_switch=switch, kmer_size=kmer), op.OracleIndex(ix, switch_point=switch, kmer_size=kmer)
    w = {"dev": dev, "orc": orc, "op": op}
    reads = []
    for ln in (50, 100, 150):
        reads += synth.sample_reads(g, 500, ln, seed=sparseness + switch + ln, n_frac=0.02, edit_choices=(0, 1, 2, 3, 4, 5))
    _compare(w, "multiple_opt", "edit", "dynamic", 4, reads, dups_rare=False)
    _compare(w, "kuch1", "hamming", "dynamic", 3, reads, dups_rare=False)
    _compare(w, "pigeon", "edit", "uniform", 2, reads, dups_rare=False)
    dev.close()


def test_inconsistent_index_arrays_are_refused(world):
    """Arrays that do not belong together (here: a sparse suffix array of sparseness 8 declared as 4) would make the LF
    walk of the locate endless; cmb_index_create probes for it and refuses."""
    import dataclasses
    g = world["genome"][:300_000]
    ix8 = ib.build_index(g.tobytes(), sparseness=8, device="cuda")
    ca.Index(ix8).close()   # (consistent: accepted)
    with pytest.raises(ca.CmbError) as e:
        ca.Index(dataclasses.replace(ix8, sparseness=4))
    assert e.value.code == -1 and "do not belong together" in str(e.value)


def test_errors_are_loud(world):
    st = ca.SearchStrategy("multiple_opt")
    with pytest.raises(ca.CmbError) as e:   # distance without scheme
        ca.match_batch(world["dev"], st, 3, [b"ACGT" * 30])
    with pytest.raises(ca.CmbError) as e:   # beyond the in-text matrix of the device (k <= 7) / without a scheme
        ca.match_batch(world["dev"], ca.SearchStrategy("pigeon"), 8, [b"ACGT" * 30])
    with pytest.raises(ca.CmbError):
        ca.match_batch(world["dev"], st, 4, [b"A" * 481])   # (reads of up to 480 characters are supported)
    with pytest.raises(ca.CmbError) as e:   # seeds placed for 4-mers on an index with a 10-mer table
        ca.match_batch(world["dev"], ca.SearchStrategy("01*0", "edit", "dynamic"), 2, [b"ACGT" * 37 + b"AC"])
    assert e.value.code == -1 and "seeds of a read overlap" in str(e.value)
    # empty batch is fine
    occ, offs, _ = ca.match_batch(world["dev"], st, 4, [])
    assert len(occ) == 0 and offs.tolist() == [0]


@pytest.mark.parametrize("spec,metric,k", [("multiple_opt", "edit", 4), ("columba", "edit", 6), ("kuch1", "edit", 2), ("columba", "edit", 9),
                                           ("kuch1", "hamming", 3), ("kuch1", "edit", 0), ("columba", "edit", 12)])
def test_alignments_cigar_and_sequence(world, oracle_built, spec, metric, k):
    """SURVEY.md §8f rank 1 on the device: the CIGAR of every final occurrence (k_cigar) is what the reference's
    findCIGAR gives for (read on its strand, text[begin, end), distance) — asked of the oracle's restatement, which is
    pinned to the reference's findCIGAR by the golden vectors — and the sequence assignment is findSeqName's."""
    import os
    import subprocess
    g = world["genome"]
    reads = synth.sample_reads(g, 1500, 150, seed=700 + k, n_frac=0.01)
    # reads across sequence boundaries and with edits at their very ends
    starts = np.asarray(world["ix"].seq_starts, dtype=np.int64)
    for s in starts[1:-1][:20]:
        reads.append(g[int(s) - 70:int(s) + 80].tobytes())
    reads += _edge_reads(g, 200, max(k, 1), seed=71)
    b = ca.Batch(world["dev"], ca.SearchStrategy(spec, metric, "dynamic"), k, reads)
    b.want_alignments()
    b.run()
    occ, offs, _ = b.results()
    aln, ops = b.alignments()
    assert len(aln) == len(occ) > (1000 if k else 300)
    text = g.tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    cmds, who = [], []
    for i, r in enumerate(reads):
        fw = bytes(c if c in b"ACGT" else ord("N") for c in r.upper())
        rc = fw.translate(comp)[::-1]
        for j in range(int(offs[i]), int(offs[i + 1])):
            o = occ[j]
            seq = rc if o["strand"] else fw
            if metric == "edit" and k > 0:
                cmds.append(f"findcigar {seq.decode()} {text[int(o['begin']):int(o['end'])].decode()} {int(o['distance'])}")
                who.append(j)
            else:   # Hamming / exact occurrences: no gaps
                assert ca.cigar_string(ops[int(aln[j]['cigar_off']):int(aln[j]['cigar_off']) + int(aln[j]['cigar_len'])]) == f"{len(r)}M"
    if cmds:
        res = subprocess.run([os.path.join(oracle_built, "oracle_driver")], input="\n".join(cmds) + "\n", capture_output=True,
                             text=True, check=True).stdout.splitlines()
        assert len(res) == len(cmds)
        gaps = 0
        for j, want in zip(who, res):
            a = aln[j]
            got = ca.cigar_string(ops[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])])
            assert got == want, (j, occ[j], got, want)
            gaps += ("I" in got) or ("D" in got)
        assert gaps > 20 or k < 2
    # IndexInterface::findSeqName (indexinterface.cpp:799-832)
    idx = np.searchsorted(starts, occ["begin"].astype(np.int64), side="right") - 1
    assert np.array_equal(aln["seq_id"], idx.astype(np.uint32))
    assert np.array_equal(aln["seq_begin"], (occ["begin"].astype(np.int64) - starts[idx]).astype(np.uint32))
    assert np.array_equal(aln["spans"] != 0, occ["end"].astype(np.int64) > starts[idx + 1])
    assert (aln["spans"] != 0).sum() > 0 or k == 0
    # the occurrences themselves are what a run without alignments gives
    o2, f2, _ = ca.match_batch(world["dev"], ca.SearchStrategy(spec, metric, "dynamic"), k, reads)
    assert np.array_equal(o2, occ) and np.array_equal(f2, offs)
    b.close()


@pytest.mark.parametrize("spec,metric,x,min_identity", [("columba", "edit", 0, 96), ("columba", "edit", 1, 96),
                                                        ("multiple_opt", "edit", 0, 97), ("kuch1", "hamming", 0, 98),
                                                        ("minU", "edit", 2, 97), ("columba", "edit", 0, 95),
                                                        ("columba", "hamming", 0, 91), ("columba", "hamming", 1, 90),
                                                        ("columba", "edit", 0, 93), ("kuch1", "edit", 0, 50), ("columba", "edit", 0, 91)])
def test_best_mode(world, spec, metric, x, min_identity):
    """BEST (+x strata) mode — the reference's default (`-a best`, SearchStrategy::matchApproxBestPlusX,
    searchstrategy.cpp:623-746): per read the best distance, the number of hits at it, and the alignments of the best
    x + 1 strata in the reference's order, with sequence assignment and CIGAR, against the oracle's restatement.
    (Cut-off: min(13, what strategy and device support, len * (100 - identity) / 100); the device supports 7 errors —
    all the columba strategy has schemes for — and the oracle is given the same limit.)"""
    import schemes_py as sp
    op = world["op"]
    g = world["genome"]
    reads = synth.sample_reads(g, 2500, 150, seed=800 + x, n_frac=0.01, edit_choices=(0, 0, 1, 2, 3, 5, 6, 9) + ((11, 13) if min_identity < 95 else ()))
    starts = np.asarray(world["ix"].seq_starts, dtype=np.int64)
    for s in starts[1:-1][:12]:   # reads across sequence boundaries: trimmed or dropped (findSeqName)
        reads.append(g[int(s) - 75:int(s) + 75].tobytes())
        reads.append(g[int(s) - 3:int(s) + 147].tobytes())
        reads.append(g[int(s) - 147:int(s) + 3].tobytes())
    reads += [b"ACGT" * 37 + b"AC", b"N" * 150]
    if min_identity == 50:   # strata of reads not longer than the number of parts: naive backtracking inside a stratum's batch
        reads = reads[:600] + [b"A", b"AC", b"ACG", b"ACGTA", b"GATTACA", b""]
    if (metric, min_identity) == ("edit", 91):   # strata up to 13 errors: a smaller chunk (the oracle walks them on the CPU)
        reads = reads[:160] + reads[2500:]
    spec_tables = sp.BY_NAME[spec]
    max_sup = 0
    while (max_sup + 1) in spec_tables["schemes"]:
        max_sup += 1
    max_sup = min(max_sup, 13)   # (MAX_K)
    o_occ, o_sid, o_sb, o_cig, o_off, o_best, o_hits, o_cnt = op.match_best(
        world["orc"], op.OracleStrategy(spec_tables, metric, "dynamic"), reads, x=x, min_identity=min_identity,
        max_supported=max_sup, threads=8)
    d_occ, d_aln, d_ops, d_off, d_best, d_hits, d_cnt = ca.match_best(
        world["dev"], ca.SearchStrategy(spec, metric, "dynamic"), reads, x=x, min_identity=min_identity)
    assert np.array_equal(o_best, d_best)
    # (multiple_opt has no scheme for 1 error: its best mode stops at exact matches, searchstrategy.h:2744-2750)
    # (with x > 0 the reference never looks at stratum 0 — its loop over the strata to check starts at prevK + 1 = 1,
    # searchstrategy.cpp:688 — so reads that only match exactly stay unmapped there; Hamming cut-off 3 at 98 %)
    few = (metric, min_identity) == ("edit", 91)
    assert (o_best != 0xFFFFFFFF).sum() > (100 if few else 1500 if (spec, x, metric) == ("columba", 0, "edit") else 300) and (few or (o_best == 0xFFFFFFFF).sum() > 0)
    assert np.array_equal(o_hits, d_hits)
    assert np.array_equal(o_off, d_off)
    for f in ("begin", "end", "distance", "strand"):
        assert np.array_equal(o_occ[f], d_occ[f]), f
    assert np.array_equal(o_sid, d_aln["seq_id"]) and np.array_equal(o_sb, d_aln["seq_begin"])
    for j in range(len(d_occ)):
        a = d_aln[j]
        got = ca.cigar_string(d_ops[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])])
        assert got == o_cig[j], (j, d_occ[j], got, o_cig[j])
    for n in ("NODE_COUNTER", "IN_TEXT_STARTED", "SEARCH_STARTED", "EXPANSIONS", "IMMEDIATE_SWITCH"):
        assert o_cnt[n] == d_cnt[n], (n, o_cnt[n], d_cnt[n])


@pytest.mark.parametrize("spec,metric,k,xa", [("columba", "edit", 4, False), ("multiple_opt", "edit", 2, True),
                                              ("kuch1", "hamming", 2, False), ("kuch1", "edit", 0, False),
                                              ("columba", "edit", 9, False), ("columba", "edit", 13, False)])
def test_sam_records_of_a_chunk(world, spec, metric, k, xa):
    """The SAM text of a chunk in ALL mode (cmb_batch_sam = generateOutputSingleEnd + generateSE_SAM[_XATag]) against
    the oracle's restatement: sequence names, 1-based positions, flags, mapping qualities, CIGARs, the read as it aligns
    (reverse complement and reversed quality on the other strand), secondary lines / XA tag, unmapped records, and
    occurrences trimmed at sequence ends."""
    import schemes_py as sp
    op = world["op"]
    g = world["genome"]
    rng = np.random.default_rng(5 + k)
    reads = synth.sample_reads(g, 600, 150, seed=950 + k, n_frac=0.01, edit_choices=(0, 1, 2, 4, 9))
    starts = np.asarray(world["ix"].seq_starts, dtype=np.int64)
    for s in starts[1:-1][:10]:
        reads.append(g[int(s) - 75:int(s) + 75].tobytes())
        reads.append(g[int(s) - 2:int(s) + 148].tobytes())
    ids = [("@" if i % 2 else ">") + f"read{i}/1 some description" for i in range(len(reads))]
    quals = ["".join(chr(33 + int(q)) for q in rng.integers(0, 41, len(r))) for r in reads]
    names = [f"chr{j + 1}" for j in range(len(starts) - 1)]
    want = op.match_batch_sam(world["orc"], op.OracleStrategy(sp.BY_NAME[spec], metric, "dynamic"), k, reads, ids, quals,
                              names, unmapped=True, xa=xa)
    b = ca.Batch(world["dev"], ca.SearchStrategy(spec, metric, "dynamic"), k, reads)
    b.want_alignments()
    b.run()
    got = b.sam(ids, quals, names, unmapped=True, xa=xa)
    b.close()
    gl, wl = got.splitlines(), want.splitlines()
    assert len(wl) > 600
    # k = 0: the reference reports exact matches in suffix-array order, forward strand then reverse complement
    # (searchstrategy.cpp:499-510), and its primary record is the first of them; the device returns them sorted by
    # position, so WHICH of several exact matches is the primary one may differ: compared there are the records
    # without what depends on that choice (the secondary flag, and the sequence / quality only the primary shows)
    if k == 0:
        def norm(lines):
            out = []
            for x in lines:
                f = x.split("\t")
                f[1] = str(int(f[1]) & ~256)
                f[9] = f[10] = "."
                out.append("\t".join(f))
            return sorted(out)
        gl, wl = norm(gl), norm(wl)
    for a, w in zip(gl, wl):
        assert a == w
    assert len(gl) == len(wl)
    assert any("\t4\t*\t0\t0\t*" in x for x in wl) and any("\t16\t" in x or "\t272\t" in x for x in wl)


def test_paired_end_chunk_end_to_end(world):
    """read pairs sampled as FR fragments: both mates through the GPU matcher, pairing and SAM text by cmb_pair_sam — every
    pair whose mates carry few errors comes back as a proper pair at the fragment it was cut from (the pairing logic itself is
    tested on the CPU, tests/test_pairing.py)"""
    g = world["genome"]
    rng = np.random.default_rng(77)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    n, L = 300, 100
    reads1, reads2, truth = [], [], []
    for i in range(n):
        frag = int(rng.integers(220, 420))
        p0 = int(rng.integers(1000, len(g) - 1000))
        f = bytearray(g[p0:p0 + frag].tobytes())
        m1, m2 = bytearray(f[:L]), bytearray(bytes(f[-L:]).translate(comp)[::-1])
        for m in (m1, m2):
            for _ in range(int(rng.integers(0, 3))):
                q = int(rng.integers(1, L - 1))
                m[q] = b"ACGT"[(b"ACGT".index(bytes([m[q]])) + 1) % 4] if bytes([m[q]]) in b"ACGT" else m[q]
        if i % 2:  # the fragment from the other strand: mate 1 is the reverse-complement end
            m1, m2 = m2, m1
        reads1.append(bytes(m1))
        reads2.append(bytes(m2))
        truth.append((p0, frag))
    ids1 = [f"@frag{i}/1" for i in range(n)]
    ids2 = [f"@frag{i}/2" for i in range(n)]
    quals = ["I" * L] * n
    strat = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
    names = [f"seq{i}" for i in range(len(world["dev"].seq_starts()) - 1)] or ["seq0"]
    text, mapped = ca.pair_chunk_sam(world["dev"], strat, 2, reads1, reads2, ids1, ids2, quals, quals, names, ca.ORIENTATION_FR, 600, 100,
                                     True, True)
    lines = [ln.split("\t") for ln in text.splitlines()]
    by_pair = {}
    for f in lines:
        by_pair.setdefault(f[0].split("/")[0], []).append(f)
    assert len(by_pair) == n and mapped >= 0.9 * n
    starts = world["dev"].seq_starts()
    right = 0
    for i in range(n):
        recs = by_pair[f"frag{i}"]
        prim = [f for f in recs if int(f[1]) & 2 and not int(f[1]) & 256]
        if len(prim) == 2:
            p0, frag = truth[i]
            sid = int(np.searchsorted(starts, p0, side="right") - 1)
            fwd = [f for f in prim if not int(f[1]) & 16][0]
            right += names[sid] == fwd[2] and abs(int(fwd[3]) - 1 - (p0 - int(starts[sid]))) <= 2 and abs(abs(int(fwd[8])) - frag) <= 4
    assert right >= 0.85 * n, right
    # the strands of a mate filtered each by itself (matchApproxPairedEndAll's mapRead): the same proper pairs here
    text2, mapped2 = ca.pair_chunk_sam(world["dev"], strat, 2, reads1, reads2, ids1, ids2, quals, quals, names, ca.ORIENTATION_FR, 600, 100,
                                       True, True, per_strand=False)  # (the joint filter of pairSingleEndedMatchesAll's view)
    prop = lambda t: sorted(ln for ln in t.splitlines() if int(ln.split("\t")[1]) & 2 and not int(ln.split("\t")[1]) & 256)
    assert mapped2 == mapped and prop(text2) == prop(text)


def test_paired_end_best_mode_chunk_end_to_end(world):
    """read pairs in BEST mode (matchApproxPairedEndBestPlusX): the chunk walks through its strata together, one device batch per mate and
    distance asked for (ca.pair_chunk_sam_best over cmb_pair_best_*).  Fragments with few errors come back as proper pairs where they were cut;
    the proper pairs are the ALL-mode pairs (distance = the cut-off) of the smallest total distance; a mate that maps nowhere leaves its
    partner with the unmapped-mate records; a pair too far apart is reported discordant.  The walk itself is tested on the CPU
    (tests/test_pairing_best.py)."""
    g = world["genome"]
    rng = np.random.default_rng(78)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    n, L = 240, 100
    reads1, reads2, truth = [], [], []
    for i in range(n):
        frag = int(rng.integers(220, 420))
        p0 = int(rng.integers(1000, len(g) - 3000))
        p1 = p0 + frag - L if i % 20 != 7 else p0 + 1500  # (every twentieth pair: too far apart for a proper pair)
        m1, m2 = bytearray(g[p0:p0 + L].tobytes()), bytearray(g[p1:p1 + L].tobytes().translate(comp)[::-1])
        for m, lim in ((m1, 3), (m2, 6 if i % 5 == 0 else 3)):
            for _ in range(int(rng.integers(0, lim))):
                q = int(rng.integers(1, L - 1))
                m[q] = b"ACGT"[(b"ACGT".index(bytes([m[q]])) + 1) % 4] if bytes([m[q]]) in b"ACGT" else m[q]
        if i % 20 == 13:
            m2 = bytearray(rng.choice(np.frombuffer(b"ACGT", np.uint8), L).tobytes())  # a mate from nowhere
        if i % 2:
            m1, m2 = m2, m1
        reads1.append(bytes(m1))
        reads2.append(bytes(m2))
        truth.append((p0, p1 + L - p0))
    ids1 = [f"@frag{i}/1" for i in range(n)]
    ids2 = [f"@frag{i}/2" for i in range(n)]
    quals = ["I" * L] * n
    strat = ca.SearchStrategy("columba", "edit", "dynamic")
    starts = world["dev"].seq_starts()
    names = [f"seq{i}" for i in range(len(starts) - 1)] or ["seq0"]
    text, mapped, batches = ca.pair_chunk_sam_best(world["dev"], strat, reads1, reads2, ids1, ids2, quals, quals, names, x=0, min_identity=95,
                                                   orientation=ca.ORIENTATION_FR, max_frag=600, min_frag=100)
    assert batches <= 80, batches  # (two mates x the distances 0 .. 5 x a few rounds — not one search per pair and stratum: 48 here)
    by_pair = {}
    for ln in text.splitlines():
        f = ln.split("\t")
        by_pair.setdefault(f[0].split("/")[0], []).append(f)
    assert len(by_pair) == n
    text_all, _ = ca.pair_chunk_sam(world["dev"], strat, 5, reads1, reads2, ids1, ids2, quals, quals, names, ca.ORIENTATION_FR, 600, 100, True, True)
    all_pairs = {}
    for ln in text_all.splitlines():
        f = ln.split("\t")
        all_pairs.setdefault(f[0].split("/")[0], []).append(f)

    def proper(recs):  # {(sequence, forward mate's position, template length)} -> total distance, over the records of proper pairs
        out = {}
        for f in recs:
            if int(f[1]) & 2:
                key = (f[2], min(int(f[3]), int(f[7])), abs(int(f[8])))
                out[key] = out.get(key, 0) + int([t for t in f[11:] if t.startswith("NM:i:")][0][5:])
        return out

    right = same = with_pair = half_mapped = 0
    for i in range(n):
        recs = by_pair[f"frag{i}"]
        flags = [int(f[1]) for f in recs]
        if i % 20 == 13:  # the mate from nowhere is unmapped; its partner's records carry "mate unmapped" — or the partner is reported unmapped as
            # well: after pairDiscordantlyBest has looked at every stratum by itself (mapStratum), findBestAlignments only asks the strata
            # 0, 1, 3, 5, ... whether THEY hold something (hasUpdate, searchstrategy.cpp:674-681), so a partner whose best hit has 2 or 4
            # errors is not seen.  The reference's behaviour, kept.
            assert any(fl & 4 for fl in flags) and not any(fl & 2 for fl in flags), (i, flags)
            half_mapped += any(fl & 8 and not fl & 4 for fl in flags)
            continue
        if i % 20 == 7:
            assert not any(fl & 2 for fl in flags) and all(fl & 1 for fl in flags), (i, flags)
            continue
        best = proper(recs)
        ref = proper(all_pairs[f"frag{i}"])
        if ref:
            with_pair += 1
            lo = min(ref.values())
            same += best == {k: v for k, v in ref.items() if v == lo}
        p0, frag = truth[i]
        sid = int(np.searchsorted(starts, p0, side="right") - 1)
        right += any(k[0] == names[sid] and abs(k[1] - 1 - (p0 - int(starts[sid]))) <= 2 and abs(k[2] - frag) <= 4 for k in best)
    assert right >= 0.85 * n * 0.9 and with_pair > 150 and same >= 0.97 * with_pair and half_mapped >= 4, (right, with_pair, same, half_mapped)
    # the strata above the best one (x = 1) and the other orientations run through the same walk
    t1, m1, _ = ca.pair_chunk_sam_best(world["dev"], strat, reads1[:60], reads2[:60], ids1[:60], ids2[:60], quals[:60], quals[:60], names, x=1,
                                       min_identity=95, orientation=ca.ORIENTATION_FR, max_frag=600, min_frag=100)
    assert m1 >= 45 and t1.count("\n") >= 120
    t2, m2, _ = ca.pair_chunk_sam_best(world["dev"], strat, reads1[:60], reads2[:60], ids1[:60], ids2[:60], quals[:60], quals[:60], names, x=0,
                                       min_identity=95, orientation=ca.ORIENTATION_FF, max_frag=600, min_frag=100)
    assert not any(int(ln.split("\t")[1]) & 2 for ln in t2.splitlines()) and m2 >= 45  # (FR fragments under FF: discordant pairs)


def test_paired_mates_across_sequence_boundaries_are_trimmed(world):
    """A mate whose only hit runs over the end of its sequence: assignSequence -> findSeqName trims it (indexinterface.cpp:833-899,
    FOUND_WITH_TRIMMING: new range, distance, CIGAR) and the pair forms with the trimmed mate — the same record the single-end
    path (cmb_batch_sam) writes for that read; it is not reported unmapped."""
    g = world["genome"]
    starts = world["dev"].seq_starts()
    assert len(starts) >= 3
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    names = [f"seq{i}" for i in range(len(starts) - 1)]
    L, k = 100, 2
    reads1, reads2, want = [], [], []
    for j, over in enumerate((1, 2, 2, 1)):          # characters of mate 2's window that lie beyond the boundary
        b = int(starts[1 + j % (len(starts) - 2)])   # a boundary between two sequences
        frag_end = b + over                          # the fragment ends `over` characters into the next sequence
        p0 = frag_end - 300
        f = g[p0:frag_end].tobytes()
        reads1.append(f[:L])
        reads2.append(f[-L:].translate(comp)[::-1])
        want.append((int(np.searchsorted(starts, p0, side="right") - 1), p0, over))
    ids1 = [f"@s{i}/1" for i in range(len(reads1))]
    ids2 = [f"@s{i}/2" for i in range(len(reads1))]
    quals = ["I" * L] * len(reads1)
    strat = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
    text, mapped = ca.pair_chunk_sam(world["dev"], strat, k, reads1, reads2, ids1, ids2, quals, quals, names, ca.ORIENTATION_FR, 600, 100, True, True)
    assert mapped == len(reads1), text
    # the single-end records of the mates that cross the boundary (trimmed by cmb_batch_sam)
    b2 = ca.Batch(world["dev"], strat, k, reads2)
    b2.want_alignments()
    b2.run()
    se = [ln.split("\t") for ln in b2.sam([i[1:] for i in ids2], quals, names).splitlines() if not int(ln.split("\t")[1]) & 256]
    assert len(se) == len(reads2)  # (one primary record per read, in read order)
    b2.close()
    for i, (sid, p0, over) in enumerate(want):
        recs = [ln.split("\t") for ln in text.splitlines() if ln.split("\t")[0].split("/")[0] == f"s{i}"]
        prim = [f for f in recs if int(f[1]) & 2 and not int(f[1]) & 256]
        assert len(prim) == 2, recs
        m2 = [f for f in prim if int(f[1]) & 128][0]
        s2 = se[i]
        assert (m2[2], m2[3], m2[5]) == (s2[2], s2[3], s2[5]), (m2, s2)   # sequence, position and CIGAR of the trimmed mate
        assert m2[2] == names[sid] and "S" not in m2[5]


def test_cigars_along_the_right_edge_of_the_band(world, oracle_built):
    """findCIGAR (cmb_cigar_windows) of alignments that run Wh = distance columns right of the diagonal — leading and inner
    insertions of up to 13 characters: beyond 9 errors the match words of 32-row blocks do not reach the band's right edge in a block's
    last rows, and k_cigar_wide takes them from the read's bit-strings instead (a trimmed occurrence with eleven leading insertions lost
    its SAM record before).  Against the oracle's findCIGAR on the reference's matrices (64 bits up to 10 errors, 128 beyond)."""
    import ctypes as C
    import os
    import subprocess
    g = world["genome"]
    L = ca.lib()
    base = g[700_000:700_200].tobytes()
    ins = b"ACGTACGTACGTACG"
    cases = []
    for n_ins in (6, 9, 10, 11, 12, 13):
        for d in range(max(n_ins, 8), 14):
            cases.append((ins[:n_ins] + base[:150 - n_ins], 700_000, 700_000 + 150 - n_ins, d))
            cases.append((base[:70] + ins[:n_ins] + base[70:150 - n_ins], 700_000, 700_000 + 150 - n_ins, d))
            cases.append((base[:150 - n_ins] + ins[:n_ins], 700_000, 700_000 + 150 - n_ins, d))
            cases.append((base[:60] + base[60 + n_ins:150 + n_ins], 700_000, 700_000 + 150 + n_ins, d))
    cmds, got = [], []
    for read, b, e, d in cases:
        bb, ee, dd = np.array([b], np.uint32), np.array([e], np.uint32), np.array([d], np.uint32)
        ops, nops = np.zeros(40, np.uint16), np.zeros(1, np.uint32)
        rc = L.cmb_cigar_windows(world["dev"].h, read, len(read), bb.ctypes.data_as(C.c_void_p), ee.ctypes.data_as(C.c_void_p),
                                 dd.ctypes.data_as(C.c_void_p), 1, ops.ctypes.data_as(C.c_void_p), 40, nops.ctypes.data_as(C.c_void_p))
        assert rc == 0, (read, d, L.cmb_last_error())
        got.append(ca.cigar_string(ops[:int(nops[0])]))
        cmds.append(f"findcigar {read.decode()} {g[b:e].tobytes().decode()} {d}")
    res = subprocess.run([os.path.join(oracle_built, "oracle_driver")], input="\n".join(cmds) + "\n", capture_output=True, text=True,
                         check=True).stdout.splitlines()
    assert res == got
    assert sum("I" in c for c in got) > 50 and sum("D" in c for c in got) > 20
