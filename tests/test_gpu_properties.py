"""Size-independent properties of the HIP path at sizes the CPU oracle would not finish quickly
(the domain's invariants stand in for a full-size oracle run):

* batching invariance — the occurrences of a read do not depend on which other reads share its batch;
* strand symmetry — a read and its reverse complement have the same occurrences on opposite strands;
* the growth path of the frontier pools (too-small pools are enlarged and the search re-runs) gives the
  same result as pools that were large enough from the start.
"""
import os

import numpy as np
import pytest

import columba_amd as ca
from columba_amd import indexbuild as ib
from columba_amd import synth

pytestmark = pytest.mark.gpu

N_READS = 120_000


@pytest.fixture(scope="module")
def big():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    g, starts = synth.genome_human_like(48_000_000, seed=77, device="cuda")
    ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
    dev = ca.Index(ix)
    buf, _ = synth.sample_reads_fast(torch.from_numpy(ix.text[:-1]).cuda(), N_READS, 150, seed=5, device="cuda")
    reads = [buf[i * 150:(i + 1) * 150].tobytes() for i in range(N_READS)]
    return {"dev": dev, "reads": reads}


def _run(dev, reads, k=4, spec="multiple_opt", part="dynamic", metric="edit"):
    return ca.match_batch(dev, ca.SearchStrategy(spec, metric, part), k, reads)


def _per_read(occ, off, i):
    return [(int(o["begin"]), int(o["end"]), int(o["distance"]), int(o["strand"])) for o in occ[int(off[i]):int(off[i + 1])]]


COUNTERS = ["NODE_COUNTER", "IN_TEXT_STARTED", "ABORTED_IN_TEXT_VERIF", "CIGARS_IN_TEXT_VERIFICATION", "IMMEDIATE_SWITCH",
            "SEARCH_STARTED", "EXPANSIONS", "DFS_EXPANSIONS", "LF_STEPS", "LOCATED_ROWS", "TEXT_BYTES", "MATRIX_ROWS",
            "TOTAL_REPORTED_POSITIONS"]


def test_batching_invariance(big):
    occ, off, cnt = _run(big["dev"], big["reads"])
    assert len(occ) > N_READS // 2
    q = N_READS // 4
    tot = {n: 0 for n in COUNTERS}
    for j in range(4):
        o2, f2, c2 = _run(big["dev"], big["reads"][j * q:(j + 1) * q])
        assert np.array_equal(f2 - f2[0], off[j * q:(j + 1) * q + 1] - off[j * q])
        a = occ[int(off[j * q]):int(off[(j + 1) * q])]
        for f in ("begin", "end", "distance", "strand"):
            assert np.array_equal(o2[f], a[f]), (j, f)
        for n in COUNTERS:
            tot[n] += c2[n]
    for n in COUNTERS:  # the work counters are sums over reads
        assert tot[n] == cnt[n], (n, tot[n], cnt[n])


def test_strand_symmetry(big):
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    sub = big["reads"][:20_000]
    rc = [r.translate(comp)[::-1] for r in sub]
    o1, f1, _ = _run(big["dev"], sub)
    o2, f2, _ = _run(big["dev"], rc)
    assert np.array_equal(f1, f2)
    for f in ("begin", "end", "distance"):
        assert np.array_equal(o1[f], o2[f]), f
    # strands are mirrored, except where one (begin, end, distance) exists on both strands (the device then
    # reports the forward strand)
    flipped = o1["strand"] != o2["strand"]
    assert flipped.mean() > 0.99


@pytest.mark.parametrize("cfg", [dict(), dict(k=2, spec="kuch1", part="uniform", metric="hamming")],
                         ids=["edit-k4-multiple_opt", "hamming-k2-kuch1"])
def test_pool_growth_path(big, cfg):
    sub = big["reads"][:30_000]
    o1, f1, c1 = _run(big["dev"], sub, **cfg)
    os.environ["CMB_TEST_SMALL_POOLS"] = "1"
    try:
        o2, f2, c2 = _run(big["dev"], sub, **cfg)
    finally:
        del os.environ["CMB_TEST_SMALL_POOLS"]
    assert np.array_equal(f1, f2)
    for f in ("begin", "end", "distance", "strand"):
        assert np.array_equal(o1[f], o2[f]), f
    for n in COUNTERS:
        assert c1[n] == c2[n], n


def test_composite_batch(big):
    """A batch split into concurrent sub-batches (CMB_SUBBATCHES; automatic from 10^6 reads) returns the same
    per-read lists and the same counters as the same batch run in one piece."""
    sub = big["reads"][:50_001]
    o1, f1, c1 = _run(big["dev"], sub)
    os.environ["CMB_SUBBATCHES"] = "3"
    try:
        o2, f2, c2 = _run(big["dev"], sub)
    finally:
        del os.environ["CMB_SUBBATCHES"]
    assert np.array_equal(f1, f2)
    for f in ("begin", "end", "distance", "strand"):
        assert np.array_equal(o1[f], o2[f]), f
    for n in COUNTERS:
        assert c1[n] == c2[n], n


@pytest.mark.parametrize("env", ["CMB_TRACE_WIDE", "CMB_MATRIX_WIDE"])
@pytest.mark.parametrize("k", [2, 4])
def test_narrow_and_wide_matrix_formats(big, k, env):
    """For k <= 4 the in-text verification runs the banded matrix on 32-bit words / 8-row blocks
    (CMB_MATRIX_WIDE=1: the reference's 64-bit words / 32-row blocks, used for k > 4), and `k_traceback` keeps
    16 + 16 bits per matrix row (two band-edge bits are implied; CMB_TRACE_WIDE=1: 32 + 32 bits, whose kernel also
    checks the implied-bit rules on every step — a violation fails the batch).  Results and counters must agree."""
    sub = big["reads"][:60_000]
    spec = "multiple_opt" if k == 4 else "kuch1"
    o1, f1, c1 = _run(big["dev"], sub, k=k, spec=spec)
    os.environ[env] = "1"
    try:
        o2, f2, c2 = _run(big["dev"], sub, k=k, spec=spec)
    finally:
        del os.environ[env]
    assert np.array_equal(f1, f2)
    for f in ("begin", "end", "distance", "strand"):
        assert np.array_equal(o1[f], o2[f]), f
    for n in COUNTERS:
        assert c1[n] == c2[n], n


def test_one_index_many_host_threads(big):
    """SURVEY.md §8b / include/columba_amd.h: cmb_match_batch is re-entrant — N host threads share ONE index and ONE
    strategy handle (as the reference's workers share index + strategy, parallel.cpp:1143-1146; every batch owns its
    stream and scratch).  Four threads match four different chunks at the same time, twice; results and counters
    must be those of the same chunks matched one after the other."""
    import threading
    st = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
    world = {"dev": big["dev"]}
    chunks = [big["reads"][20000 * t:20000 * t + 15000 + 1000 * t] for t in range(4)]
    serial = [ca.match_batch(world["dev"], st, 4, c) for c in chunks]
    for _round in range(2):
        out, errs = [None] * 4, []

        def work(t):
            try:
                out[t] = ca.match_batch(world["dev"], st, 4, chunks[t])
            except Exception as e:  # noqa: BLE001
                errs.append((t, repr(e)))

        th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
        for t in range(4):
            assert np.array_equal(out[t][1], serial[t][1]) and np.array_equal(out[t][0], serial[t][0]), t
            assert out[t][2] == serial[t][2], t
    # mixed: Hamming and exact batches beside an edit-distance batch on the same handle
    mixed = [("edit", 4, "multiple_opt"), ("hamming", 2, "kuch1"), ("edit", 0, "kuch1"), ("edit", 2, "columba")]
    ser = [ca.match_batch(world["dev"], ca.SearchStrategy(n, m, "dynamic"), k, chunks[0]) for m, k, n in mixed]
    out, errs = [None] * 4, []

    def work2(t):
        try:
            m, k, n = mixed[t]
            out[t] = ca.match_batch(world["dev"], ca.SearchStrategy(n, m, "dynamic"), k, chunks[0])
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=work2, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(4):
        assert np.array_equal(out[t][0], ser[t][0]) and out[t][2] == ser[t][2], t


def test_staged_chunks_stream_through_one_batch(big):
    """cmb_batch_stage_reads: create(A); stage(B); run() -> A while B travels; stage(C); run() -> B; run() -> C —
    results are those of fresh batches on those chunks, for a plain and for a composite batch (sub-batches)."""
    st = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
    for sub in (None, "3"):
        if sub:
            os.environ["CMB_SUBBATCHES"] = sub
        try:
            a, b_, c = (big["reads"][i * 20000:(i + 1) * 20000] for i in range(3))
            want = [ca.match_batch(big["dev"], st, 4, x) for x in (a, b_, c)]
            chunks = [ca.pack_reads(x) for x in (a, b_, c)]
            batch = ca.Batch(big["dev"], st, 4, packed=chunks[0])
            got = []
            for i in range(3):
                if i + 1 < 3:
                    batch.stage(chunks[i + 1])
                batch.run()
                got.append(batch.results())
            batch.close()
        finally:
            os.environ.pop("CMB_SUBBATCHES", None)
        for w, g in zip(want, got):
            assert np.array_equal(w[0], g[0]) and np.array_equal(w[1], g[1]) and w[2] == g[2]


@pytest.fixture(scope="module")
def world():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    g, starts = synth.genome_rep(seed=11, n=1_000_000, scale=1.5)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
    return {"genome": g, "dev": ca.Index(ix)}


def test_reads_matched_by_naive_backtracking_leave_the_rest_of_the_chunk_alone(world):
    """Reads not longer than the number of parts (searchstrategy.cpp:148-152) are matched by naive backtracking on the device
    (tests/test_gpu_naive.py holds the parity tests): the chunk runs as a whole, those reads are marked in the read status,
    and the lists of all other reads are what they are without them"""
    g = world["genome"]
    reads = synth.sample_reads(g, 400, 150, seed=77)
    mixed = reads[:100] + [b"ACGTACG"[:5], b"ACGTA", b"TTGCA"] + reads[100:]
    st = ca.SearchStrategy("multiple_opt", "edit", "dynamic")
    b = ca.Batch(world["dev"], st, 4, mixed)
    b.run()
    occ, offs, cnt = b.results()
    status = b.read_status()
    assert status.tolist() == [0] * 100 + [1, 1, 1] + [0] * 300
    assert int(offs[103]) > int(offs[100])   # five characters with four errors match all over the text
    ref_occ, ref_offs, ref_cnt = ca.match_batch(world["dev"], st, 4, reads)
    keep = np.r_[0:100, 103:403]
    assert np.array_equal(np.diff(offs.astype(np.int64))[keep], np.diff(ref_offs.astype(np.int64)))
    assert np.array_equal(np.concatenate([occ[:int(offs[100])], occ[int(offs[103]):]]), ref_occ)
    assert cnt["SEARCH_STARTED"] == ref_cnt["SEARCH_STARTED"]
    b.close()
