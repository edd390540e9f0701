"""Model-independent checks of the matcher: SOUNDNESS and COMPLETENESS against textbook dynamic programming.

Every other parity test compares the HIP path with `oracle/`, which for the orchestration level (extension, search-scheme
DFS, partitioning, redundancy filter) is the builder's own reading of the reference (parity unpinned there).
A misreading shared by oracle and kernels is invisible to those tests.  These are not: `oracle/groundtruth.c` is plain
O(mn) edit distance and Sellers' semi-global alignment — no index, no scheme, no bit-parallel matrix.

  soundness     every reported occurrence (begin, end, distance, strand): the read (its reverse complement on strand 1)
                really aligns with text[begin, end) at `distance` edits or fewer, and never at more than k.  For Hamming
                distance the window has the read's length and exactly `distance` mismatches.
  completeness  Columba is lossless (search schemes cover every distribution of <= k errors; the final filter only removes
                occurrences that have a better-or-equal neighbour within 2k positions, indexinterface.cpp:1445-1485).  For
                every end position j of the text with min_b ED(read, text[b, j)) = d <= k there must be a reported
                occurrence with distance <= d ending within 4k of j (2k for the redundancy window on the begin, 2k for the
                width of an alignment with k indels), on either strand of the window.

What this does NOT prove: that the list is the reference's list (which representative of a 2k window survives, counters,
order) — that is what the oracle comparison is for.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from columba_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(os.path.dirname(HERE), "oracle")


@pytest.fixture(scope="module")
def gt(oracle_built):
    so = os.path.join(ORACLE, "libgroundtruth.so")
    src = os.path.join(ORACLE, "groundtruth.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-o", so, src])
    L = C.CDLL(so)
    L.gt_edit_distance.restype = C.c_uint32
    L.gt_edit_distance.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32]
    L.gt_hamming_distance.restype = C.c_uint32
    L.gt_hamming_distance.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
    L.gt_semiglobal_ends.restype = None
    L.gt_semiglobal_ends.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return L


def clean(read: bytes) -> bytes:
    """reads.h:43-58: upper case, everything outside ACGT becomes N"""
    a = np.frombuffer(read.upper(), dtype=np.uint8).copy()
    a[~np.isin(a, np.frombuffer(b"ACGT", dtype=np.uint8))] = ord("N")
    return a.tobytes()


def check_soundness(gt, text: bytes, reads, occ, offs, k: int, metric: str):
    """returns (occurrences checked, occurrences whose reported distance is above the true distance of their window)"""
    n_checked = n_loose = 0
    for i, rd in enumerate(reads):
        fw = clean(rd)
        pats = (fw, synth.revcomp(fw))
        for j in range(int(offs[i]), int(offs[i + 1])):
            b, e, d, s = int(occ["begin"][j]), int(occ["end"][j]), int(occ["distance"][j]), int(occ["strand"][j])
            assert 0 <= b < e <= len(text) and d <= k and s in (0, 1), (i, b, e, d, s)
            win = text[b:e]
            p = pats[s]
            if metric == "hamming":
                assert e - b == len(p), (i, b, e)
                assert gt.gt_hamming_distance(p, win, len(p)) == d, (i, b, e, d)
            else:
                true = gt.gt_edit_distance(p, len(p), win, len(win))
                assert true <= d, (i, b, e, d, true, "reported distance below what the window allows")
                n_loose += true < d
            n_checked += 1
    return n_checked, n_loose


def check_completeness(gt, text: bytes, reads, occ, offs, k: int, metric: str):
    """returns (end positions within k, of those: covered only through the chain rule)"""
    n = len(text)
    best = np.zeros(n + 1, np.uint8)
    n_hits = n_chain = 0
    for i, rd in enumerate(reads):
        fw = clean(rd)
        lo, hi = int(offs[i]), int(offs[i + 1])
        ends = occ["end"][lo:hi].astype(np.int64)
        begins = occ["begin"][lo:hi].astype(np.int64)
        dist = occ["distance"][lo:hi].astype(np.int64)
        for strand, p in enumerate((fw, synth.revcomp(fw))):
            if metric == "hamming":
                # windows of the read's length with <= k mismatches: compare directly
                pa = np.frombuffer(p, dtype=np.uint8)
                ta = np.frombuffer(text, dtype=np.uint8)
                m = len(pa)
                if m > n:
                    continue
                mism = np.zeros(n - m + 1, np.int32)
                for c in range(m):
                    col = ta[c:n - m + 1 + c]
                    mism += (col != pa[c]) | (pa[c] == ord("N"))
                for b in np.flatnonzero(mism <= k):
                    n_hits += 1
                    ok = np.any((begins == b) & (dist == mism[b]))
                    assert ok, (i, strand, int(b), int(mism[b]), "Hamming window within k not reported")
                continue
            gt.gt_semiglobal_ends(text, n, p, len(p), k, best.ctypes.data_as(C.c_void_p), None)
            for j in np.flatnonzero(best <= k):
                d = int(best[j])
                n_hits += 1
                near = np.abs(ends - j) <= 4 * k
                if np.any(near & (dist <= d)):
                    continue
                # the redundancy filter is sequential in the last occurrence KEPT (indexinterface.cpp:1451-1485): a run
                # of ever better occurrences each within 2k of its predecessor replaces one another, so the survivor can
                # lie further than 2k from the first.  Accept a survivor reached through such a run, no further than
                # one more window away, and count how often that is needed.
                far = np.abs(ends - j) <= 8 * k
                assert np.any(far & (dist <= d)), (i, strand, int(j), d, list(zip(begins.tolist(), ends.tolist(), dist.tolist())))
                n_chain += 1
    return n_hits, n_chain


CONFIGS = [
    ("multiple_opt", "edit", "dynamic", 4),
    ("kuch1", "edit", "static", 3),
    ("columba", "edit", "dynamic", 2),
    ("pigeon", "edit", "uniform", 1),
    ("minU", "edit", "dynamic", 5),
    ("kuch1", "hamming", "dynamic", 2),
    ("multiple_opt", "hamming", "uniform", 4),
]


def _small_world(n=120_000):
    from columba_amd import indexbuild as ib
    g, starts = synth.genome_rep(seed=23, n=n, scale=4.0)
    return g, starts, ib


def _reads_for(g, k, count, length=100, seed=0):
    return synth.sample_reads(g, count, length, seed=seed, n_frac=0.03, edit_choices=(0, 1, 2, max(k - 1, 0), k, k, k + 1))


# ------------------------------------------------------------------------------------------------ CPU: the oracle
@pytest.fixture(scope="module")
def cpu_world(oracle_built):
    import oracle_py as op
    g, starts, ib = _small_world()
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cpu")
    return {"genome": g, "text": g.tobytes(), "orc": op.OracleIndex(ix), "orc4": op.OracleIndex(ix, kmer_size=4), "op": op}


# (11 ... 13 errors under edit distance: the reference runs the parts whose upper bound exceeds 10 on its 128-bit in-index matrix,
# indexinterface.cpp:391-398)
@pytest.mark.parametrize("spec,metric,partition,k", CONFIGS + [("columba", "edit", "dynamic", 9), ("columba", "edit", "dynamic", 11),
                                                              ("columba", "edit", "uniform", 12), ("columba", "edit", "static", 13)])
def test_oracle_is_sound_and_complete(cpu_world, gt, spec, metric, partition, k):
    import schemes_py as sp
    op = cpu_world["op"]
    reads = _reads_for(cpu_world["genome"], k, 60 if k <= 7 else (100 if k >= 11 else 24), length=100 if k <= 7 else 150, seed=500 + k)
    reads += [b"N" * 60, cpu_world["text"][:100], cpu_world["text"][-100:], b"ACGT" * 20]
    st = op.OracleStrategy(sp.BY_NAME[spec], metric, partition)
    occ, offs, _ = op.match_batch(cpu_world["orc"], st, k, reads, threads=4)
    checked, loose = check_soundness(gt, cpu_world["text"], reads, occ, offs, k, metric)
    hits, chain = check_completeness(gt, cpu_world["text"], reads, occ, offs, k, metric)
    assert checked > (40 if k <= 7 else 20) and hits > (40 if k <= 7 else 20)
    assert chain * 50 <= hits  # the chain rule is the exception
    # a reported distance ABOVE the window's true distance is legal (a scheme bounds the errors per part) but must stay the exception too
    assert loose * 50 <= checked, (loose, checked)


def test_ground_truth_functions_against_brute_force(gt):
    """the checker itself: Sellers' ends against the O(mn^2) definition on tiny inputs"""
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for trial in range(40):
        n, m, k = int(rng.integers(5, 60)), int(rng.integers(1, 12)), int(rng.integers(0, 4))
        text = rng.choice(acgt[:2 + trial % 3], n).tobytes()
        pat = rng.choice(acgt[:2 + trial % 3], m).tobytes()
        if trial % 5 == 0:
            pat = pat[:m // 2] + b"N" + pat[m // 2 + 1:]
        best = np.zeros(n + 1, np.uint8)
        begin = np.zeros(n + 1, np.uint64)
        gt.gt_semiglobal_ends(text, n, pat, m, k, best.ctypes.data_as(C.c_void_p), begin.ctypes.data_as(C.c_void_p))
        for j in range(n + 1):
            ds = [gt.gt_edit_distance(pat, m, text[b:j], j - b) for b in range(j + 1)]
            d = min(ds)
            assert int(best[j]) == min(d, k + 1), (trial, j)
            if d <= k:
                b = int(begin[j])
                assert ds[b] == d and all(x > d for x in ds[b + 1:]), (trial, j, b)
    assert gt.gt_edit_distance(b"ACGT", 4, b"AGT", 3) == 1 and gt.gt_edit_distance(b"ANGT", 4, b"ANGT", 4) == 1
    assert gt.gt_hamming_distance(b"ACGN", b"ACGN", 4) == 1


# ------------------------------------------------------------------------------------------------ GPU: the HIP path
@pytest.fixture(scope="module")
def gpu_world(oracle_built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import columba_amd as ca
    g, starts, ib = _small_world(200_000)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda")
    return {"genome": g, "text": g.tobytes(), "dev": ca.Index(ix), "dev4": ca.Index(ix, kmer_size=4), "ca": ca, "ix": ix}


@pytest.mark.gpu
@pytest.mark.parametrize("spec,metric,partition,k", CONFIGS + [
    ("columba", "edit", "dynamic", 7), ("multiple_opt", "edit", "dynamic", 6), ("kianfar", "edit", "dynamic", 3),
    ("01*0", "edit", "static", 2), ("kuch2", "hamming", "dynamic", 3),
    # beyond 7 errors (greedy schemes, wide device tables and records, the in-text matrix with the wide left margin)
    ("columba", "edit", "dynamic", 8), ("columba", "edit", "uniform", 10), ("columba", "hamming", "dynamic", 9),
    ("columba", "hamming", "static", 13),
    # ... and beyond 10: the in-index matrix with 16-row blocks, the in-text matrix with 8-row blocks
    ("columba", "edit", "dynamic", 11), ("columba", "edit", "static", 12), ("columba", "edit", "uniform", 13)])
def test_device_is_sound_and_complete(gpu_world, gt, spec, metric, partition, k):
    ca = gpu_world["ca"]
    g = gpu_world["genome"]
    n1, n2 = (250, 80) if k <= 7 else ((80, 30) if k >= 11 else (60, 30))   # (the lists grow quickly with k on this repeat-rich text)
    reads = _reads_for(g, k, n1, seed=900 + k) + _reads_for(g, k, n2, length=151, seed=950 + k)
    reads += [b"N" * 60, gpu_world["text"][:100], gpu_world["text"][-100:], b"ACGT" * 20]
    dev = gpu_world["dev4" if spec in ("kuch2", "01*0") else "dev"]
    occ, offs, _ = ca.match_batch(dev, ca.SearchStrategy(spec, metric, partition), k, reads)
    checked, loose = check_soundness(gt, gpu_world["text"], reads, occ, offs, k, metric)
    hits, chain = check_completeness(gt, gpu_world["text"], reads, occ, offs, k, metric)
    assert checked > (200 if k <= 7 else 60) and hits > (200 if k <= 7 else 60)
    assert chain * 50 <= hits
    assert loose * 50 <= checked, (loose, checked)


@pytest.mark.gpu
def test_device_cigars_rederive_the_distance(gpu_world, gt):
    """CIGAR of every final occurrence: consumes the whole read and exactly text[begin, end); its number of edits is the
    true edit distance of that window (findCIGAR aligns on a fresh matrix, bitparallelmatrix.h:460-527) and never above
    the reported distance"""
    ca = gpu_world["ca"]
    g = gpu_world["genome"]
    for spec, k in (("multiple_opt", 4), ("columba", 7), ("kuch1", 2)):
        reads = _reads_for(g, k, 300, seed=1300 + k)
        b = ca.Batch(gpu_world["dev"], ca.SearchStrategy(spec, "edit", "dynamic"), k, reads)
        b.want_alignments()
        b.run()
        occ, offs, _ = b.results()
        aln, ops = b.alignments()
        n_checked = 0
        for i, rd in enumerate(reads):
            fw = clean(rd)
            for j in range(int(offs[i]), int(offs[i + 1])):
                o = ops[int(aln["cigar_off"][j]):int(aln["cigar_off"][j]) + int(aln["cigar_len"][j])]
                p = fw if int(occ["strand"][j]) == 0 else synth.revcomp(fw)
                win = gpu_world["text"][int(occ["begin"][j]):int(occ["end"][j])]
                qi = ti = edits = 0
                for w in o.tolist():
                    kind, ln = "MID?"[w & 3], w >> 2  # (length << 2 | op; 0 = M, 1 = I, 2 = D: include/columba_amd.h)
                    if kind == "M":
                        edits += sum(1 for a, c in zip(p[qi:qi + ln], win[ti:ti + ln]) if a != c or a == ord("N"))
                        qi += ln
                        ti += ln
                    elif kind == "I":
                        qi += ln
                        edits += ln
                    else:
                        assert kind == "D"
                        ti += ln
                        edits += ln
                assert qi == len(p) and ti == len(win), (i, j)
                true = gt.gt_edit_distance(p, len(p), win, len(win))
                assert edits == true <= int(occ["distance"][j]), (i, j, edits, true, int(occ["distance"][j]))
                n_checked += 1
        b.close()
        assert n_checked > 200
