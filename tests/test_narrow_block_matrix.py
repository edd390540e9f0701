"""An experiment on the CPU oracle, kept as a test because the next step of the device build rests on its outcome.

The reference runs the in-index matrix of a search part on 64-bit words where the part's upper bound is at most 10 and on its 128-bit
words beyond (indexinterface.cpp:391-398).  A device kernel would rather carry ONE matrix per node: a 64-bit word with 16-row blocks
holds the band of 13 errors by the reference's own bound (bitparallelmatrix.h:313 with BLOCK 16: (64 - 16 - 2) / 3 = 15).  Cells of at
most maxED are the same in any matrix that contains the band, so valid rows, final-column values, cluster centres and first columns
agree; the open question was `onlyVerticalGapsLeft`, which reads HN bits of cells that may exceed maxED and, in the reference, shifts by a
negative count near the end of a block (a region where its answer is `true` whatever the bits hold).  With ORC_NARROW_BLOCKS=1 the
oracle's search runs every part on BitParallelEDT<uint64_t, 16> and answers that predicate as the part's own matrix would
(onlyVerticalGapsLeftAs); occurrences and every counter must equal the run on the reference's matrices.

Round 4, the same question one size down: a 32-bit word with 8-row blocks holds the band of 7 errors by the same bound ((32 - 8 - 2) / 3
= 7, LEFT 15, DIAG 14) — ORC_NARROW_BLOCKS=32 runs every part on BitParallelEDT<uint32_t, 8>, the matrix the device's frontier kernels
carry per node since round 4 (dev_bfs_edit.hpp: GeoN32).  AT the bound the window has no slack beside the band and the matrix is not the
reference's (a few matrix rows more or fewer in replays on periodic texts at 7 errors: test_at_the_bound_the_small_matrix_differs); with
two spare columns — batches of up to 6 errors, Wv <= 12 — nothing differs: tools/soak_narrow32.py (human-like texts) and
tools/soak_narrow32_periodic.py (tandem repeats) are the long runs."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

from columba_amd import indexbuild as ib, synth  # noqa: E402


@pytest.fixture(scope="module")
def world(oracle_built):
    import oracle_py as op
    g, starts = synth.genome_rep(seed=41, n=400_000, scale=2.0)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cpu")
    return {"g": g, "op": op, "orc": {sw: op.OracleIndex(ix, switch_point=sw) for sw in (0, 4)}}


@pytest.mark.parametrize("partition,k,length,switch", [("dynamic", 5, 100, 4), ("dynamic", 8, 150, 4), ("uniform", 10, 150, 0), ("dynamic", 11, 150, 4),
                                                       ("static", 12, 250, 0), ("dynamic", 13, 150, 0), ("uniform", 13, 400, 4),
                                                       ("dynamic", 13, 480, 0), ("dynamic", 7, 60, 0)])
def test_one_narrow_block_matrix_stands_in_for_both_reference_matrices(world, partition, k, length, switch):
    import schemes_py as sp
    op = world["op"]
    reads = synth.sample_reads(world["g"], 400, length, seed=60 + k + length, n_frac=0.01, edit_choices=(0, 2, 5, 8, k - 1, k, k, k + 1))
    reads += [world["g"][:length].tobytes(), world["g"][-length - 1:-1].tobytes(), b"N" * length]
    st = op.OracleStrategy(sp.BY_NAME["columba"], "edit", partition)
    os.environ.pop("ORC_NARROW_BLOCKS", None)
    a_occ, a_off, a_cnt = op.match_batch(world["orc"][switch], st, k, reads, threads=8)
    os.environ["ORC_NARROW_BLOCKS"] = "1"
    try:
        b_occ, b_off, b_cnt = op.match_batch(world["orc"][switch], st, k, reads, threads=8)
    finally:
        del os.environ["ORC_NARROW_BLOCKS"]
    assert len(a_occ) > 300
    assert np.array_equal(a_off, b_off) and np.array_equal(a_occ, b_occ)
    assert a_cnt == b_cnt, {n: (a_cnt[n], b_cnt[n]) for n in a_cnt if a_cnt[n] != b_cnt[n]}


def test_the_experiment_sees_the_predicate(world):
    """self-check: with the predicate's answer inverted the runs differ — the comparison above does exercise it"""
    import schemes_py as sp
    op = world["op"]
    reads = synth.sample_reads(world["g"], 300, 150, seed=7, n_frac=0.01, edit_choices=(0, 2, 5, 8, 11, 12))
    st = op.OracleStrategy(sp.BY_NAME["columba"], "edit", "dynamic")
    os.environ.pop("ORC_NARROW_BLOCKS", None)
    _, _, a_cnt = op.match_batch(world["orc"][0], st, 12, reads, threads=8)
    os.environ["ORC_NARROW_BLOCKS"] = "inverted"
    try:
        _, _, b_cnt = op.match_batch(world["orc"][0], st, 12, reads, threads=8)
    finally:
        del os.environ["ORC_NARROW_BLOCKS"]
    assert a_cnt["NODE_COUNTER"] != b_cnt["NODE_COUNTER"]


@pytest.mark.parametrize("spec,partition,k,length,switch", [("multiple_opt", "dynamic", 4, 150, 4), ("columba", "dynamic", 6, 150, 4), ("columba", "uniform", 5, 480, 0),
                                                            ("kuch1", "static", 3, 100, 0), ("kianfar", "dynamic", 4, 60, 4), ("columba", "static", 6, 60, 4),
                                                            ("pigeon", "dynamic", 2, 250, 0), ("columba", "dynamic", 6, 40, 0)])
def test_the_32_bit_matrix_stands_in_up_to_six_errors(world, spec, partition, k, length, switch):
    import ctypes as C
    import schemes_py as sp
    op = world["op"]
    reads = synth.sample_reads(world["g"], 150 if spec == "kianfar" else 400, length, seed=90 + k + length, n_frac=0.01, edit_choices=(0, 1, 2, k - 1, k, k, k + 1))
    reads += [world["g"][:length].tobytes(), world["g"][-length - 1:-1].tobytes(), b"N" * length]
    st = op.OracleStrategy(sp.BY_NAME[spec], "edit", partition)
    os.environ.pop("ORC_NARROW_BLOCKS", None)
    a_occ, a_off, a_cnt = op.match_batch(world["orc"][switch], st, k, reads, threads=8)
    L = C.CDLL(op.build())
    stats = (C.c_uint64 * 2)()
    L.orc_narrow32_stats(stats, 1)
    os.environ["ORC_NARROW_BLOCKS"] = "32"
    try:
        b_occ, b_off, b_cnt = op.match_batch(world["orc"][switch], st, k, reads, threads=8)
    finally:
        del os.environ["ORC_NARROW_BLOCKS"]
    L.orc_narrow32_stats(stats, 1)
    assert stats[0] > 100, "the experiment did not run on the 32-bit matrix"
    assert len(a_occ) > 100
    assert np.array_equal(a_off, b_off) and np.array_equal(a_occ, b_occ)
    assert a_cnt == b_cnt, {n: (a_cnt[n], b_cnt[n]) for n in a_cnt if a_cnt[n] != b_cnt[n]}


def test_the_32_bit_experiment_sees_the_predicate(world):
    import schemes_py as sp
    op = world["op"]
    reads = synth.sample_reads(world["g"], 300, 150, seed=8, n_frac=0.01, edit_choices=(0, 2, 4, 5, 6))
    st = op.OracleStrategy(sp.BY_NAME["columba"], "edit", "dynamic")
    os.environ.pop("ORC_NARROW_BLOCKS", None)
    _, _, a_cnt = op.match_batch(world["orc"][0], st, 6, reads, threads=8)
    os.environ["ORC_NARROW_BLOCKS"] = "32inverted"
    try:
        _, _, b_cnt = op.match_batch(world["orc"][0], st, 6, reads, threads=8)
    finally:
        del os.environ["ORC_NARROW_BLOCKS"]
    assert a_cnt["NODE_COUNTER"] != b_cnt["NODE_COUNTER"]


def _periodic_case(op):
    import schemes_py as sp
    t = (b"ACGTTGCA" * 26)[:200]
    ix = ib.build_index(t, sparseness=4, seq_starts=np.array([0, 66, 200], np.uint32), device="cpu")
    orc = op.OracleIndex(ix, kmer_size=4, switch_point=0)
    g = np.frombuffer(t, np.uint8)
    # (found by tools/soak_narrow32_periodic.py and a search over seeds: 510 nodes on the reference's matrix, 512 on the small one at its bound)
    reads = [b"CGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTGCAACGTTGCAACGTTGCAAACGTT",
             b"GTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTCAACGTTGCAACGTTGCAACGTTGGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCA"]
    reads += synth.sample_reads(g, 200, 60, seed=5, edit_choices=(0, 1, 2, 6, 7, 8))
    return orc, op.OracleStrategy(sp.BY_NAME["columba"], "edit", "dynamic"), reads


def test_at_the_bound_the_small_matrix_differs(world, monkeypatch):
    """7 errors on a tandem repeat: with the small matrix forced to its sizing bound (no spare column beside the band) the search computes a
    different number of matrix rows than on the reference's matrices — the reason GeoN32 stops at 6 errors and two spare columns; with
    the bounds the device uses the same batch runs on the reference's matrix and is identical."""
    import ctypes as C
    op = world["op"]
    orc, st, reads = _periodic_case(op)
    os.environ.pop("ORC_NARROW_BLOCKS", None)
    a = op.match_batch(orc, st, 7, reads, threads=8)
    monkeypatch.setenv("ORC_NARROW_BLOCKS", "32")
    L = C.CDLL(op.build())
    stats = (C.c_uint64 * 2)()
    L.orc_narrow32_stats(stats, 1)
    b = op.match_batch(orc, st, 7, reads, threads=8)
    L.orc_narrow32_stats(stats, 1)
    assert stats[0] == 0 and a[2] == b[2] and np.array_equal(a[0], b[0])          # the device's bounds: 7 errors stay on 64 bits
    monkeypatch.setenv("ORC_NARROW32_MAX", "7")
    monkeypatch.setenv("ORC_NARROW32_SLACK", "0")
    c = op.match_batch(orc, st, 7, reads, threads=8)
    L.orc_narrow32_stats(stats, 1)
    assert stats[0] > 100
    assert np.array_equal(a[0], c[0])                                              # (the occurrences survive; the traversal does not)
    assert a[2]["MATRIX_ROWS"] != c[2]["MATRIX_ROWS"] or a[2]["NODE_COUNTER"] != c[2]["NODE_COUNTER"]
    # 6 errors with two spare columns on the same text: identical, on the small matrix
    monkeypatch.delenv("ORC_NARROW32_MAX")
    monkeypatch.delenv("ORC_NARROW32_SLACK")
    monkeypatch.delenv("ORC_NARROW_BLOCKS")
    d = op.match_batch(orc, st, 6, reads, threads=8)
    monkeypatch.setenv("ORC_NARROW_BLOCKS", "32")
    L.orc_narrow32_stats(stats, 1)
    e = op.match_batch(orc, st, 6, reads, threads=8)
    L.orc_narrow32_stats(stats, 1)
    assert stats[0] > 100 and d[2] == e[2] and np.array_equal(d[0], e[0]) and np.array_equal(d[1], e[1])
