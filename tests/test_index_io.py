"""Index container: builder self-consistency, file round trip in the reference's on-disk formats, and
the oracle's format restatements against the builder (BitvecIntl / rank9 / EncodedText)."""
import numpy as np

from columba_amd import indexbuild as ib
from columba_amd import synth


def _naive_sa(t: bytes):
    return sorted(range(len(t)), key=lambda i: t[i:])


def test_builder_matches_naive_suffix_array():
    rng = np.random.default_rng(0)
    for n in (1, 2, 17, 64, 200, 1000):
        t = synth.ACGT[rng.integers(0, 4, n)].tobytes() + b"$"
        ix = ib.build_index(t)
        sa = np.array(_naive_sa(t))
        # sampled rows: SA[i] % 4 == 0, in row order
        assert np.array_equal(ix.sa_samples, sa[sa % 4 == 0].astype(np.uint32))
        mark = (sa % 4 == 0)
        bits = np.unpackbits(ix.sa_bv.view(np.uint8), bitorder="little")[:len(t)].astype(bool)
        assert np.array_equal(bits, mark)
        # BWT through the cumulative bitvectors: symbol = first set bit (0 = '$')
        bwt = bytes(t[i - 1] for i in sa)
        bv = ix.bv_fwd.reshape(-1, 4)
        for i in range(len(t)):
            code = 0
            for c in range(4):
                if (int(bv[i // 64, c]) >> (i % 64)) & 1:
                    code = c + 1
                    break
            assert b"$ACGT"[code] == bwt[i]
        assert ix.dollar_pos_fwd == bwt.index(b"$")


def test_save_load_round_trip(tmp_path):
    g, starts = synth.genome_small(seed=4, n=50_000)
    ix = ib.build_index(g.tobytes(), seq_starts=starts, seq_names=["chrA", "chrB"])
    ib.save_index(ix, str(tmp_path / "i"))
    ld = ib.load_index(str(tmp_path / "i"))
    for f in ("text", "counts", "bv_fwd", "cnt_fwd", "bv_rev", "cnt_rev", "bwt_words", "sa_bv", "sa_bv_counts",
              "sa_samples", "seq_starts"):
        assert np.array_equal(getattr(ix, f), getattr(ld, f)), f
    assert (ld.dollar_pos_fwd, ld.dollar_pos_rev, ld.seq_names) == (ix.dollar_pos_fwd, ix.dollar_pos_rev, ["chrA", "chrB"])
    # .brt size formula of SURVEY.md §5 (n = 1 000 001 -> 625 104 B)
    n1 = ix.n + 1
    assert (tmp_path / "i.brt").stat().st_size == 16 + 8 * (4 * ((n1 + 63) // 64) + 8 * ((n1 + 511) // 512))


def test_oracle_format_restatements_agree_with_builder(oracle_built):
    import ctypes as C
    import oracle_py as op
    rng = np.random.default_rng(3)
    for n in (5, 64, 513, 4000):
        t = synth.ACGT[rng.integers(0, 4, n)].tobytes() + b"$"
        ix = ib.build_index(t)
        sa = np.array(_naive_sa(t))
        codes = np.array([b"$ACGT".index(t[i - 1]) for i in sa], dtype=np.uint8)
        N = len(t) + 1
        bv = np.zeros(4 * ((N + 63) // 64), np.uint64)
        cnt = np.zeros(8 * ((N + 511) // 512), np.uint64)
        dp = C.c_uint64()
        op.lib().orc_build_bitvec_intl(codes.ctypes.data, len(t), bv.ctypes.data, cnt.ctypes.data, C.byref(dp))
        assert np.array_equal(bv, ix.bv_fwd) and np.array_equal(cnt, ix.cnt_fwd) and dp.value == ix.dollar_pos_fwd
        words = np.zeros(ix.bwt_words.shape[0] + 1, np.uint64)
        op.lib().orc_encode_bwt(codes.ctypes.data, len(t), words.ctypes.data)
        assert np.array_equal(words[:-1], ix.bwt_words)
        c9 = np.zeros(ix.sa_bv_counts.shape[0], np.uint64)
        op.lib().orc_build_bitvec9_counts(ix.sa_bv.ctypes.data, ix.sa_bv.shape[0], c9.ctypes.data)
        assert np.array_equal(c9, ix.sa_bv_counts)
