"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of ``oracle/liboracle.so`` (the CPU restatement of the reference's
hot path, see oracle_core.hpp).  Only tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py may import this module, and there only as the
checker.  The product package ``columba_amd`` never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

COUNTER_NAMES = [
    "NODE_COUNTER", "TOTAL_REPORTED_POSITIONS", "IN_TEXT_STARTED", "ABORTED_IN_TEXT_VERIF",
    "CIGARS_IN_TEXT_VERIFICATION", "IMMEDIATE_SWITCH", "SEARCH_STARTED",
    "EXPANSIONS", "LF_STEPS", "LOCATED_ROWS", "TEXT_BYTES", "MATRIX_ROWS",
    "SURVIVING_DUP_ROWS", "SURVIVING_DUP_LF", "ROW_STEPS",
]


def build(force: bool = False) -> str:
    """Compile liboracle.so with g++ (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle_api.cpp", "oracle_core.hpp", "oracle_search.hpp", "oracle_move.hpp",
                                             "oracle_move_search.hpp")]
    if not force and os.path.exists(_LIB_PATH) and all(
            os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return _LIB_PATH
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                           "-o", _LIB_PATH, srcs[0]], cwd=_HERE)
    return _LIB_PATH


class _IndexDesc(C.Structure):
    _fields_ = [
        ("text_length", C.c_uint64), ("text", C.c_void_p), ("counts", C.c_uint64 * 5),
        ("dollar_pos_fwd", C.c_uint64), ("bv_fwd", C.c_void_p), ("cnt_fwd", C.c_void_p),
        ("dollar_pos_rev", C.c_uint64), ("bv_rev", C.c_void_p), ("cnt_rev", C.c_void_p),
        ("bwt_words", C.c_void_p), ("sa_bv", C.c_void_p), ("sa_bv_counts", C.c_void_p),
        ("sa_samples", C.c_void_p), ("sa_sparseness", C.c_uint32), ("switch_point", C.c_uint32),
        ("kmer_size", C.c_uint32), ("n_seqs", C.c_uint32), ("seq_starts", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_index_create.restype = C.c_void_p
        L.orc_index_create.argtypes = [C.POINTER(_IndexDesc)]
        L.orc_index_destroy.argtypes = [C.c_void_p]
        L.orc_index_kmer_table.restype = C.c_void_p
        L.orc_index_kmer_table.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_strategy_create.restype = C.c_void_p
        L.orc_strategy_create.argtypes = [C.c_int, C.c_int, C.c_uint32]
        L.orc_strategy_destroy.argtypes = [C.c_void_p]
        L.orc_strategy_add_scheme.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_strategy_set_partition_params.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_scheme_critical_part.restype = C.c_uint32
        L.orc_scheme_critical_part.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_match_batch.restype = C.c_void_p
        L.orc_match_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_uint32]
        L.orc_result_error.restype = C.c_char_p
        L.orc_result_error.argtypes = [C.c_void_p]
        L.orc_result_size.restype = C.c_uint64
        L.orc_result_size.argtypes = [C.c_void_p]
        L.orc_result_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_result_free.argtypes = [C.c_void_p]
        L.orc_num_counters.restype = C.c_uint32
        L.orc_rank_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_occ_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                    C.c_void_p, C.c_void_p]
        L.orc_extend_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_locate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_verify_batch.restype = C.c_uint64
        L.orc_verify_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                       C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_matrix_dump.restype = C.c_uint32
        L.orc_matrix_dump.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_search_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_build_bitvec_intl.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_build_bitvec9_counts.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_encode_bwt.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_bwt_at.restype = C.c_uint64
        L.orc_bwt_at.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_move_create.restype = C.c_void_p
        L.orc_move_create.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64] + [C.c_void_p] * 9
        L.orc_move_destroy.argtypes = [C.c_void_p]
        L.orc_move_complete_range.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_move_rows.restype = C.c_uint64
        L.orc_move_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_move_extend.restype = C.c_uint64
        L.orc_move_extend.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_move_match_exact.restype = C.c_uint64
        L.orc_move_match_exact.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_move_kmer_table.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_move_match_batch.restype = C.c_void_p
        L.orc_move_match_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_move_locate.restype = C.c_uint64
        L.orc_move_locate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


OCC_DTYPE = np.dtype([("begin", np.uint32), ("end", np.uint32), ("distance", np.uint32),
                      ("strand", np.uint32)])


class OracleIndex:
    def __init__(self, ix, switch_point: int = 4, kmer_size: int = 10):
        self.ix = ix  # keep arrays alive
        d = _IndexDesc()
        d.text_length = ix.n
        d.text = _p(ix.text)
        for i in range(5):
            d.counts[i] = int(ix.counts[i])
        d.dollar_pos_fwd = ix.dollar_pos_fwd
        d.bv_fwd = _p(ix.bv_fwd)
        d.cnt_fwd = _p(ix.cnt_fwd)
        d.dollar_pos_rev = ix.dollar_pos_rev
        d.bv_rev = _p(ix.bv_rev)
        d.cnt_rev = _p(ix.cnt_rev)
        self._bwt = np.concatenate([ix.bwt_words, np.zeros(1, np.uint64)])
        d.bwt_words = _p(self._bwt)
        d.sa_bv = _p(ix.sa_bv)
        d.sa_bv_counts = _p(ix.sa_bv_counts)
        d.sa_samples = _p(ix.sa_samples)
        d.sa_sparseness = ix.sparseness
        d.switch_point = switch_point
        d.kmer_size = kmer_size
        self._starts = np.ascontiguousarray(ix.seq_starts, dtype=np.uint32)
        d.n_seqs = self._starts.shape[0]
        d.seq_starts = _p(self._starts)
        self.h = lib().orc_index_create(C.byref(d))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_index_destroy(self.h)
            self.h = None

    def kmer_table(self) -> np.ndarray:
        n = C.c_uint64()
        ptr = lib().orc_index_kmer_table(self.h, C.byref(n))
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n.value, 4))
        return arr.copy()

    def rank(self, rev: int, c: np.ndarray, p: np.ndarray) -> np.ndarray:
        c = np.ascontiguousarray(c, np.uint32)
        p = np.ascontiguousarray(p, np.uint64)
        out = np.zeros(p.shape[0], np.uint64)
        lib().orc_rank_batch(self.h, rev, _p(c), _p(p), p.shape[0], _p(out))
        return out

    def occ(self, rev: int, c: np.ndarray, p: np.ndarray):
        c = np.ascontiguousarray(c, np.uint32)
        p = np.ascontiguousarray(p, np.uint64)
        o = np.zeros(p.shape[0], np.uint64)
        q = np.zeros(p.shape[0], np.uint64)
        lib().orc_occ_batch(self.h, rev, _p(c), _p(p), p.shape[0], _p(o), _p(q))
        return o, q

    def extend(self, mode: int, ranges: np.ndarray):
        r = np.ascontiguousarray(ranges, np.uint32).reshape(-1, 4)
        out = np.zeros((r.shape[0], 4, 4), np.uint32)
        ok = np.zeros((r.shape[0], 4), np.uint8)
        lib().orc_extend_batch(self.h, mode, _p(r), r.shape[0], _p(out), _p(ok))
        return out, ok

    def locate(self, rows: np.ndarray):
        rows = np.ascontiguousarray(rows, np.uint32)
        out = np.zeros(rows.shape[0], np.uint32)
        lf = C.c_uint64()
        lib().orc_locate_batch(self.h, _p(rows), rows.shape[0], _p(out), C.byref(lf))
        return out, lf.value

    def verify(self, pattern: bytes, starts: np.ndarray, max_ed: int, min_ed: int, fixed: bool):
        starts = np.ascontiguousarray(starts, np.uint32)
        cap = max(16, starts.shape[0] * 32)
        out = np.zeros(cap, OCC_DTYPE)
        cnt = np.zeros(len(COUNTER_NAMES), np.uint64)
        n = lib().orc_verify_batch(self.h, pattern, len(pattern), _p(starts), starts.shape[0],
                                   max_ed, min_ed, int(fixed), _p(out), cap, _p(cnt))
        assert n <= cap
        return out[:n], dict(zip(COUNTER_NAMES, cnt.tolist()))


METRIC = {"hamming": 0, "edit": 1}
PARTITION = {"uniform": 0, "static": 1, "dynamic": 2}


class OracleStrategy:
    """Strategy described by explicit tables (see schemes_py.py)."""

    def __init__(self, spec: Dict, metric: str = "edit", partition: str = "dynamic"):
        self.h = lib().orc_strategy_create(METRIC[metric], PARTITION[partition], spec.get("kmer_cutoff", 20))
        self.spec = spec
        for k, schemes in spec["schemes"].items():
            for sch in schemes:
                pi = np.ascontiguousarray([s[0] for s in sch], np.uint32)
                lo = np.ascontiguousarray([s[1] for s in sch], np.uint32)
                up = np.ascontiguousarray([s[2] for s in sch], np.uint32)
                rc = lib().orc_strategy_add_scheme(self.h, k, pi.shape[0], pi.shape[1], _p(pi), _p(lo), _p(up))
                if rc != 0:
                    raise RuntimeError("invalid scheme")
        for k, pp in spec.get("partition_params", {}).items():
            seed = np.ascontiguousarray(pp.get("seeding", []), np.float64)
            w = np.ascontiguousarray(pp.get("weights", []), np.uint64)
            b = np.ascontiguousarray(pp.get("begins", []), np.float64)
            lib().orc_strategy_set_partition_params(self.h, k, _p(seed), seed.shape[0], _p(w), w.shape[0],
                                                    _p(b), b.shape[0])

    def critical_part(self, k: int, scheme: int) -> int:
        return lib().orc_scheme_critical_part(self.h, k, scheme)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_strategy_destroy(self.h)
            self.h = None


def match_best(index: "OracleIndex", strat: "OracleStrategy", reads: Sequence[bytes], x: int = 0, min_identity: int = 95,
               max_supported: int = 6, threads: int = 1, word_size: int = 10):
    """BEST (+x strata) mode of the oracle: (occs, seq ids, begins inside the sequence, CIGAR strings, offsets, best distance
    per read (0xFFFFFFFF: unmapped), hits at it, counters)"""
    L = lib()
    L.orc_match_best.restype = C.c_void_p
    L.orc_match_best.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.orc_best_error.restype = C.c_char_p
    L.orc_best_error.argtypes = [C.c_void_p]
    L.orc_best_size.restype = C.c_uint64
    L.orc_best_size.argtypes = [C.c_void_p]
    L.orc_best_copy.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    L.orc_best_cigar.restype = C.c_char_p
    L.orc_best_cigar.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_best_free.argtypes = [C.c_void_p]
    buf, offs = pack_reads(reads)
    if isinstance(index, OracleMoveIndex):  # the RUN_LENGTH_COMPRESSION flavour (attach_text first: CIGARs, trimming)
        L.orc_move_match_best.restype = C.c_void_p
        L.orc_move_match_best.argtypes = L.orc_match_best.argtypes + [C.c_uint32]
        r = L.orc_move_match_best(index.h, strat.h, x, min_identity, max_supported, _p(buf), _p(offs), len(reads), threads, word_size)
    else:
        r = L.orc_match_best(index.h, strat.h, x, min_identity, max_supported, _p(buf), _p(offs), len(reads), threads)
    try:
        err = L.orc_best_error(r)
        if err:
            raise RuntimeError(err.decode())
        n = int(L.orc_best_size(r))
        occs = np.zeros(max(n, 1), OCC_DTYPE)
        sid, sb = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        ro = np.zeros(len(reads) + 1, np.uint64)
        best, hits = np.zeros(max(len(reads), 1), np.uint32), np.zeros(max(len(reads), 1), np.uint32)
        cnt = np.zeros(len(COUNTER_NAMES), np.uint64)
        L.orc_best_copy(r, _p(occs), _p(sid), _p(sb), _p(ro), _p(best), _p(hits), _p(cnt))
        cig = [L.orc_best_cigar(r, i).decode() for i in range(n)]
        return occs[:n], sid[:n], sb[:n], cig, ro, best[:len(reads)], hits[:len(reads)], dict(zip(COUNTER_NAMES, cnt.tolist()))
    finally:
        L.orc_best_free(r)


def match_batch_sam(index: "OracleIndex", strat: "OracleStrategy", k: int, reads: Sequence[bytes], ids, quals, seq_names,
                    unmapped: bool = True, xa: bool = False) -> str:
    """SAM text of a chunk of reads in ALL mode (generateOutputSingleEnd, searchstrategy.cpp:1824-1902)"""
    L = lib()
    L.orc_match_batch_sam.restype = C.c_void_p
    L.orc_match_batch_sam.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p,
                                      C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.orc_text_get.restype = C.c_char_p
    L.orc_text_get.argtypes = [C.c_void_p]
    L.orc_text_error.restype = C.c_char_p
    L.orc_text_error.argtypes = [C.c_void_p]
    L.orc_text_free.argtypes = [C.c_void_p]
    buf, offs = pack_reads(reads)
    r = L.orc_match_batch_sam(index.h, strat.h, k, _p(buf), _p(offs), len(reads), "\n".join(ids).encode(),
                              "\n".join(quals).encode(), "\n".join(seq_names).encode(), int(unmapped), int(xa))
    try:
        err = L.orc_text_error(r)
        if err:
            raise RuntimeError(err.decode())
        return L.orc_text_get(r).decode()
    finally:
        L.orc_text_free(r)


def pack_reads(reads: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    buf = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    return buf, offs


def match_batch(index: OracleIndex, strat: OracleStrategy, k: int, reads: Sequence[bytes] = None,
                threads: int = 1, packed=None):
    """Returns (occs structured array, offs uint64[n+1], counters dict).  packed = (characters, offsets): reads already
    packed (a timed caller packs before the clock starts)."""
    buf, offs = packed if packed is not None else pack_reads(reads)
    reads = range(offs.shape[0] - 1) if reads is None else reads
    if buf.shape[0] == 0:
        buf = np.zeros(1, np.uint8)
    r = lib().orc_match_batch(index.h, strat.h, k, _p(buf), _p(offs), len(reads), threads)
    try:
        err = lib().orc_result_error(r)
        if err:
            raise RuntimeError(err.decode())
        n = lib().orc_result_size(r)
        occs = np.zeros(max(n, 1), OCC_DTYPE)
        ro = np.zeros(len(reads) + 1, np.uint64)
        cnt = np.zeros(lib().orc_num_counters(), np.uint64)
        lib().orc_result_copy(r, _p(occs), _p(ro), _p(cnt))
    finally:
        lib().orc_result_free(r)
    return occs[:n], ro, dict(zip(COUNTER_NAMES, cnt.tolist()))


def matrix_dump(X: bytes, Y: bytes, max_ed: int, init_ed: Sequence[int] = ()):
    ie = np.ascontiguousarray(init_ed, np.uint32)
    rows = np.zeros((len(Y) + 1, 10), np.uint64)
    geom = np.zeros(5, np.uint32)
    n = lib().orc_matrix_dump(X, len(X), Y, len(Y), max_ed, _p(ie), ie.shape[0], _p(rows), _p(geom))
    return rows[:n], geom


def search_info(pi, lo, up):
    pi = np.ascontiguousarray(pi, np.uint32)
    lo = np.ascontiguousarray(lo, np.uint32)
    up = np.ascontiguousarray(up, np.uint32)
    n = pi.shape[0]
    out = np.zeros(5 * n, np.uint32)
    lib().orc_search_info(_p(pi), _p(lo), _p(up), n, _p(out))
    return out.reshape(5, n)


# ---- run-length compressed backend (oracle_move.hpp) ------------------------------------------------------------------
MOVE_RANGE_DTYPE = np.dtype([("begin", np.uint64), ("end", np.uint64), ("begin_run", np.uint64), ("end_run", np.uint64),
                             ("rev_begin", np.uint64), ("rev_end", np.uint64), ("rev_begin_run", np.uint64),
                             ("rev_end_run", np.uint64), ("toehold", np.uint64), ("original_depth", np.uint32),
                             ("runs_valid", np.uint8), ("rev_runs_valid", np.uint8), ("toehold_represents_end", np.uint8),
                             ("reserved", np.uint8)])
assert MOVE_RANGE_DTYPE.itemsize == 80


class OracleMoveIndex:
    """The restated b-move index (bmove/bmove.{h,cpp}) over the arrays of columba_amd.movebuild.build_move."""

    def __init__(self, mv, with_locate: bool = True):
        L = lib()
        self._keep = [np.ascontiguousarray(a) for a in (mv.lfbp_fwd, mv.lfbp_rev, mv.smpf, mv.smpl, mv.rev_smpf, mv.rev_smpl,
                                                        mv.pred_first, mv.first_to_run, mv.pred_last, mv.last_to_run, mv.plcp)]
        k = self._keep
        loc = [_p(a) for a in k[6:]] if with_locate else [None] * 5
        self.h = L.orc_move_create(_p(k[0]), k[0].nbytes, _p(k[1]), k[1].nbytes, _p(k[2]), _p(k[3]), _p(k[4]), _p(k[5]), *loc)
        if not self.h:
            raise ValueError("malformed move table")
        self.n = mv.n

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_move_destroy(self.h)
            self.h = None

    def complete_range(self) -> np.ndarray:
        out = np.zeros(1, dtype=MOVE_RANGE_DTYPE)
        lib().orc_move_complete_range(self.h, _p(out))
        return out

    def rows(self, rev: int) -> np.ndarray:
        r = lib().orc_move_rows(self.h, rev, None)
        out = np.zeros((r + 1, 4), dtype=np.uint64)
        lib().orc_move_rows(self.h, rev, _p(out))
        return out

    def extend(self, mode: int, parents: np.ndarray, c: np.ndarray):
        """children, ok flags and the number of table rows the reference's walks step over"""
        parents = np.ascontiguousarray(parents, dtype=MOVE_RANGE_DTYPE)
        c = np.ascontiguousarray(c, dtype=np.uint8)
        children = np.zeros(parents.shape[0], dtype=MOVE_RANGE_DTYPE)
        ok = np.zeros(parents.shape[0], dtype=np.uint8)
        steps = lib().orc_move_extend(self.h, mode, parents.shape[0], _p(parents), _p(c), _p(children), _p(ok))
        return children, ok, int(steps)

    def locate(self, rng: np.ndarray) -> np.ndarray:
        rng = np.ascontiguousarray(rng, dtype=MOVE_RANGE_DTYPE)
        cap = int(rng["end"][0] - rng["begin"][0]) + 8
        out = np.zeros(cap, dtype=np.uint64)
        cnt = lib().orc_move_locate(self.h, _p(rng), _p(out), cap)
        return out[:min(cnt, cap)]

    def match_exact(self, reads: Sequence[bytes]):
        """(occurrences n x {begin, end, distance, strand}, offsets, counters) in the reference's order"""
        buf, offs = pack_reads(reads)
        offs = offs.astype(np.uint64)
        o = np.zeros(len(reads) + 1, dtype=np.uint64)
        cnt = np.zeros(2, dtype=np.uint64)
        cap = 1 << 16
        while True:
            occ = np.zeros((cap, 4), dtype=np.uint64)
            tot = lib().orc_move_match_exact(self.h, buf.tobytes(), _p(offs), len(reads), _p(occ), cap, _p(o), _p(cnt))
            if tot <= cap:
                return occ[:tot], o, {"NODE_COUNTER": int(cnt[0]), "TOTAL_REPORTED_POSITIONS": int(cnt[1])}
            cap = int(tot)

    def kmer_table(self, word_size: int) -> np.ndarray:
        out = np.zeros(4 ** word_size, dtype=MOVE_RANGE_DTYPE)
        lib().orc_move_kmer_table(self.h, word_size, _p(out))
        return out

    def attach_text(self, text, seq_starts, word_size: int = 10):
        """the text beside the index, for match_best (CIGARs and trimming read the matched string of an occurrence from it);
        seq_starts as IndexArrays.seq_starts"""
        t = np.frombuffer(text, np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, np.uint8)
        st = np.ascontiguousarray(seq_starts, np.uint32)
        lib().orc_move_attach_text.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
        lib().orc_move_attach_text(self.h, word_size, _p(t), t.shape[0], _p(st), st.shape[0])

    def prepare(self, word_size: int = 10):
        """build the k-mer table of the search (index loading in the reference: not part of a timed match)"""
        lib().orc_move_prepare.argtypes = [C.c_void_p, C.c_uint32]
        lib().orc_move_prepare(self.h, word_size)

    def match_batch(self, strat: "OracleStrategy", k: int, reads: Sequence[bytes] = None, threads: int = 1, word_size: int = 10, packed=None):
        """SearchStrategy::matchApprox of the RUN_LENGTH_COMPRESSION flavour for a chunk of reads (ALL mode):
        (occurrences, offsets, counters) as match_batch of the FM-index flavour"""
        buf, offs = packed if packed is not None else pack_reads(reads)
        reads = range(offs.shape[0] - 1) if reads is None else reads
        if buf.shape[0] == 0:
            buf = np.zeros(1, np.uint8)
        r = lib().orc_move_match_batch(self.h, strat.h, k, _p(buf), _p(offs), len(reads), threads, word_size)
        try:
            err = lib().orc_result_error(r)
            if err:
                raise RuntimeError(err.decode())
            n = lib().orc_result_size(r)
            occs = np.zeros(max(n, 1), OCC_DTYPE)
            ro = np.zeros(len(reads) + 1, np.uint64)
            cnt = np.zeros(lib().orc_num_counters(), np.uint64)
            lib().orc_result_copy(r, _p(occs), _p(ro), _p(cnt))
        finally:
            lib().orc_result_free(r)
        return occs[:n], ro, dict(zip(COUNTER_NAMES, cnt.tolist()))
