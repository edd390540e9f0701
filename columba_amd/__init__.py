"""columba_amd — MI355X-native search-scheme FM-index matcher (hot path of biointec/columba).

Python host layer = plumbing around the C-ABI shared library ``libcolumba_amd.so``
(include/columba_amd.h): ctypes bindings, index container/builder for synthetic data, and the
read-sharding helper used by bench.py.  There is NO CPU fallback: every compute entry point goes to
the HIP library and raises if it (or a GPU) is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMB_LIB") or os.path.join(_HERE, "libcolumba_amd.so")  # (CMB_LIB: a variant build, for A/B runs)
_SRC = os.path.join(_HERE, "csrc", "columba_amd.hip")

CMB_OK = 0
CMB_ERR_INVALID, CMB_ERR_DEVICE, CMB_ERR_UNSUPPORTED, CMB_ERR_OVERFLOW, CMB_ERR_INTERNAL = -1, -2, -3, -4, -5
ERRORS = {-1: "CMB_ERR_INVALID", -2: "CMB_ERR_DEVICE", -3: "CMB_ERR_UNSUPPORTED", -4: "CMB_ERR_OVERFLOW",
          -5: "CMB_ERR_INTERNAL"}
COUNTER_NAMES = ["NODE_COUNTER", "TOTAL_REPORTED_POSITIONS", "IN_TEXT_STARTED", "ABORTED_IN_TEXT_VERIF",
                 "CIGARS_IN_TEXT_VERIFICATION", "IMMEDIATE_SWITCH", "SEARCH_STARTED", "EXPANSIONS", "LF_STEPS",
                 "LOCATED_ROWS", "TEXT_BYTES", "MATRIX_ROWS", "DFS_EXPANSIONS", "TABLE_ROWS", "DFS_TABLE_ROWS"]
METRIC = {"hamming": 0, "edit": 1}
PARTITION = {"uniform": 0, "static": 1, "dynamic": 2}
OCC_DTYPE = np.dtype([("begin", np.uint32), ("end", np.uint32), ("distance", np.uint32), ("strand", np.uint32)])
ALN_DTYPE = np.dtype([("seq_id", np.uint32), ("seq_begin", np.uint32), ("cigar_off", np.uint64), ("cigar_len", np.uint16),
                      ("spans", np.uint16), ("reserved", np.uint32)])


def cigar_string(ops) -> str:
    """run-length operations (length << 2 | op) -> "57M1I92M" """
    return "".join(f"{int(o) >> 2}{'MID'[int(o) & 3]}" for o in ops)

EXPORTS = [
    "cmb_index_create", "cmb_index_destroy", "cmb_index_device_bytes", "cmb_index_kmer_table",
    "cmb_index_layout_of", "cmb_index_seq_starts", "cmb_index_create_empty", "cmb_index_device_arrays", "cmb_index_validate",
    "cmb_strategy_create_named", "cmb_strategy_create_from_dir", "cmb_strategy_create",
    "cmb_strategy_add_scheme", "cmb_strategy_set_partition_params", "cmb_strategy_destroy",
    "cmb_strategy_describe", "cmb_strategy_export_scheme", "cmb_strategy_export_partition", "cmb_match_batch", "cmb_batch_create", "cmb_batch_run", "cmb_batch_stage_reads",
    "cmb_batch_result_size", "cmb_batch_results", "cmb_batch_timings", "cmb_batch_destroy",
    "cmb_rank_batch", "cmb_extend_batch", "cmb_extend_bench", "cmb_locate_batch", "cmb_verify_batch", "cmb_verify_batch_staged", "cmb_verify_window", "cmb_cigar_windows",
    "cmb_batch_want_alignments", "cmb_batch_alignments", "cmb_sam_se", "cmb_sam_se_xa", "cmb_sam_unmapped_se",
    "cmb_sam_pe", "cmb_sam_unpaired", "cmb_sam_unmapped_pe", "cmb_pair_sam", "cmb_pair_infer",
    "cmb_pair_best_create", "cmb_pair_best_set_trim", "cmb_pair_best_cutoff", "cmb_pair_best_seed", "cmb_pair_best_advance", "cmb_pair_best_supply", "cmb_pair_best_sam",
    "cmb_pair_best_destroy",
    "cmb_read_prepare", "cmb_batch_sam", "cmb_batch_filter_per_strand", "cmb_match_best", "cmb_best_sizes", "cmb_best_results",
    "cmb_best_destroy",
    "cmb_move_create", "cmb_move_destroy", "cmb_move_device_bytes", "cmb_move_info", "cmb_move_complete_range", "cmb_move_rows",
    "cmb_move_extend_batch", "cmb_move_extend_bench", "cmb_move_locate_batch", "cmb_move_match_exact", "cmb_move_last_timings", "cmb_move_kmer_table",
    "cmb_move_layout_of", "cmb_move_create_empty", "cmb_move_device_arrays", "cmb_move_validate",
    "cmb_batch_allow_unsupported", "cmb_batch_read_status", "cmb_trim_occurrence",
    "cmb_move_match_batch", "cmb_move_batch_create", "cmb_move_batch_run", "cmb_move_batch_result_size", "cmb_move_batch_results",
    "cmb_move_batch_timings", "cmb_move_batch_destroy", "cmb_move_attach_text", "cmb_move_text_index", "cmb_index_create_text_only", "cmb_sam_chunk", "cmb_move_batch_want_alignments",
    "cmb_move_batch_alignments", "cmb_move_batch_filter_per_strand", "cmb_move_match_best",
    "cmb_last_error", "cmb_version",
]


class CmbError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


def build_library(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU): one object per translation unit
    (the matcher; the b-move backend; the paired-end records; the pairing in BEST mode), linked into one shared library."""
    csrc = os.path.join(_HERE, "csrc")
    header = os.path.join(os.path.dirname(_HERE), "include", "columba_amd.h")
    units = {"columba_amd.hip": [f for f in os.listdir(csrc) if not f.startswith(("move_", "pair_"))],
             # (the b-move backend shares the event handler, the matrix and the wave helpers with the matcher)
             "move_backend.hip": [f for f in os.listdir(csrc) if not f.startswith("pair_") and f != "columba_amd.hip"],
             "pair_sam.hip": ["pair_sam.hip", "host_sam.hpp"],
             "pair_best.hip": ["pair_best.hip", "host_sam.hpp"]}
    every = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [header]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in every):
        return LIB_PATH  # (the objects are build scratch: only the library travels to the GPU box)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(_HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    objs, relink = [], force or not os.path.exists(LIB_PATH)
    for unit, deps in units.items():
        obj = os.path.join(objdir, unit.replace(".hip", ".o"))
        srcs = [os.path.join(csrc, f) for f in deps] + [header]
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(s) for s in srcs):
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                   "-Wno-unused-variable", "-c", "-o", obj, os.path.join(csrc, unit)])
            relink = True
        relink = relink or os.path.getmtime(obj) > os.path.getmtime(LIB_PATH)
        objs.append(obj)
    if relink:
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


class _IndexDesc(C.Structure):
    _fields_ = [
        ("text_length", C.c_uint64), ("text", C.c_void_p), ("counts", C.c_uint64 * 5),
        ("dollar_pos_fwd", C.c_uint64), ("bv_fwd", C.c_void_p), ("cnt_fwd", C.c_void_p),
        ("dollar_pos_rev", C.c_uint64), ("bv_rev", C.c_void_p), ("cnt_rev", C.c_void_p),
        ("sa_bv", C.c_void_p), ("sa_bv_counts", C.c_void_p), ("sa_samples", C.c_void_p),
        ("n_samples", C.c_uint64), ("sa_sparseness", C.c_uint32), ("seq_starts", C.c_void_p),
        ("n_seqs", C.c_uint32), ("kmer_size", C.c_uint32), ("in_text_switch", C.c_uint32),
    ]


class _MoveDesc(C.Structure):
    """cmb_move_desc (include/columba_amd.h)"""
    _fields_ = [
        ("lfbp", C.c_void_p), ("lfbp_bytes", C.c_uint64), ("rev_lfbp", C.c_void_p), ("rev_lfbp_bytes", C.c_uint64),
        ("length_bits", C.c_uint32),
        ("samples_first", C.c_void_p), ("samples_last", C.c_void_p), ("rev_samples_first", C.c_void_p),
        ("rev_samples_last", C.c_void_p),
        ("pred_first", C.c_void_p), ("first_to_run", C.c_void_p), ("pred_last", C.c_void_p), ("last_to_run", C.c_void_p),
        ("plcp_pos", C.c_void_p), ("plcp_sum", C.c_void_p), ("n_plcp", C.c_uint64),
    ]


MOVE_DEV_ARRAYS = 15


class MoveLayout(C.Structure):
    """cmb_move_layout (include/columba_amd.h): everything but the array contents of a device b-move index"""
    _fields_ = [("text_length", C.c_uint64), ("runs", C.c_uint64 * 2), ("zero_char_pos", C.c_uint64 * 2),
                ("set_count", C.c_uint64 * 3), ("set_shift", C.c_uint32 * 3), ("has_locate", C.c_uint32),
                ("bytes", C.c_uint64 * MOVE_DEV_ARRAYS)]


DEV_ARRAYS = 6


class IndexLayout(C.Structure):
    """cmb_index_layout (include/columba_amd.h): everything but the array contents of a device index"""
    _fields_ = [
        ("text_length", C.c_uint64), ("counts", C.c_uint64 * 5), ("dollar_pos_fwd", C.c_uint64),
        ("dollar_pos_rev", C.c_uint64), ("n_samples", C.c_uint64), ("sa_sparseness", C.c_uint32),
        ("kmer_size", C.c_uint32), ("in_text_switch", C.c_uint32), ("n_seqs", C.c_uint32),
        ("bytes", C.c_uint64 * DEV_ARRAYS),
    ]


class _DevArray:
    """a raw device allocation seen through __cuda_array_interface__ (torch.as_tensor wraps it without a copy)"""

    def __init__(self, ptr: int, nbytes: int, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._owner = owner


class PairOcc(C.Structure):
    """cmb_pair_occ"""
    _fields_ = [("seq_id", C.c_uint32), ("begin", C.c_uint32), ("end", C.c_uint32), ("index_begin", C.c_uint32),
                ("distance", C.c_uint32), ("strand", C.c_uint32), ("cigar_ops", C.c_void_p), ("n_ops", C.c_uint32)]


class PairRead(C.Structure):
    """cmb_pair_read"""
    _fields_ = [("id", C.c_char_p), ("seq", C.c_char_p), ("revcomp", C.c_char_p), ("qual", C.c_char_p), ("revqual", C.c_char_p),
                ("occ", C.POINTER(PairOcc)), ("n_occ", C.c_uint32)]


class PairParams(C.Structure):
    """cmb_pair_params"""
    _fields_ = [("orientation", C.c_uint32), ("max_frag", C.c_uint32), ("min_frag", C.c_uint32), ("discordant_allowed", C.c_int),
                ("unmapped_records", C.c_int)]


class PairInferred(C.Structure):
    """cmb_pair_inferred"""
    _fields_ = [("n_pairs", C.c_uint64), ("inferred", C.c_uint32), ("orientation", C.c_uint32), ("max_insert", C.c_uint32),
                ("min_insert", C.c_uint32), ("mean_insert", C.c_float), ("stddev_insert", C.c_float)]


ORIENTATION_FR, ORIENTATION_RF, ORIENTATION_FF = 0, 1, 2


def pair_infer(samples) -> "PairInferred":
    """orientation and insert-size bounds from unambiguously mapped pairs; samples: n x (begin1, end1, strand1, begin2, end2, strand2)"""
    a = np.ascontiguousarray(samples, dtype=np.uint32).reshape(-1, 6)
    out = PairInferred()
    _chk(lib().cmb_pair_infer(_p(a), a.shape[0], C.byref(out)))
    return out


class SamHit(C.Structure):
    """cmb_sam_hit"""
    _fields_ = [("seq_name", C.c_char_p), ("pos0", C.c_uint32), ("distance", C.c_uint32), ("revcomp", C.c_uint32),
                ("cigar_ops", C.c_void_p), ("n_ops", C.c_uint32)]


_lib = None


def lib():
    """Load libcolumba_amd.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (the product has no CPU fallback)")
        # torch bundles its own HIP runtime: load it first so that this process ends up with ONE
        # libamdhip64 (loading ours first makes the second runtime see no devices)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.cmb_last_error.restype = C.c_char_p
        L.cmb_version.restype = C.c_char_p
        L.cmb_index_create.argtypes = [C.POINTER(_IndexDesc), i32, C.POINTER(vp)]
        L.cmb_index_destroy.argtypes = [vp]
        L.cmb_index_device_bytes.restype = u64
        L.cmb_index_device_bytes.argtypes = [vp]
        L.cmb_index_kmer_table.argtypes = [vp, vp]
        L.cmb_index_layout_of.argtypes = [vp, C.POINTER(IndexLayout)]
        L.cmb_index_seq_starts.argtypes = [vp, vp]
        L.cmb_index_create_empty.argtypes = [C.POINTER(IndexLayout), vp, i32, C.POINTER(vp)]
        L.cmb_index_device_arrays.argtypes = [vp, C.POINTER(vp * DEV_ARRAYS), C.POINTER(u64 * DEV_ARRAYS)]
        L.cmb_index_validate.argtypes = [vp]
        L.cmb_strategy_create_named.argtypes = [C.c_char_p, i32, i32, C.POINTER(vp)]
        L.cmb_strategy_create_from_dir.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(vp)]
        L.cmb_strategy_create.argtypes = [i32, i32, u32, C.POINTER(vp)]
        L.cmb_strategy_add_scheme.argtypes = [vp, u32, u32, u32, vp, vp, vp]
        L.cmb_strategy_set_partition_params.argtypes = [vp, u32, vp, u32, vp, u32, vp, u32]
        L.cmb_strategy_destroy.argtypes = [vp]
        L.cmb_strategy_describe.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32), vp, u32]
        L.cmb_strategy_export_scheme.argtypes = [vp, u32, u32, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]
        L.cmb_strategy_export_partition.argtypes = [vp, u32, vp, vp, vp, u32, C.POINTER(u32)]
        L.cmb_match_batch.argtypes = [vp, vp, u32, vp, vp, u32, vp, u64, vp, vp, C.POINTER(u64)]
        L.cmb_batch_create.argtypes = [vp, vp, u32, vp, vp, u32, C.POINTER(vp)]
        L.cmb_batch_run.argtypes = [vp]
        L.cmb_batch_stage_reads.argtypes = [vp, vp, vp, u32]
        L.cmb_batch_result_size.argtypes = [vp, C.POINTER(u64)]
        L.cmb_batch_results.argtypes = [vp, vp, u64, vp, vp]
        L.cmb_batch_timings.argtypes = [vp, vp, vp, u32]
        L.cmb_batch_destroy.argtypes = [vp]
        L.cmb_rank_batch.argtypes = [vp, i32, vp, vp, u64, vp]
        L.cmb_extend_batch.argtypes = [vp, i32, vp, u64, vp, vp]
        L.cmb_extend_bench.argtypes = [vp, i32, vp, u64, vp, vp, u32, C.POINTER(C.c_float)]
        L.cmb_locate_batch.argtypes = [vp, vp, u64, vp, C.POINTER(u64)]
        L.cmb_verify_batch.argtypes = [vp, C.c_char_p, u32, vp, u64, u32, u32, i32, vp, u64, C.POINTER(u64), vp]
        L.cmb_verify_batch_staged.argtypes = L.cmb_verify_batch.argtypes
        L.cmb_cigar_windows.argtypes = [vp, C.c_char_p, u32, vp, vp, vp, u64, vp, u32, vp]
        L.cmb_verify_window.argtypes = [vp, C.c_char_p, u32, u32, u32, u32, u32, vp, u64, C.POINTER(u64), vp]
        L.cmb_sam_se.restype = C.c_int64
        L.cmb_sam_se.argtypes = [C.c_char_p, C.POINTER(SamHit), i32, u32, u32, C.c_char_p, C.c_char_p, vp, u64]
        L.cmb_sam_se_xa.restype = C.c_int64
        L.cmb_sam_se_xa.argtypes = [C.c_char_p, C.POINTER(SamHit), u32, u32, C.c_char_p, C.c_char_p, vp, u64]
        L.cmb_sam_pe.restype = C.c_int64
        L.cmb_sam_pe.argtypes = [C.c_char_p, C.POINTER(SamHit), i32, C.POINTER(SamHit), u32, u32, u32, i32, i32, C.c_char_p, C.c_char_p, vp, u64]
        L.cmb_sam_unpaired.restype = C.c_int64
        L.cmb_sam_unpaired.argtypes = [C.c_char_p, C.POINTER(SamHit), i32, u32, u32, i32, C.c_char_p, C.c_char_p, vp, u64]
        L.cmb_sam_unmapped_pe.restype = C.c_int64
        L.cmb_sam_unmapped_pe.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, i32, i32, i32, vp, u64]
        L.cmb_pair_infer.argtypes = [vp, u64, C.POINTER(PairInferred)]
        L.cmb_pair_best_create.argtypes = [C.POINTER(PairParams), u32, u32, u32, i32, vp, u32, C.POINTER(PairRead), C.POINTER(PairRead), C.POINTER(vp)]
        L.cmb_pair_best_set_trim.argtypes = [vp, vp, vp]
        L.cmb_pair_best_cutoff.argtypes = [vp, u32, u32, C.POINTER(u32)]
        L.cmb_pair_best_seed.argtypes = [vp, u32, vp, vp, u64, vp, vp, vp, u64, vp, i32]
        L.cmb_pair_best_advance.argtypes = [vp, vp, u64, C.POINTER(u64)]
        L.cmb_pair_best_supply.argtypes = [vp, u32, u32, u32, u32, vp, vp, u64, vp]
        L.cmb_pair_best_sam.restype = C.c_int64
        L.cmb_pair_best_sam.argtypes = [vp, u32, vp, vp, u64, C.POINTER(u32)]
        L.cmb_pair_best_destroy.argtypes = [vp]
        L.cmb_pair_best_destroy.restype = None
        L.cmb_pair_sam.restype = C.c_int64
        L.cmb_pair_sam.argtypes = [C.POINTER(PairParams), C.POINTER(PairRead), C.POINTER(PairRead), vp, vp, u64, C.POINTER(u32)]
        L.cmb_sam_unmapped_se.restype = C.c_int64
        L.cmb_sam_unmapped_se.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, vp, u64]
        L.cmb_read_prepare.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, vp, vp, vp, vp]
        L.cmb_batch_sam.restype = C.c_int64
        L.cmb_batch_sam.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, u64]
        L.cmb_batch_filter_per_strand.argtypes = [vp, i32]
        L.cmb_trim_occurrence.argtypes = [vp, vp, u32, u32, i32, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(i32)]
        L.cmb_batch_allow_unsupported.argtypes = [vp, i32]
        L.cmb_batch_read_status.argtypes = [vp, vp, C.POINTER(u32)]
        L.cmb_match_best.argtypes = [vp, vp, u32, u32, vp, vp, u32, C.POINTER(vp)]
        L.cmb_best_sizes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.cmb_best_results.argtypes = [vp, vp, vp, u64, vp, u64, vp, vp, vp, vp]
        L.cmb_best_destroy.argtypes = [vp]
        L.cmb_batch_want_alignments.argtypes = [vp, i32]
        L.cmb_batch_alignments.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
        L.cmb_move_create.argtypes = [C.POINTER(_MoveDesc), i32, C.POINTER(vp)]
        L.cmb_move_destroy.argtypes = [vp]
        L.cmb_move_device_bytes.restype = u64
        L.cmb_move_device_bytes.argtypes = [vp]
        L.cmb_move_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
        L.cmb_move_complete_range.argtypes = [vp, vp]
        L.cmb_move_rows.argtypes = [vp, i32, u64, u64, vp]
        L.cmb_move_extend_batch.argtypes = [vp, i32, vp, u64, vp, vp]
        L.cmb_move_extend_bench.argtypes = [vp, i32, vp, u64, vp, vp, u32, C.POINTER(C.c_float)]
        L.cmb_move_locate_batch.argtypes = [vp, vp, u64, vp, vp]
        L.cmb_move_match_exact.argtypes = [vp, vp, vp, u64, vp, u64, vp, C.POINTER(u64), vp]
        L.cmb_move_last_timings.argtypes = [vp, u32]
        L.cmb_move_match_batch.argtypes = [vp, vp, u32, u32, vp, vp, u32, vp, u64, vp, vp, C.POINTER(u64)]
        L.cmb_move_batch_create.argtypes = [vp, vp, u32, u32, vp, vp, u32, C.POINTER(vp)]
        L.cmb_move_batch_run.argtypes = [vp]
        L.cmb_move_batch_result_size.argtypes = [vp, C.POINTER(u64)]
        L.cmb_move_batch_results.argtypes = [vp, vp, u64, vp, vp]
        L.cmb_move_batch_timings.argtypes = [vp, vp, vp, u32]
        L.cmb_move_batch_destroy.argtypes = [vp]
        L.cmb_move_batch_destroy.restype = None
        L.cmb_move_attach_text.argtypes = [vp, vp, u64, vp, u32]
        L.cmb_sam_chunk.argtypes = [vp, u32, i32, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, u64]
        L.cmb_sam_chunk.restype = C.c_int64
        L.cmb_move_text_index.argtypes = [vp]
        L.cmb_move_text_index.restype = vp
        L.cmb_index_create_text_only.argtypes = [vp, u64, vp, u32, i32, C.POINTER(vp)]
        L.cmb_move_batch_want_alignments.argtypes = [vp, i32]
        L.cmb_move_batch_filter_per_strand.argtypes = [vp, i32]
        L.cmb_move_match_best.argtypes = [vp, vp, u32, u32, u32, vp, vp, u32, C.POINTER(vp)]
        L.cmb_move_batch_alignments.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
        L.cmb_move_kmer_table.argtypes = [vp, u32, vp]
        L.cmb_move_layout_of.argtypes = [vp, C.POINTER(MoveLayout)]
        L.cmb_move_create_empty.argtypes = [C.POINTER(MoveLayout), i32, C.POINTER(vp)]
        L.cmb_move_device_arrays.argtypes = [vp, C.POINTER(vp * MOVE_DEV_ARRAYS), C.POINTER(u64 * MOVE_DEV_ARRAYS)]
        L.cmb_move_validate.argtypes = [vp]
        _lib = L
    return _lib


def _chk(rc: int):
    if rc != CMB_OK:
        raise CmbError(rc, lib().cmb_last_error().decode(errors="replace"))


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Index:
    """Device-resident bidirectional FM-index (handle of ``cmb_index_create``).

    Mirrors the constructor of the reference's FMIndex (src/fmindex/fmindex.h:403):
    ``Index(arrays, in_text_switch=4, sa_sparse=arrays.sparseness, kmer_size=10)``.
    """

    def __init__(self, ix, in_text_switch: int = 4, kmer_size: int = 10, device: int = 0, _handle=None):
        self.device = device
        if _handle is not None:  # (Index.empty_like)
            self.h, self.kmer_size, self.n = _handle, kmer_size, ix
            return
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
        d.sa_bv = _p(ix.sa_bv)
        d.sa_bv_counts = _p(ix.sa_bv_counts)
        d.sa_samples = _p(ix.sa_samples)
        d.n_samples = ix.sa_samples.shape[0]
        d.sa_sparseness = ix.sparseness
        starts = np.ascontiguousarray(ix.seq_starts, np.uint32)
        d.seq_starts = _p(starts)
        d.n_seqs = starts.shape[0]
        d.kmer_size = kmer_size
        d.in_text_switch = in_text_switch
        self.kmer_size = kmer_size
        self.n = ix.n
        h = C.c_void_p()
        _chk(lib().cmb_index_create(C.byref(d), device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.cmb_index_destroy(self.h)
            self.h = None

    __del__ = close

    def device_bytes(self) -> int:
        return int(lib().cmb_index_device_bytes(self.h))

    # ---- replication on other GPUs (columba_amd/dist.py: broadcast_device_index)
    def layout(self) -> IndexLayout:
        lay = IndexLayout()
        _chk(lib().cmb_index_layout_of(self.h, C.byref(lay)))
        return lay

    def seq_starts(self) -> np.ndarray:
        out = np.zeros(max(int(self.layout().n_seqs), 1), np.uint32)
        _chk(lib().cmb_index_seq_starts(self.h, _p(out)))
        return out[:int(self.layout().n_seqs)]

    @classmethod
    def empty_like(cls, layout: IndexLayout, seq_starts: np.ndarray, device: int = 0) -> "Index":
        """an index with the arrays of `layout` allocated but not filled (the receiving side of a broadcast)"""
        starts = np.ascontiguousarray(seq_starts, np.uint32)
        h = C.c_void_p()
        _chk(lib().cmb_index_create_empty(C.byref(layout), _p(starts), device, C.byref(h)))
        return cls(int(layout.text_length), kmer_size=int(layout.kmer_size), device=device, _handle=h)

    def validate(self):
        """consistency probe of the device arrays (after they were filled by a collective)"""
        _chk(lib().cmb_index_validate(self.h))

    def device_tensors(self):
        """the device arrays of the index as flat uint8 torch tensors sharing the library's memory (no copy)"""
        import torch
        ptrs, nbytes = (C.c_void_p * DEV_ARRAYS)(), (C.c_uint64 * DEV_ARRAYS)()
        _chk(lib().cmb_index_device_arrays(self.h, C.byref(ptrs), C.byref(nbytes)))
        out = []
        for i in range(DEV_ARRAYS):
            if not nbytes[i]:
                out.append(None)
                continue
            out.append(torch.as_tensor(_DevArray(int(ptrs[i]), int(nbytes[i]), self), device=torch.device("cuda", self.device)))
        return out

    def kmer_table(self) -> np.ndarray:
        out = np.zeros((4 ** self.kmer_size, 4), np.uint32)
        _chk(lib().cmb_index_kmer_table(self.h, _p(out)))
        return out

    def rank(self, rev: int, c, p) -> np.ndarray:
        c = np.ascontiguousarray(c, np.uint32)
        p = np.ascontiguousarray(p, np.uint64)
        out = np.zeros(p.shape[0], np.uint64)
        _chk(lib().cmb_rank_batch(self.h, rev, _p(c), _p(p), p.shape[0], _p(out)))
        return out

    def extend(self, mode: int, ranges) -> Tuple[np.ndarray, np.ndarray]:
        r = np.ascontiguousarray(ranges, np.uint32).reshape(-1, 4)
        out = np.zeros((r.shape[0], 4, 4), np.uint32)
        ok = np.zeros((r.shape[0], 4), np.uint8)
        _chk(lib().cmb_extend_batch(self.h, mode, _p(r), r.shape[0], _p(out), _p(ok)))
        return out, ok

    def locate(self, rows) -> Tuple[np.ndarray, int]:
        rows = np.ascontiguousarray(rows, np.uint32)
        out = np.zeros(rows.shape[0], np.uint32)
        lf = C.c_uint64()
        _chk(lib().cmb_locate_batch(self.h, _p(rows), rows.shape[0], _p(out), C.byref(lf)))
        return out, int(lf.value)

    def verify_window(self, pattern: bytes, start: int, end: int, max_ed: int, min_ed: int = 0):
        """FMIndex::inTextVerificationOneString: the pattern against text[start, end), fixed start"""
        out = np.zeros(64, OCC_DTYPE)
        n = C.c_uint64()
        cnt = np.zeros(len(COUNTER_NAMES), np.uint64)
        _chk(lib().cmb_verify_window(self.h, pattern, len(pattern), start, end, max_ed, min_ed, _p(out), 64, C.byref(n), _p(cnt)))
        return out[:n.value], dict(zip(COUNTER_NAMES, cnt.tolist()))

    def verify(self, pattern: bytes, starts, max_ed: int, min_ed: int, fixed: bool, staged: bool = False):
        """FMIndex::inTextVerification for one pattern; staged=True: through the production edit-distance path"""
        starts = np.ascontiguousarray(starts, np.uint32)
        cap = starts.shape[0] * 32 + 64
        out = np.zeros(cap, OCC_DTYPE)
        n = C.c_uint64()
        cnt = np.zeros(len(COUNTER_NAMES), np.uint64)
        fn = lib().cmb_verify_batch_staged if staged else lib().cmb_verify_batch
        _chk(fn(self.h, pattern, len(pattern), _p(starts), starts.shape[0], max_ed, min_ed,
                                    int(fixed), _p(out), cap, C.byref(n), _p(cnt)))
        return out[:n.value], dict(zip(COUNTER_NAMES, cnt.tolist()))


class SearchStrategy:
    """Search schemes + partitioning (handle of ``cmb_strategy_*``).

    ``SearchStrategy("kuch1" | "kuch2" | "kianfar" | "01*0" | "pigeon" | "minU" | "columba" | "multiple_opt",
    metric, partition)`` mirrors ``Parameters::createStrategy`` (src/parameters/alignparameters.cpp:1313-1376);
    ``SearchStrategy.from_dir(path, mode)`` the ``-d`` (mode "multiple"), ``-c -nD`` ("custom") and ``-c``
    ("custom_dynamic") options.
    """
    DIR_MODES = {"custom": 0, "multiple": 1, "custom_dynamic": 2, False: 0, True: 1}

    def __init__(self, name: Optional[str] = "multiple_opt", metric: str = "edit", partition: str = "dynamic",
                 _handle=None):
        if _handle is not None:
            self.h = _handle
            return
        h = C.c_void_p()
        _chk(lib().cmb_strategy_create_named(name.encode(), METRIC[metric], PARTITION[partition], C.byref(h)))
        self.h = h

    @classmethod
    def from_dir(cls, path: str, multiple="multiple", metric: str = "edit", partition: str = "dynamic"):
        h = C.c_void_p()
        _chk(lib().cmb_strategy_create_from_dir(path.encode(), cls.DIR_MODES[multiple], METRIC[metric],
                                                PARTITION[partition], C.byref(h)))
        return cls(_handle=h)

    @classmethod
    def from_tables(cls, spec: Dict, metric: str = "edit", partition: str = "dynamic"):
        """spec = {"kmer_cutoff": int, "schemes": {k: [[(pi, L, U), ...], ...]}, "partition_params": {...}}"""
        h = C.c_void_p()
        _chk(lib().cmb_strategy_create(METRIC[metric], PARTITION[partition], spec.get("kmer_cutoff", 20), C.byref(h)))
        self = cls(_handle=h)
        for k, schemes in spec["schemes"].items():
            for sch in schemes:
                pi = np.ascontiguousarray([s[0] for s in sch], np.uint32)
                lo = np.ascontiguousarray([s[1] for s in sch], np.uint32)
                up = np.ascontiguousarray([s[2] for s in sch], np.uint32)
                _chk(lib().cmb_strategy_add_scheme(h, k, pi.shape[0], pi.shape[1], _p(pi), _p(lo), _p(up)))
        for k, pp in spec.get("partition_params", {}).items():
            seed = np.ascontiguousarray(pp.get("seeding", []), np.float64)
            w = np.ascontiguousarray(pp.get("weights", []), np.uint64)
            b = np.ascontiguousarray(pp.get("begins", []), np.float64)
            _chk(lib().cmb_strategy_set_partition_params(h, k, _p(seed), seed.shape[0], _p(w), w.shape[0], _p(b),
                                                         b.shape[0]))
        return self

    def describe(self, k: int):
        ns, npart = C.c_uint32(), C.c_uint32()
        crit = np.zeros(8, np.uint32)
        _chk(lib().cmb_strategy_describe(self.h, k, C.byref(ns), C.byref(npart), _p(crit), 8))
        return ns.value, npart.value, crit[:ns.value].tolist()

    def scheme(self, k: int, idx: int):
        """searches of alternative ``idx`` for distance k as a list of (pi, L, U) (host-only, no GPU needed)"""
        ns, npart = C.c_uint32(), C.c_uint32()
        cap = 32 * 32
        pi, lo, up = (np.zeros(cap, np.uint32) for _ in range(3))
        _chk(lib().cmb_strategy_export_scheme(self.h, k, idx, _p(pi), _p(lo), _p(up), cap, C.byref(ns), C.byref(npart)))
        n, p = ns.value, npart.value
        return [(pi[i * p:(i + 1) * p].tolist(), lo[i * p:(i + 1) * p].tolist(), up[i * p:(i + 1) * p].tolist())
                for i in range(n)]

    def partition_params(self, k: int):
        """(seeding positions, weights, begin positions, k-mer cut-off) in force for distance k"""
        _, p, _ = self.describe(k)
        seed, w, b = np.zeros(32, np.float64), np.zeros(32, np.uint64), np.zeros(32, np.float64)
        cut = C.c_uint32()
        _chk(lib().cmb_strategy_export_partition(self.h, k, _p(seed), _p(w), _p(b), 32, C.byref(cut)))
        return seed[:max(p - 2, 0)].tolist(), w[:p].tolist(), b[:p - 1].tolist(), cut.value

    def supports(self, k: int) -> bool:
        ns = C.c_uint32()
        return lib().cmb_strategy_describe(self.h, k, C.byref(ns), None, None, 0) == CMB_OK

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.cmb_strategy_destroy(self.h)
            self.h = None

    __del__ = close


def parse_cigar(cigar: str) -> np.ndarray:
    """"57M1I92M" -> run-length operations (length << 2 | op)"""
    import re
    return np.array([(int(n) << 2) | "MID".index(o) for n, o in re.findall(r"(\d+)([MID])", cigar)], np.uint16)


def _sam_hit(seq_name: bytes, pos0: int, distance: int, revcomp: bool, ops: np.ndarray):
    ops = np.ascontiguousarray(ops, np.uint16)
    h = SamHit(seq_name, pos0, distance, int(revcomp), ops.ctypes.data if ops.size else None, ops.shape[0])
    h._keep = (seq_name, ops)
    return h


def _sam_call(fn, *args) -> str:
    n = fn(*args, None, 0)
    if n < 0:
        raise CmbError(int(n), lib().cmb_last_error().decode(errors="replace"))
    buf = C.create_string_buffer(int(n) + 1)
    fn(*args, buf, int(n) + 1)
    return buf.value.decode()


def sam_se(read_id: str, hit, primary: bool, n_hits: int, min_score: int, seq: str, qual: str) -> str:
    """one SAM line (TextOcc::generateSAMSingleEnd); hit = (seq_name, pos0, distance, revcomp, cigar ops)"""
    h = _sam_hit(hit[0].encode(), *hit[1:])
    return _sam_call(lib().cmb_sam_se, read_id.encode(), C.byref(h), int(primary), n_hits, min_score, seq.encode(), qual.encode())


def sam_se_xa(read_id: str, hits, n_hits: int, seq: str, qual: str) -> str:
    hs = [_sam_hit(h[0].encode(), *h[1:]) for h in hits]
    arr = (SamHit * len(hs))(*hs)
    return _sam_call(lib().cmb_sam_se_xa, read_id.encode(), arr, len(hs), n_hits, seq.encode(), qual.encode())


def sam_unmapped_se(read_id: str, seq: str, qual: str) -> str:
    return _sam_call(lib().cmb_sam_unmapped_se, read_id.encode(), seq.encode(), qual.encode())


def sam_pe(read_id: str, hit, first_in_pair: bool, mate, n_pairs: int, min_score: int, frag_size: int, discordant: bool, primary: bool,
           seq: str, qual: str) -> str:
    """one record of a paired read (TextOcc::generateSAMPairedEnd); mate = None: the mate is not mapped"""
    h = _sam_hit(hit[0].encode(), *hit[1:])
    m = _sam_hit(mate[0].encode(), *mate[1:]) if mate is not None else None
    return _sam_call(lib().cmb_sam_pe, read_id.encode(), C.byref(h), int(first_in_pair), C.byref(m) if m is not None else None, n_pairs,
                     min_score, frag_size, int(discordant), int(primary), seq.encode(), qual.encode())


def sam_unpaired(read_id: str, hit, first_in_pair: bool, n_hits: int, min_score: int, primary: bool, seq: str, qual: str) -> str:
    h = _sam_hit(hit[0].encode(), *hit[1:])
    return _sam_call(lib().cmb_sam_unpaired, read_id.encode(), C.byref(h), int(first_in_pair), n_hits, min_score, int(primary),
                     seq.encode(), qual.encode())


def sam_unmapped_pe(read_id: str, seq: str, qual: str, first_in_pair: bool, mate_mapped: bool, mate_revcomp: bool) -> str:
    return _sam_call(lib().cmb_sam_unmapped_pe, read_id.encode(), seq.encode(), qual.encode(), int(first_in_pair), int(mate_mapped),
                     int(mate_revcomp))


def pair_sam(read1, read2, seq_names, orientation: int = ORIENTATION_FR, max_frag: int = 500, min_frag: int = 0,
             discordant_allowed: bool = True, unmapped_records: bool = True):
    """SAM text of one read pair in ALL mode (cmb_pair_sam).  read = (id, seq, revcomp, qual, revqual, occurrences) with
    occurrences = [(seq_id or None, begin, end, index_begin, distance, strand, cigar ops)]; returns (text, TOTAL_UNIQUE_PAIRS)."""
    keep = []

    def mk(rd):
        rid, seq, rc, qual, rq, occs = rd
        arr = (PairOcc * max(len(occs), 1))()
        for i, (sid, b, e, ib, d, st, ops) in enumerate(occs):
            o = np.ascontiguousarray(ops, dtype=np.uint16)
            keep.append(o)
            arr[i] = PairOcc(0xFFFFFFFF if sid is None else sid, b, e, ib, d, st, o.ctypes.data, o.shape[0])
        keep.append(arr)
        return PairRead(rid.encode(), seq.encode(), rc.encode(), qual.encode(), rq.encode(), arr, len(occs))

    r1, r2 = mk(read1), mk(read2)
    names = (C.c_char_p * len(seq_names))(*[n.encode() for n in seq_names])
    prm = PairParams(orientation, max_frag, min_frag, int(discordant_allowed), int(unmapped_records))
    n_pairs = C.c_uint32()
    n = lib().cmb_pair_sam(C.byref(prm), C.byref(r1), C.byref(r2), names, None, 0, C.byref(n_pairs))
    if n < 0:
        _chk(int(n))
    buf = C.create_string_buffer(int(n) + 1)
    lib().cmb_pair_sam(C.byref(prm), C.byref(r1), C.byref(r2), names, buf, int(n) + 1, C.byref(n_pairs))
    return buf.value.decode(), int(n_pairs.value)


PAIR_REQUEST_DTYPE = np.dtype([("pair", np.uint32), ("mate", np.uint32), ("strand", np.uint32), ("max_distance", np.uint32)])  # cmb_pair_request
PAIR_TRIM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                           C.POINTER(C.c_uint32))


class PairBest:
    """A chunk of read pairs walking through their strata in BEST (+x) mode (cmb_pair_best_*: SearchStrategy::matchApproxPairedEndBestPlusX,
    searchstrategy.cpp:1091-1179).  reads1 / reads2: per pair (id, seq, revcomp, qual, revqual) as read_prepare leaves them.
    advance() -> requests (PAIR_REQUEST_DTYPE) for the lists the unfinished pairs wait for; supply(...) hands one in; sam(i, names)."""

    def __init__(self, reads1, reads2, x: int, min_identity: int, max_supported: int, orientation: int = ORIENTATION_FR, max_frag: int = 500,
                 min_frag: int = 0, discordant_allowed: bool = True, unmapped_records: bool = True, metric: str = "edit", text_index=None,
                 trim=None):
        n = len(reads1)
        assert len(reads2) == n
        self._keep = []

        def mk(rds):
            arr = (PairRead * max(n, 1))()
            for i, (rid, seq, rc, qual, rq) in enumerate(rds):
                arr[i] = PairRead(rid.encode(), seq.encode(), rc.encode(), qual.encode(), rq.encode(), None, 0)
            return arr

        a1, a2 = mk(reads1), mk(reads2)
        prm = PairParams(orientation, max_frag, min_frag, int(discordant_allowed), int(unmapped_records))
        self.h = C.c_void_p()
        self.n = n
        _chk(lib().cmb_pair_best_create(C.byref(prm), x, min_identity, max_supported, METRIC[metric], text_index.h if text_index is not None else None,
                                        n, a1, a2, C.byref(self.h)))
        if trim is not None:  # trim(pair, mate, strand, largest_stratum, (begin, end, distance)) -> None or (begin, end, distance, seq_id, seq_begin, ops)
            def hook(_user, pair, mate, strand, stratum, occ_p, aln_p, ops_p, ops_cap, n_ops_p):
                oc = np.ctypeslib.as_array(C.cast(occ_p, C.POINTER(C.c_uint32)), shape=(4,))
                res = trim(int(pair), int(mate), int(strand), int(stratum), (int(oc[0]), int(oc[1]), int(oc[2])))
                if res is None:
                    return 0
                b, e, d, sid, sb, ops = res
                oc[0], oc[1], oc[2] = b, e, d
                al = np.ctypeslib.as_array(C.cast(aln_p, C.POINTER(C.c_uint32)), shape=(2,))
                al[0], al[1] = sid, sb
                out = np.ctypeslib.as_array(C.cast(ops_p, C.POINTER(C.c_uint16)), shape=(int(ops_cap),))
                out[:len(ops)] = ops
                n_ops_p[0] = len(ops)
                return 1
            self._hook = PAIR_TRIM_FN(hook)
            _chk(lib().cmb_pair_best_set_trim(self.h, C.cast(self._hook, C.c_void_p), None))

    def cutoff(self, pair: int, mate: int) -> int:
        v = C.c_uint32()
        _chk(lib().cmb_pair_best_cutoff(self.h, pair, mate, C.byref(v)))
        return int(v.value)

    def seed(self, pair: int, se1, se2, read2_done: bool):
        """start pair `pair` from the mates' single-end BEST results (cmb_pair_best_seed): se = (occ, aln, ops)"""
        a = [np.ascontiguousarray(se1[0], dtype=OCC_DTYPE), np.ascontiguousarray(se1[1], dtype=ALN_DTYPE), np.ascontiguousarray(se1[2], dtype=np.uint16),
             np.ascontiguousarray(se2[0], dtype=OCC_DTYPE), np.ascontiguousarray(se2[1], dtype=ALN_DTYPE), np.ascontiguousarray(se2[2], dtype=np.uint16)]
        ptr = [(_p(x) if x.shape[0] else None) for x in a]
        _chk(lib().cmb_pair_best_seed(self.h, pair, ptr[0], ptr[1], a[0].shape[0], ptr[2], ptr[3], ptr[4], a[3].shape[0], ptr[5], int(read2_done)))

    def advance(self) -> np.ndarray:
        req = np.zeros(max(self.n, 1), dtype=PAIR_REQUEST_DTYPE)
        n = C.c_uint64()
        _chk(lib().cmb_pair_best_advance(self.h, _p(req), req.shape[0], C.byref(n)))
        return req[:int(n.value)]

    def supply(self, pair: int, mate: int, strand: int, max_distance: int, occ: np.ndarray, aln: np.ndarray, ops: np.ndarray):
        occ = np.ascontiguousarray(occ, dtype=OCC_DTYPE)
        aln = np.ascontiguousarray(aln, dtype=ALN_DTYPE)
        ops = np.ascontiguousarray(ops, dtype=np.uint16)
        _chk(lib().cmb_pair_best_supply(self.h, pair, mate, strand, max_distance, _p(occ) if occ.shape[0] else None, _p(aln) if aln.shape[0] else None,
                                        occ.shape[0], _p(ops) if ops.shape[0] else None))

    def sam(self, pair: int, seq_names):
        names = (C.c_char_p * len(seq_names))(*[n.encode() for n in seq_names])
        n_pairs = C.c_uint32()
        n = lib().cmb_pair_best_sam(self.h, pair, names, None, 0, C.byref(n_pairs))
        if n < 0:
            _chk(int(n))
        buf = C.create_string_buffer(int(n) + 1)
        lib().cmb_pair_best_sam(self.h, pair, names, buf, int(n) + 1, C.byref(n_pairs))
        return buf.value.decode(), int(n_pairs.value)

    def close(self):
        if getattr(self, "h", None):
            lib().cmb_pair_best_destroy(self.h)
            self.h = None

    __del__ = close


def read_prepare(read_id: str, seq: str, qual: str = ""):
    """(identifier, cleaned read, reverse complement, reversed quality) as Read / ReadBundle hold them (reads.h)"""
    bufs = [C.create_string_buffer(len(x.encode()) + 2) for x in (read_id, seq, seq, qual)]
    _chk(lib().cmb_read_prepare(read_id.encode(), seq.encode(), qual.encode(), *bufs))
    return tuple(b.value.decode() for b in bufs)


def pack_reads(reads: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    offs = np.zeros(len(reads) + 1, np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads])
    buf = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    if buf.shape[0] == 0:
        buf = np.zeros(1, np.uint8)
    return buf, offs


class Batch:
    """A batch of reads resident in HBM (handle of ``cmb_batch_*``)."""

    def __init__(self, index: Index, strategy: SearchStrategy, max_distance: int, reads=None, packed=None):
        buf, offs = packed if packed is not None else pack_reads(reads)
        self.n_reads = offs.shape[0] - 1
        self._keep = (index, strategy)
        self._packed = (buf, offs)
        h = C.c_void_p()
        _chk(lib().cmb_batch_create(index.h, strategy.h, max_distance, _p(buf), _p(offs), self.n_reads, C.byref(h)))
        self.h = h

    def run(self):
        # a run consumes the staged chunk (cmb_batch_run waits for its copy first): from here on the batch's reads ARE that
        # chunk — sam() must see them — and the chunk before it may still be the source of nothing
        staged = getattr(self, "_staged", None)
        if staged is not None:
            self._retired = self._packed  # (one more generation alive: its upload finished before this run started)
            self._packed = staged
            self._staged = None
        _chk(lib().cmb_batch_run(self.h))

    def stage(self, packed):
        """copy the next chunk (buf, offs: same number of reads) to the device while the current one is matched"""
        buf, offs = packed
        # a chunk staged earlier and never run may still be the source of a copy in flight: keep it alive as well
        if getattr(self, "_staged", None) is not None:
            self._superseded = self._staged
        self._staged = (buf, offs)
        _chk(lib().cmb_batch_stage_reads(self.h, _p(buf), _p(offs), offs.shape[0] - 1))

    def results(self, reuse: bool = False):
        """(occurrences, per-read offsets, counters); reuse=True hands out views of buffers the Batch keeps between
        calls (a streaming caller consumes a chunk's results before the next run)"""
        n = C.c_uint64()
        _chk(lib().cmb_batch_result_size(self.h, C.byref(n)))
        need = max(int(n.value), 1)
        if reuse and getattr(self, "_res", None) is not None and self._res[0].shape[0] >= need:
            occs, offs = self._res
        else:
            occs = np.zeros(need + (need // 8 if reuse else 0), OCC_DTYPE)
            offs = np.zeros(self.n_reads + 1, np.uint64)
            if reuse:
                self._res = (occs, offs)
        cnt = np.zeros(len(COUNTER_NAMES), np.uint64)
        _chk(lib().cmb_batch_results(self.h, _p(occs), occs.shape[0], _p(offs), _p(cnt)))
        return occs[:n.value], offs, dict(zip(COUNTER_NAMES, cnt.tolist()))

    def allow_unsupported(self, on: bool = True):
        """kept for older callers; no effect (reads not longer than the number of parts are matched by naive backtracking on the device)"""
        _chk(lib().cmb_batch_allow_unsupported(self.h, int(on)))

    def read_status(self) -> np.ndarray:
        """per read: bit 0 = matched by naive backtracking instead of a search scheme (searchstrategy.cpp:148-152)"""
        st = np.zeros(max(self.n_reads, 1), np.uint8)
        n = C.c_uint32()
        _chk(lib().cmb_batch_read_status(self.h, _p(st), C.byref(n)))
        return st[:self.n_reads]

    def want_alignments(self, on: bool = True):
        _chk(lib().cmb_batch_want_alignments(self.h, int(on)))

    def alignments(self):
        """(cmb_aln records parallel to the occurrences, pool of CIGAR run-length operations)"""
        n = C.c_uint64()
        _chk(lib().cmb_batch_result_size(self.h, C.byref(n)))
        aln = np.zeros(max(int(n.value), 1), ALN_DTYPE)
        nops = C.c_uint64()
        rc = lib().cmb_batch_alignments(self.h, _p(aln), 0, None, 0, C.byref(nops))  # (sizes first)
        ops = np.zeros(max(int(nops.value), 1), np.uint16)
        _chk(lib().cmb_batch_alignments(self.h, _p(aln), aln.shape[0], _p(ops), ops.shape[0], C.byref(nops)))
        return aln[:n.value], ops[:nops.value]

    def sam(self, ids, quals, seq_names, unmapped: bool = True, xa: bool = False) -> str:
        """SAM text of the chunk (SearchStrategy::generateOutputSingleEnd); needs want_alignments() before run()"""
        def arr(strs):
            a = (C.c_char_p * len(strs))(*[s.encode() for s in strs])
            return a
        ai, aq, an = arr(ids), arr(quals), arr(seq_names)
        buf = self._packed[0]
        n = lib().cmb_batch_sam(self.h, _p(buf), ai, aq, an, int(unmapped), int(xa), None, 0)
        if n < 0:
            raise CmbError(int(n), lib().cmb_last_error().decode(errors="replace"))
        out = C.create_string_buffer(int(n) + 1)
        lib().cmb_batch_sam(self.h, _p(buf), ai, aq, an, int(unmapped), int(xa), out, int(n) + 1)
        return out.value.decode()

    def timings(self) -> Dict[str, float]:
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = lib().cmb_batch_timings(self.h, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.cmb_batch_destroy(self.h)
            self.h = None

    __del__ = close


def match_batch(index: Index, strategy: SearchStrategy, max_distance: int, reads: Sequence[bytes]):
    """``SearchStrategy::matchApprox`` for a whole chunk (src/parallel.cpp:67-78): returns
    (occurrences, per-read offsets, counters)."""
    b = Batch(index, strategy, max_distance, reads)
    try:
        b.run()
        return b.results()
    finally:
        b.close()


def match_best(index, strategy: SearchStrategy, reads: Sequence[bytes], x: int = 0, min_identity: int = 95, kmer_size: int = 10):
    """``SearchStrategy::matchApproxBestPlusX`` for a whole chunk (the reference's default mode): returns
    (occurrences, alignments, CIGAR operations, per-read offsets, best distance per read, hits at that distance, counters).
    ``index``: an ``Index``, or a ``MoveIndex`` with its text attached (``kmer_size``: the k-mer table of that flavour's search)"""
    buf, offs = pack_reads(reads)
    h = C.c_void_p()
    if isinstance(index, MoveIndex):
        _chk(lib().cmb_move_match_best(index.h, strategy.h, x, min_identity, kmer_size, _p(buf), _p(offs), len(reads), C.byref(h)))
    else:
        _chk(lib().cmb_match_best(index.h, strategy.h, x, min_identity, _p(buf), _p(offs), len(reads), C.byref(h)))
    try:
        n, nops = C.c_uint64(), C.c_uint64()
        _chk(lib().cmb_best_sizes(h, C.byref(n), C.byref(nops)))
        occ = np.zeros(max(int(n.value), 1), OCC_DTYPE)
        aln = np.zeros(max(int(n.value), 1), ALN_DTYPE)
        ops = np.zeros(max(int(nops.value), 1), np.uint16)
        o = np.zeros(len(reads) + 1, np.uint64)
        best = np.zeros(max(len(reads), 1), np.uint32)
        hits = np.zeros(max(len(reads), 1), np.uint32)
        cnt = np.zeros(len(COUNTER_NAMES), np.uint64)
        _chk(lib().cmb_best_results(h, _p(occ), _p(aln), occ.shape[0], _p(ops), ops.shape[0], _p(o), _p(best), _p(hits), _p(cnt)))
        return (occ[:n.value], aln[:n.value], ops[:nops.value], o, best[:len(reads)], hits[:len(reads)],
                dict(zip(COUNTER_NAMES, cnt.tolist())))
    finally:
        lib().cmb_best_destroy(h)


def shard_bounds(n_reads: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous read shard of ``rank``: ceil(N/G) reads per rank."""
    per = (n_reads + world_size - 1) // world_size
    lo = min(n_reads, rank * per)
    return lo, min(n_reads, lo + per)


# ---- b-move: the run-length compressed backend (first stage: index structures, extension, locate) ----------------------
MOVE_RANGE_DTYPE = np.dtype([("begin", np.uint64), ("end", np.uint64), ("begin_run", np.uint64), ("end_run", np.uint64),
                             ("rev_begin", np.uint64), ("rev_end", np.uint64), ("rev_begin_run", np.uint64),
                             ("rev_end_run", np.uint64), ("toehold", np.uint64), ("original_depth", np.uint32),
                             ("runs_valid", np.uint8), ("rev_runs_valid", np.uint8), ("toehold_represents_end", np.uint8),
                             ("reserved", np.uint8)])
assert MOVE_RANGE_DTYPE.itemsize == 80  # cmb_move_range
MOVE_OCC_DTYPE = np.dtype([("begin", np.uint64), ("end", np.uint64), ("distance", np.uint32), ("strand", np.uint32)])  # cmb_move_occ


def plcp_runs(plcp: np.ndarray):
    """Run-length form of a PLCP array as cmb_move_desc takes it: the positions where PLCP[q] != PLCP[q - 1] - 1 (0 among
    them) and PLCP[q] + q there."""
    p = np.asarray(plcp).astype(np.int64)
    brk = np.concatenate([[True], p[1:] != p[:-1] - 1])
    pos = np.flatnonzero(brk).astype(np.uint64)
    return pos, (p[brk] + np.flatnonzero(brk)).astype(np.uint64)


class MoveIndex:
    """Device-resident b-move index (cmb_move_index) over the arrays of columba_amd.movebuild.build_move, or any object
    with the same members read from the reference's files."""

    def __init__(self, mv, device: int = 0, with_locate: bool = True, length_bits: int = 64, _handle=None):
        L = lib()
        self.device = device
        if _handle is not None:  # (the receiving side of a broadcast: MoveIndex.empty_like)
            self.h = _handle
            n, r, rr = C.c_uint64(), C.c_uint64(), C.c_uint64()
            _chk(L.cmb_move_info(self.h, C.byref(n), C.byref(r), C.byref(rr)))
            self.n, self.runs, self.rev_runs = n.value, r.value, rr.value
            return
        keep = [np.ascontiguousarray(a) for a in (mv.lfbp_fwd, mv.lfbp_rev)]
        keep += [np.ascontiguousarray(a, dtype=np.uint64) for a in (mv.smpf, mv.smpl, mv.rev_smpf, mv.rev_smpl)]
        d = _MoveDesc()
        d.lfbp, d.lfbp_bytes, d.rev_lfbp, d.rev_lfbp_bytes = keep[0].ctypes.data, keep[0].nbytes, keep[1].ctypes.data, keep[1].nbytes
        d.length_bits = length_bits
        d.samples_first, d.samples_last, d.rev_samples_first, d.rev_samples_last = (a.ctypes.data for a in keep[2:6])
        if with_locate:
            pos, sm = plcp_runs(mv.plcp)
            loc = [np.ascontiguousarray(a, dtype=np.uint64) for a in (mv.pred_first, mv.first_to_run, mv.pred_last, mv.last_to_run, pos, sm)]
            keep += loc
            d.pred_first, d.first_to_run, d.pred_last, d.last_to_run, d.plcp_pos, d.plcp_sum = (a.ctypes.data for a in loc)
            d.n_plcp = pos.shape[0]
        h = C.c_void_p()
        _chk(L.cmb_move_create(C.byref(d), device, C.byref(h)))
        self.h = h
        n, r, rr = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(L.cmb_move_info(self.h, C.byref(n), C.byref(r), C.byref(rr)))
        self.n, self.runs, self.rev_runs = n.value, r.value, rr.value

    def close(self):
        if getattr(self, "h", None):
            lib().cmb_move_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self) -> int:
        return int(lib().cmb_move_device_bytes(self.h))

    def layout(self) -> MoveLayout:
        lay = MoveLayout()
        _chk(lib().cmb_move_layout_of(self.h, C.byref(lay)))
        return lay

    @classmethod
    def empty_like(cls, layout: MoveLayout, device: int = 0) -> "MoveIndex":
        """an index with the arrays of `layout` allocated but not filled (the receiving side of a broadcast)"""
        h = C.c_void_p()
        _chk(lib().cmb_move_create_empty(C.byref(layout), device, C.byref(h)))
        return cls(None, device=device, _handle=h)

    def validate(self):
        """the consistency checks of cmb_move_create on arrays that were filled by a collective"""
        _chk(lib().cmb_move_validate(self.h))

    def device_tensors(self):
        """the device arrays of the index as flat uint8 torch tensors sharing the library's memory (no copy)"""
        import torch
        ptrs, nbytes = (C.c_void_p * MOVE_DEV_ARRAYS)(), (C.c_uint64 * MOVE_DEV_ARRAYS)()
        _chk(lib().cmb_move_device_arrays(self.h, C.byref(ptrs), C.byref(nbytes)))
        return [torch.as_tensor(_DevArray(int(ptrs[i]), int(nbytes[i]), self), device=torch.device("cuda", self.device)) if nbytes[i] else None
                for i in range(MOVE_DEV_ARRAYS)]

    def complete_range(self) -> np.ndarray:
        out = np.zeros(1, dtype=MOVE_RANGE_DTYPE)
        _chk(lib().cmb_move_complete_range(self.h, _p(out)))
        return out

    def rows(self, rev: int) -> np.ndarray:
        cnt = (self.rev_runs if rev else self.runs) + 1
        out = np.zeros((cnt, 4), dtype=np.uint64)
        _chk(lib().cmb_move_rows(self.h, rev, 0, cnt, _p(out)))
        return out

    def extend(self, mode: int, parents: np.ndarray):
        """all four children of every parent: (n x 4 ranges, n x 4 ok flags)"""
        parents = np.ascontiguousarray(parents, dtype=MOVE_RANGE_DTYPE)
        n = parents.shape[0]
        children = np.zeros((n, 4), dtype=MOVE_RANGE_DTYPE)
        ok = np.zeros((n, 4), dtype=np.uint8)
        _chk(lib().cmb_move_extend_batch(self.h, mode, _p(parents), n, _p(children), _p(ok)))
        return children, ok

    def locate(self, ranges: np.ndarray):
        """(positions, offsets): range i holds positions[offsets[i]:offsets[i + 1]] in the reference's order"""
        ranges = np.ascontiguousarray(ranges, dtype=MOVE_RANGE_DTYPE)
        n = ranges.shape[0]
        offs = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(ranges["end"] - ranges["begin"], out=offs[1:])
        pos = np.zeros(int(offs[-1]), dtype=np.uint64)
        _chk(lib().cmb_move_locate_batch(self.h, _p(ranges), n, _p(offs), _p(pos)))
        return pos, offs

    def match_exact(self, reads):
        """k = 0 end to end: (occurrences, per-read offsets, counters); occurrences in the reference's order"""
        buf = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
        offs = np.zeros(len(reads) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(r) for r in reads])
        o = np.zeros(len(reads) + 1, dtype=np.uint64)
        cnt = np.zeros(2, dtype=np.uint64)
        n_occ = C.c_uint64()
        cap = max(1024, 64 * len(reads))  # (a second pass only for chunks with more occurrences than that)
        while True:
            occ = np.zeros(cap, dtype=MOVE_OCC_DTYPE)
            rc = lib().cmb_move_match_exact(self.h, _p(buf), _p(offs), len(reads), _p(occ), cap, _p(o), C.byref(n_occ), _p(cnt))
            if rc == CMB_ERR_OVERFLOW:
                cap = int(n_occ.value)
                continue
            _chk(rc)
            ms = np.zeros(3, dtype=np.float32)
            _chk(lib().cmb_move_last_timings(_p(ms), 3))
            self.last_ms = {"extend": float(ms[0]), "scan": float(ms[1]), "locate": float(ms[2])}
            return occ[:n_occ.value], o, {"NODE_COUNTER": int(cnt[0]), "TOTAL_REPORTED_POSITIONS": int(cnt[1])}

    def kmer_table(self, word_size: int) -> np.ndarray:
        out = np.zeros(4 ** word_size, dtype=MOVE_RANGE_DTYPE)
        _chk(lib().cmb_move_kmer_table(self.h, word_size, _p(out)))
        return out

    def attach_text(self, text, seq_starts=None):
        """the text beside the index (for the alignments of the occurrences: MoveBatch.want_alignments); `text` bytes or uint8
        array with or without the final '$', seq_starts: begin positions of the sequences + the final n - 1, as IndexArrays.seq_starts"""
        t = np.frombuffer(text, np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, np.uint8)
        st = None if seq_starts is None else np.ascontiguousarray(seq_starts, np.uint32)
        _chk(lib().cmb_move_attach_text(self.h, _p(t), t.shape[0], None if st is None else _p(st), 0 if st is None else st.shape[0]))

    def match_batch(self, strategy: "SearchStrategy", max_distance: int, reads, kmer_size: int = 10):
        """``SearchStrategy::matchApprox`` (ALL mode) of the RUN_LENGTH_COMPRESSION flavour for a chunk of reads:
        (occurrences, per-read offsets, counters)"""
        b = MoveBatch(self, strategy, max_distance, reads, kmer_size=kmer_size)
        try:
            b.run()
            return b.results()
        finally:
            b.close()


class MoveBatch:
    """A chunk of reads resident on the device for the b-move search (handle of ``cmb_move_batch_*``)."""

    def __init__(self, index: MoveIndex, strategy: "SearchStrategy", max_distance: int, reads=None, packed=None, kmer_size: int = 10):
        buf, offs = packed if packed is not None else pack_reads(reads)
        self.n_reads = offs.shape[0] - 1
        self.max_distance = max_distance
        self._keep = (index, strategy, buf, offs)
        h = C.c_void_p()
        _chk(lib().cmb_move_batch_create(index.h, strategy.h, max_distance, kmer_size, _p(buf), _p(offs), self.n_reads, C.byref(h)))
        self.h = h

    def run(self):
        _chk(lib().cmb_move_batch_run(self.h))

    def results(self):
        n = C.c_uint64()
        _chk(lib().cmb_move_batch_result_size(self.h, C.byref(n)))
        occ = np.zeros(max(int(n.value), 1), MOVE_OCC_DTYPE)
        offs = np.zeros(self.n_reads + 1, np.uint64)
        cnt = np.zeros(len(COUNTER_NAMES), np.uint64)
        _chk(lib().cmb_move_batch_results(self.h, _p(occ), occ.shape[0], _p(offs), _p(cnt)))
        return occ[:n.value], offs, dict(zip(COUNTER_NAMES, cnt.tolist()))

    def filter_per_strand(self, on: bool = True):
        """every strand of a read filtered by itself (``mapRead``: BEST mode's strata); before run()"""
        _chk(lib().cmb_move_batch_filter_per_strand(self.h, int(on)))

    def want_alignments(self, on: bool = True):
        """CIGAR and sequence of every occurrence (needs MoveIndex.attach_text)"""
        _chk(lib().cmb_move_batch_want_alignments(self.h, int(on)))

    def alignments(self):
        """(cmb_aln records parallel to the occurrences, pool of CIGAR run-length operations), as Batch.alignments"""
        n = C.c_uint64()
        _chk(lib().cmb_move_batch_result_size(self.h, C.byref(n)))
        aln = np.zeros(max(int(n.value), 1), ALN_DTYPE)
        nops = C.c_uint64()
        lib().cmb_move_batch_alignments(self.h, _p(aln), 0, None, 0, C.byref(nops))  # (sizes first)
        ops = np.zeros(max(int(nops.value), 1), np.uint16)
        _chk(lib().cmb_move_batch_alignments(self.h, _p(aln), aln.shape[0], _p(ops), ops.shape[0], C.byref(nops)))
        return aln[:n.value], ops[:nops.value]

    def sam(self, ids, quals, seq_names, unmapped: bool = True, xa: bool = False, metric: str = "edit") -> str:
        """SAM text of the chunk (SearchStrategy::generateOutputSingleEnd) from the batch's occurrences and alignments; needs
        MoveIndex.attach_text and want_alignments() before run()"""
        index, strategy, buf, offs = self._keep
        occ, occ_offs, _ = self.results()
        aln, ops = self.alignments()
        occ32 = np.zeros(max(len(occ), 1), OCC_DTYPE)
        for f in ("begin", "end", "distance", "strand"):
            occ32[f][:len(occ)] = occ[f]
        arr = lambda strs: (C.c_char_p * len(strs))(*[s.encode() for s in strs])
        ai, aq, an = arr(ids), arr(quals), arr(seq_names)
        tix = lib().cmb_move_text_index(index.h)
        args = (tix, self.max_distance, METRIC[metric], _p(buf), _p(offs),
                self.n_reads, ai, aq, an, _p(occ32), _p(occ_offs), _p(aln if len(aln) else np.zeros(1, ALN_DTYPE)),
                _p(ops if len(ops) else np.zeros(1, np.uint16)), int(unmapped), int(xa))
        n = lib().cmb_sam_chunk(*args, None, 0)
        if n < 0:
            raise CmbError(int(n), lib().cmb_last_error().decode(errors="replace"))
        out = C.create_string_buffer(int(n) + 1)
        lib().cmb_sam_chunk(*args, out, int(n) + 1)
        return out.value.decode()

    def timings(self) -> Dict[str, float]:
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = lib().cmb_move_batch_timings(self.h, names, ms, 16)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.cmb_move_batch_destroy(self.h)
            self.h = None

    __del__ = close


def pair_chunk_sam(index: "Index", strategy: "SearchStrategy", max_distance: int, reads1, reads2, ids1, ids2, quals1, quals2, seq_names,
                   orientation: int = ORIENTATION_FR, max_frag: int = 500, min_frag: int = 0, discordant_allowed: bool = True,
                   unmapped_records: bool = True, per_strand: bool = True):
    """A chunk of read pairs in ALL mode, end to end: both mates through the GPU matcher (one batch each, with alignments), then
    cmb_pair_sam per pair (SearchStrategy::pairSingleEndedMatchesAll on the mates' single-end results).  per_strand: the strands
    of a mate are filtered each by itself, as matchApproxPairedEndAll's mapRead does (searchstrategy.cpp:746-776,
    searchstrategy.h:753-774) — the default — instead of together (matchApproxAllMap; the view of pairSingleEndedMatchesAll).
    Occurrences that run past the end of their sequence (cmb_aln.spans) are trimmed and verified again as assignSequence ->
    findSeqName does (indexinterface.cpp:833-899: cmb_trim_occurrence) and paired with their trimmed coordinates, distance and
    CIGAR; those for which that fails are dropped.  Returns (SAM text, number of properly or discordantly mapped pairs)."""
    per_mate = []
    for reads in (reads1, reads2):
        b = Batch(index, strategy, max_distance, reads=reads)
        b.want_alignments()
        if per_strand:
            _chk(lib().cmb_batch_filter_per_strand(b.h, 1))
        b.run()
        occ, offs, _ = b.results()
        aln, ops = b.alignments()
        per_mate.append((occ, offs, aln, ops))
        b.close() if hasattr(b, "close") else None
    starts = index.seq_starts()
    text, mapped_pairs = [], 0
    for i in range(len(reads1)):
        rd = []
        for m, (reads, ids, quals) in enumerate(((reads1, ids1, quals1), (reads2, ids2, quals2))):
            occ, offs, aln, ops = per_mate[m]
            sid, seq, rc, rq = read_prepare(ids[i], reads[i].decode() if isinstance(reads[i], bytes) else reads[i], quals[i])
            lst = []
            for j in range(int(offs[i]), int(offs[i + 1])):
                a = aln[j]
                o = ops[int(a["cigar_off"]):int(a["cigar_off"]) + int(a["cigar_len"])]
                width = int(occ["end"][j]) - int(occ["begin"][j])
                if a["spans"]:
                    oc1 = np.array([tuple(occ[j])], dtype=OCC_DTYPE)
                    al1 = np.array([tuple(a)], dtype=ALN_DTYPE)
                    ops1 = np.zeros(2 * max_distance + 8, np.uint16)
                    nops, found = C.c_uint32(), C.c_int32()
                    pat = (rc if int(occ["strand"][j]) else seq).encode()
                    _chk(lib().cmb_trim_occurrence(index.h, pat, len(pat), max_distance, METRIC["edit"], _p(oc1), _p(al1), _p(ops1), ops1.shape[0],
                                                   C.byref(nops), C.byref(found)))
                    if not found.value:
                        continue  # NOT_FOUND: the occurrence takes no part in the pairing
                    w1 = int(oc1["end"][0]) - int(oc1["begin"][0])
                    lst.append((int(al1["seq_id"][0]), int(al1["seq_begin"][0]), int(al1["seq_begin"][0]) + w1, int(oc1["begin"][0]),
                                int(oc1["distance"][0]), int(oc1["strand"][0]), ops1[:nops.value].copy()))
                else:
                    lst.append((int(a["seq_id"]), int(a["seq_begin"]), int(a["seq_begin"]) + width, int(occ["begin"][j]),
                                int(occ["distance"][j]), int(occ["strand"][j]), o))
            rd.append((sid, seq, rc, quals[i], rq, lst))
        t, n = pair_sam(rd[0], rd[1], seq_names, orientation, max_frag, min_frag, discordant_allowed, unmapped_records)
        text.append(t)
        mapped_pairs += n > 0
    return "".join(text), mapped_pairs


def pair_chunk_sam_best(index: "Index", strategy: "SearchStrategy", reads1, reads2, ids1, ids2, quals1, quals2, seq_names, x: int = 0,
                        min_identity: int = 95, orientation: int = ORIENTATION_FR, max_frag: int = 500, min_frag: int = 0,
                        discordant_allowed: bool = True, unmapped_records: bool = True, max_supported: Optional[int] = None, kmer_size: int = 10,
                        start_from=None):
    """A chunk of read pairs in BEST (+x strata) mode, end to end (SearchStrategy::matchApproxPairedEndBestPlusX,
    searchstrategy.cpp:1091-1179): the pairs walk through their strata together (PairBest); every round, the lists the unfinished pairs
    wait for — mapRead of one mate at one distance — are produced by ONE device batch per (mate, distance) over the reads that ask
    (ALL mode, every strand filtered by itself, with alignments), and both strands of a result are handed in.  `index`: an Index, or a
    MoveIndex with its text attached (the b-move backend: MoveBatch per (mate, distance); kmer_size is its seed table's).  Returns (SAM
    text, number of properly or discordantly mapped pairs, number of device batches).  start_from: the single-end results of
    infer_paired_end_best for this chunk — the pairs start from them (pairSingleEndedMatchesBest; x = 0)."""
    if start_from is not None:
        x = 0
    is_move = isinstance(index, MoveIndex)
    if is_move:
        class _TextIndex:  # (the text beside the b-move index: what an occurrence over a sequence end is trimmed on)
            h = C.c_void_p(lib().cmb_move_text_index(index.h))
        if not _TextIndex.h:
            raise CmbError(-1, "pairs on the b-move index need the text beside it (MoveIndex.attach_text)")
        trim_index = _TextIndex
    else:
        trim_index = index
    n = len(reads1)
    if max_supported is None:  # getMaxSupportedDistanceForBestMapping: the largest k such that 1 .. k all have a scheme (13 at most)
        max_supported = 0
        while max_supported < 13:
            try:
                if strategy.describe(max_supported + 1)[0] == 0:
                    break
            except CmbError:
                break
            max_supported += 1
    mates = []
    for reads, ids, quals in ((reads1, ids1, quals1), (reads2, ids2, quals2)):
        prep = []
        for i in range(n):
            sid, seq, rc, rq = read_prepare(ids[i], reads[i].decode() if isinstance(reads[i], bytes) else reads[i], quals[i])
            prep.append((sid, seq, rc, quals[i], rq))
        mates.append(prep)
    pb = PairBest(mates[0], mates[1], x, min_identity, max_supported, orientation, max_frag, min_frag, discordant_allowed, unmapped_records,
                  text_index=trim_index)
    raw = [[r if isinstance(r, bytes) else r.encode() for r in reads] for reads in (reads1, reads2)]
    if start_from is not None:
        for i in range(n):
            pb.seed(i, start_from["single"][0][i], start_from["single"][1][i], bool(start_from["read2done"][i]))
    batches = 0
    while True:
        req = pb.advance()
        if req.shape[0] == 0:
            break
        groups = {}
        for r in req:
            groups.setdefault((int(r["mate"]), int(r["max_distance"])), []).append(int(r["pair"]))
        for (mate, k), idxs in sorted(groups.items()):
            if is_move:
                b = MoveBatch(index, strategy, k, reads=[raw[mate][i] for i in idxs], kmer_size=kmer_size)
                b.want_alignments()
                b.filter_per_strand()
            else:
                b = Batch(index, strategy, k, reads=[raw[mate][i] for i in idxs])
                b.want_alignments()
                _chk(lib().cmb_batch_filter_per_strand(b.h, 1))
            b.run()
            batches += 1
            occ, offs, _ = b.results()
            aln, ops = b.alignments()
            if is_move:  # (64-bit positions there; alignments exist for texts below 2^32 only)
                occ32 = np.zeros(len(occ), OCC_DTYPE)
                for f in ("begin", "end", "distance", "strand"):
                    occ32[f] = occ[f]
                occ = occ32
            for j, i in enumerate(idxs):
                lo, hi = int(offs[j]), int(offs[j + 1])
                for strand in (0, 1):
                    pb.supply(i, mate, strand, k, occ[lo:hi], aln[lo:hi], ops)
            b.close() if hasattr(b, "close") else None
    text, mapped = [], 0
    for i in range(n):
        t, n_pairs = pb.sam(i, seq_names)
        text.append(t)
        mapped += n_pairs > 0
    pb.close()
    return "".join(text), mapped, batches



def infer_paired_end_best(index: "Index", strategy: "SearchStrategy", reads1, reads2, min_identity: int = 95, seqs_in_first_file: Optional[int] = None,
                          kmer_size: int = 10):
    """The single-end phase that infers the paired-end parameters (parallel.cpp:236-312, :700-727): read 1 of every pair in BEST mode
    (match_best, x = 0); read 2 where read 1 has exactly one match in the first reference file (it moves to the front:
    hasUnambiguousMatchInFirstFile); the pairs whose mates both do are the sample of cmb_pair_infer.  Returns a dict: "inferred"
    (PairInferred), "unambiguous_pairs", "read2done", and "single" — per mate and pair (occ, aln, ops) for
    pair_chunk_sam_best(..., start_from=...)."""
    n = len(reads1)
    lim = 0xFFFFFFFF if seqs_in_first_file is None else seqs_in_first_file
    empty = (np.zeros(0, OCC_DTYPE), np.zeros(0, ALN_DTYPE), np.zeros(0, np.uint16))

    def single(reads, ids):
        out = {}
        if not ids:
            return out
        occ, aln, ops, offs, _best, _hits, _cnt = match_best(index, strategy, [reads[i] for i in ids], x=0, min_identity=min_identity, kmer_size=kmer_size)
        for j, i in enumerate(ids):
            lo, hi = int(offs[j]), int(offs[j + 1])
            o, a = occ[lo:hi].copy(), aln[lo:hi].copy()
            a["spans"] = 0
            first = [q for q in range(hi - lo) if int(a["seq_id"][q]) < lim]
            unambiguous = len(first) == 1
            if unambiguous and first[0] != 0:
                o[[0, first[0]]] = o[[first[0], 0]]
                a[[0, first[0]]] = a[[first[0], 0]]
            out[i] = ((o, a, ops), unambiguous)
        return out

    r1 = single(reads1, list(range(n)))
    second = [i for i in range(n) if r1[i][1]]
    r2 = single(reads2, second)
    samples = []
    for i in second:
        if r2[i][1]:
            (o1, a1, _), (o2, a2, _) = r1[i][0], r2[i][0]
            samples.append((int(a1["seq_begin"][0]), int(a1["seq_begin"][0]) + int(o1["end"][0]) - int(o1["begin"][0]), int(o1["strand"][0]),
                            int(a2["seq_begin"][0]), int(a2["seq_begin"][0]) + int(o2["end"][0]) - int(o2["begin"][0]), int(o2["strand"][0])))
    return {"inferred": pair_infer(samples if samples else np.zeros((0, 6), np.uint32)), "unambiguous_pairs": len(samples),
            "read2done": [i in r2 for i in range(n)],
            "single": ([r1[i][0] for i in range(n)], [r2[i][0] if i in r2 else empty for i in range(n)])}

