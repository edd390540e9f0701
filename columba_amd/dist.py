"""Multi-GPU plumbing: the index is replicated, reads are sharded, nothing is
exchanged on the data path.  One process per GPU with torch.distributed ("nccl" = RCCL over xGMI on
the GPU box; "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import indexbuild as ib

import os

# bytes per collective: arrays of several GB (a 3 Gbp index holds 3 - 6 GB ones) go in pieces (CMB_BCAST_CHUNK: the tests' small pieces)
BCAST_CHUNK = 1 << 30


def broadcast_flat(t: "torch.Tensor", src: int = 0) -> None:
    """broadcast a flat uint8 tensor in pieces of at most BCAST_CHUNK bytes — element counts stay far below 2^31 whatever the
    backend's count type, and the pieces pipeline over the links"""
    import torch.distributed as dist
    n = int(t.numel())
    chunk = int(os.environ.get("CMB_BCAST_CHUNK", BCAST_CHUNK))
    for o in range(0, n, chunk):
        dist.broadcast(t[o:o + chunk], src=src)


INDEX_FIELDS = ["text", "counts", "bv_fwd", "cnt_fwd", "bv_rev", "cnt_rev", "bwt_words", "sa_bv",
                "sa_bv_counts", "sa_samples", "seq_starts"]


def broadcast_index(ix: Optional[ib.IndexArrays], rank: int, dev) -> ib.IndexArrays:
    """Replicate the index arrays held by rank 0 on every rank (one broadcast per array)."""
    import torch.distributed as dist
    meta = [None]
    if rank == 0:
        meta = [{"shapes": {f: (tuple(getattr(ix, f).shape), str(getattr(ix, f).dtype)) for f in INDEX_FIELDS},
                 "dpf": ix.dollar_pos_fwd, "dpr": ix.dollar_pos_rev, "sparseness": ix.sparseness,
                 "names": ix.seq_names}]
    dist.broadcast_object_list(meta, src=0)
    m = meta[0]
    arrays = {}
    for f in INDEX_FIELDS:
        shape, dt = m["shapes"][f]
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        if rank == 0:
            t = torch.from_numpy(np.ascontiguousarray(getattr(ix, f)).view(np.uint8).reshape(-1).copy()).to(dev)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if nbytes:  # (an absent array, e.g. the BWT words the device does not need, has no elements)
            broadcast_flat(t, src=0)
        arrays[f] = getattr(ix, f) if rank == 0 else t.cpu().numpy().view(dt).reshape(shape)
        del t
    if rank == 0:
        return ix
    return ib.IndexArrays(text=arrays["text"], counts=arrays["counts"], dollar_pos_fwd=m["dpf"],
                          bv_fwd=arrays["bv_fwd"], cnt_fwd=arrays["cnt_fwd"], dollar_pos_rev=m["dpr"],
                          bv_rev=arrays["bv_rev"], cnt_rev=arrays["cnt_rev"], bwt_words=arrays["bwt_words"],
                          sa_bv=arrays["sa_bv"], sa_bv_counts=arrays["sa_bv_counts"],
                          sa_samples=arrays["sa_samples"], sparseness=m["sparseness"],
                          seq_starts=arrays["seq_starts"], seq_names=m["names"])


def broadcast_device_index(index, rank: int, device: int = 0):
    """Replicate the DEVICE layout of rank 0's index on every rank: the 32-byte rank blocks (the forward ones carry the
    sampled-row bits), samples, text codes, 2-bit text and k-mer table are broadcast straight into the arrays of an empty twin index
    (one collective per array, 12.8 GB for a 3 Gbp reference; per-link bound on xGMI) — no host round trip and no
    second re-layout.  `index` is a columba_amd.Index on rank 0 and ignored elsewhere."""
    import ctypes as C
    import torch.distributed as dist
    from . import Index, IndexLayout
    meta = [None]
    if rank == 0:
        meta = [(bytes(index.layout()), index.seq_starts())]
    dist.broadcast_object_list(meta, src=0)
    raw, starts = meta[0]
    if rank != 0:
        lay = IndexLayout.from_buffer_copy(raw)
        index = Index.empty_like(lay, starts, device)
    for t in index.device_tensors():
        if t is not None:
            broadcast_flat(t, src=0)
    # every replica runs the consistency probe of cmb_index_create on what arrived (a truncated or mixed-up transfer would
    # otherwise hang the first locate) — and all ranks learn the outcome before any of them enters the next collective
    agree_on_replicas(index if rank != 0 else None)
    return index


def agree_on_replicas(replica, what: str = "index"):
    """Validate this rank's replica (None: nothing to validate here, e.g. the rank that owns the original) and make EVERY rank
    raise if ANY replica failed: a rank that raised on its own would leave the others waiting in the next collective."""
    import torch.distributed as dist
    err = ""
    if replica is not None:
        try:
            replica.validate()
        except Exception as e:  # noqa: BLE001
            err = str(e) or type(e).__name__
    errs = [None] * dist.get_world_size()
    dist.all_gather_object(errs, err)
    bad = [(r, e) for r, e in enumerate(errs) if e]
    if bad:
        raise RuntimeError(f"{what} replica failed validation on rank(s) " + ", ".join(f"{r}: {e}" for r, e in bad))


MOVE_FIELDS = ["lfbp_fwd", "lfbp_rev", "smpf", "smpl", "rev_smpf", "rev_smpl", "pred_first", "first_to_run", "pred_last",
               "last_to_run", "plcp", "sa", "rev_sa", "text"]


def broadcast_move_arrays(mv, rank: int, dev):
    """Replicate the parts of a b-move index (columba_amd.movebuild.MoveArrays: the two .LFBP files, samples, locate arrays)
    held by rank 0 on every rank, one broadcast per array; every rank then creates its own columba_amd.MoveIndex from them
    (cmb_move_create converts and checks the tables on its GPU).  The index is read-only and replicated, as the FM-index is."""
    import torch.distributed as dist
    from .movebuild import MoveArrays
    meta = [None]
    if rank == 0:
        meta = [{"n": mv.n, "shapes": {f: (tuple(getattr(mv, f).shape), str(getattr(mv, f).dtype)) for f in MOVE_FIELDS}}]
    dist.broadcast_object_list(meta, src=0)
    m = meta[0]
    arrays = {}
    for f in MOVE_FIELDS:
        shape, dt = m["shapes"][f]
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        if rank == 0:
            t = torch.from_numpy(np.ascontiguousarray(getattr(mv, f)).view(np.uint8).reshape(-1).copy()).to(dev)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if nbytes:
            broadcast_flat(t, src=0)
        arrays[f] = getattr(mv, f) if rank == 0 else t.cpu().numpy().view(dt).reshape(shape)
        del t
    return mv if rank == 0 else MoveArrays(n=m["n"], **arrays)


def broadcast_device_move_index(index, rank: int, device: int = 0):
    """Replicate the DEVICE layout of rank 0's b-move index on every rank: the 16-byte move rows of both directions, the
    samples and the locate arrays with their directories are broadcast straight into the arrays of an empty twin
    (cmb_move_layout_of / cmb_move_create_empty / cmb_move_device_arrays) — no second conversion of the .LFBP rows, no host
    round trip.  `index` is a columba_amd.MoveIndex on rank 0 and ignored elsewhere; replicas are validated on arrival."""
    import torch.distributed as dist
    from . import MoveIndex, MoveLayout
    meta = [None]
    if rank == 0:
        meta = [bytes(index.layout())]
    dist.broadcast_object_list(meta, src=0)
    if rank != 0:
        index = MoveIndex.empty_like(MoveLayout.from_buffer_copy(meta[0]), device)
    for t in index.device_tensors():
        if t is not None:
            broadcast_flat(t, src=0)
    agree_on_replicas(index if rank != 0 else None, "b-move index")  # (as for the FM-index: all ranks learn the outcome)
    return index


def _wire(dev):
    """the device collectives / point-to-point transfers run on: the GPU with RCCL ("nccl"); host memory with gloo,
    which serves scatter and send/recv for CPU tensors only (CPU tests, single-GPU rehearsals of the N > 1 path)"""
    import torch.distributed as dist
    return "cpu" if dist.get_backend() == "gloo" else dev


def allreduce_counters(cnt: dict, dev) -> dict:
    """Counters of the whole job = sum over the ranks (the reference's writer thread merges the per-chunk Counters
    the same way, src/fastq.cpp:643)."""
    import torch.distributed as dist
    names = sorted(cnt)
    dev = _wire(dev)
    t = torch.tensor([int(cnt[n]) for n in names], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    return dict(zip(names, t.cpu().tolist()))


def gather_occurrences(occ: np.ndarray, offs: np.ndarray, rank: int, world: int, dev):
    """gatherv of the per-rank occurrence lists on rank 0: every rank announces its sizes
    (all-gather), then sends its records and per-read counts point-to-point; rank 0 receives every shard into its
    slice of one buffer.  Shards are contiguous slices of the global read batch in rank order (shard_bounds), so the
    result is the occurrence list of the whole batch in read order with rebased offsets.
    Returns (occurrences, offsets) on rank 0 and (None, None) elsewhere."""
    import torch.distributed as dist
    from . import OCC_DTYPE
    dev = _wire(dev)
    n_occ, n_reads = int(occ.shape[0]), int(offs.shape[0]) - 1
    sizes = torch.tensor([n_occ, n_reads], dtype=torch.int64, device=dev)
    allsz = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allsz, sizes)
    allsz = [tuple(int(x) for x in t.cpu().tolist()) for t in allsz]
    rec = OCC_DTYPE.itemsize
    mine_occ = torch.from_numpy(np.ascontiguousarray(occ).view(np.uint8).reshape(-1).copy()).to(dev)
    counts = np.diff(offs.astype(np.int64)).astype(np.int64)
    mine_cnt = torch.from_numpy(counts).to(dev)
    if rank != 0:
        ops = []
        if n_occ:
            ops.append(dist.P2POp(dist.isend, mine_occ, 0))
        if n_reads:
            ops.append(dist.P2POp(dist.isend, mine_cnt, 0))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return None, None
    tot_occ, tot_reads = sum(a for a, _ in allsz), sum(b for _, b in allsz)
    all_occ = torch.empty(tot_occ * rec, dtype=torch.uint8, device=dev)
    all_cnt = torch.empty(tot_reads, dtype=torch.int64, device=dev)
    all_occ[:n_occ * rec] = mine_occ
    all_cnt[:n_reads] = mine_cnt
    ops, o, r = [], n_occ, n_reads
    for src in range(1, world):
        a, b = allsz[src]
        if a:
            ops.append(dist.P2POp(dist.irecv, all_occ[o * rec:(o + a) * rec], src))
        if b:
            ops.append(dist.P2POp(dist.irecv, all_cnt[r:r + b], src))
        o += a
        r += b
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    out_offs = np.zeros(tot_reads + 1, np.uint64)
    out_offs[1:] = np.cumsum(all_cnt.cpu().numpy()).astype(np.uint64)
    return all_occ.cpu().numpy().view(OCC_DTYPE), out_offs


def scatter_reads(all_reads: Optional[torch.Tensor], per_rank_bytes: int, rank: int, world: int, dev) -> np.ndarray:
    """Rank 0 holds `world` equal shards ([world, per_rank_bytes] uint8); every rank gets its own."""
    import torch.distributed as dist
    dev = _wire(dev)
    shard = torch.empty(per_rank_bytes, dtype=torch.uint8, device=dev)
    if rank == 0:
        dist.scatter(shard, [all_reads[i].contiguous().to(dev) for i in range(world)], src=0)
    else:
        dist.scatter(shard, None, src=0)
    return shard.cpu().numpy()
