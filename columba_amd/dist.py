"""Multi-GPU plumbing: the index is replicated, reads are sharded, nothing is
exchanged on the data path.  One process per GPU with torch.distributed ("nccl" = RCCL over xGMI on
the GPU box; "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import indexbuild as ib

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
            dist.broadcast(t, src=0)
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


def scatter_reads(all_reads: Optional[torch.Tensor], per_rank_bytes: int, rank: int, world: int, dev) -> np.ndarray:
    """Rank 0 holds `world` equal shards ([world, per_rank_bytes] uint8); every rank gets its own."""
    import torch.distributed as dist
    shard = torch.empty(per_rank_bytes, dtype=torch.uint8, device=dev)
    if rank == 0:
        dist.scatter(shard, [all_reads[i].contiguous() for i in range(world)], src=0)
    else:
        dist.scatter(shard, None, src=0)
    return shard.cpu().numpy()
