"""N > 1 plumbing on CPU: two gloo ranks replicate the index (broadcast), receive their read shard
(scatter) and match it; the union of the shards equals a single-process run.  The matching itself is
done by the CPU oracle here (there is no GPU in the CPU suite) — what is under test is the sharding."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import columba_amd as ca
    from columba_amd import indexbuild as ib, synth
    from columba_amd.dist import allreduce_counters, broadcast_index, gather_occurrences, scatter_reads
    import oracle_py as op
    import schemes_py as sp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CMB_BCAST_CHUNK"] = "65536"   # (arrays travel in pieces: columba_amd.dist.broadcast_flat)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    R, L = 150, 100
    ix = None
    allr = None
    if rank == 0:
        g, starts = synth.genome_rep(seed=5, n=120_000, scale=2.0)
        ix = ib.build_index(g.tobytes(), seq_starts=starts)
        reads = synth.sample_reads(g, R * world, L, seed=2)
        allr = torch.from_numpy(np.frombuffer(b"".join(reads), dtype=np.uint8).copy()).reshape(world, R * L)
    ix = broadcast_index(ix, rank, "cpu")
    buf = scatter_reads(allr, R * L, rank, world, "cpu")
    lo, hi = ca.shard_bounds(R * world, world, rank)
    assert hi - lo == R
    reads = [buf[i * L:(i + 1) * L].tobytes() for i in range(R)]
    occ, offs, cnt = op.match_batch(op.OracleIndex(ix), op.OracleStrategy(sp.MULTIPLE_OPT), 2, reads)
    tot = torch.tensor([len(occ), int(occ["begin"].astype(np.int64).sum())], dtype=torch.int64)
    dist.all_reduce(tot)
    np.save(os.path.join(tmp, f"tot{rank}.npy"), tot.numpy())
    # result path of the sharded job: gatherv of the occurrence lists on rank 0, all-reduce of the counters
    occ2 = np.zeros(len(occ), ca.OCC_DTYPE)
    for f in ("begin", "end", "distance", "strand"):
        occ2[f] = occ[f]
    g_occ, g_offs = gather_occurrences(occ2, offs, rank, world, "cpu")
    g_cnt = allreduce_counters(cnt, "cpu")
    if rank == 0:
        np.save(os.path.join(tmp, "g_occ.npy"), g_occ)
        np.save(os.path.join(tmp, "g_offs.npy"), g_offs)
        np.save(os.path.join(tmp, "g_cnt.npy"), np.array([g_cnt["NODE_COUNTER"], g_cnt["EXPANSIONS"], g_cnt["TOTAL_REPORTED_POSITIONS"]]))
    else:
        assert g_occ is None and g_offs is None
    if rank == 0:
        np.save(os.path.join(tmp, "reads.npy"), allr.numpy())
        ib.save_index(ix, os.path.join(tmp, "idx"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path, oracle_built):
    world = 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    import schemes_py as sp
    from columba_amd import indexbuild as ib
    ix = ib.load_index(str(tmp_path / "idx"))
    allr = np.load(tmp_path / "reads.npy").reshape(-1)
    reads = [allr[i * 100:(i + 1) * 100].tobytes() for i in range(300)]
    occ, offs, cnt = op.match_batch(op.OracleIndex(ix), op.OracleStrategy(sp.MULTIPLE_OPT), 2, reads)
    t0, t1 = np.load(tmp_path / "tot0.npy"), np.load(tmp_path / "tot1.npy")
    assert np.array_equal(t0, t1)
    assert t0[0] == len(occ) and t0[1] == int(occ["begin"].astype(np.int64).sum()) and len(occ) > 0
    # the gathered list IS the single-process list: same records in read order, same offsets, summed counters
    g_occ, g_offs = np.load(tmp_path / "g_occ.npy"), np.load(tmp_path / "g_offs.npy")
    assert np.array_equal(g_offs, offs)
    for f in ("begin", "end", "distance", "strand"):
        assert np.array_equal(g_occ[f], occ[f]), f
    assert np.load(tmp_path / "g_cnt.npy").tolist() == [cnt["NODE_COUNTER"], cnt["EXPANSIONS"], cnt["TOTAL_REPORTED_POSITIONS"]]


def test_shard_bounds_cover_everything():
    import columba_amd as ca
    for n in (0, 1, 7, 8, 9, 1000, 1001):
        for w in (1, 2, 3, 8):
            spans = [ca.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _move_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import columba_amd as ca
    from columba_amd import movebuild
    from columba_amd.dist import broadcast_move_arrays
    import oracle_py as op
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CMB_BCAST_CHUNK"] = "65536"   # (arrays travel in pieces: columba_amd.dist.broadcast_flat)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(9)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[np.tile(rng.integers(0, 4, 700), 6)]
    text = text.copy()
    text[rng.integers(0, text.shape[0], 40)] = ord("A")
    mv = movebuild.build_move(text.tobytes()) if rank == 0 else None
    mv = broadcast_move_arrays(mv, rank, "cpu")
    # every rank matches its shard of the reads on its replica (the oracle stands in for the GPU in the CPU suite)
    t = mv.text.tobytes()
    reads = [t[p:p + 25] for p in range(0, 2000, 10)]
    lo, hi = ca.shard_bounds(len(reads), world, rank)
    occ, offs, cnt = op.OracleMoveIndex(mv).match_exact(reads[lo:hi])
    np.save(os.path.join(tmp, f"mv_occ{rank}.npy"), occ)
    np.save(os.path.join(tmp, f"mv_cnt{rank}.npy"), np.array([cnt["NODE_COUNTER"], cnt["TOTAL_REPORTED_POSITIONS"], lo, hi]))
    if rank == 0:
        o_all, _, c_all = op.OracleMoveIndex(mv).match_exact(reads)
        np.save(os.path.join(tmp, "mv_all.npy"), o_all)
        np.save(os.path.join(tmp, "mv_call.npy"), np.array([c_all["NODE_COUNTER"], c_all["TOTAL_REPORTED_POSITIONS"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_move_index_replication(tmp_path, oracle_built):
    """the b-move index parts broadcast from rank 0: both replicas answer their read shards; shards in rank order = one run"""
    world = 2
    mp.spawn(_move_worker, args=(world, 29519, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"mv_occ{r}.npy") for r in range(world)]
    cnts = [np.load(tmp_path / f"mv_cnt{r}.npy") for r in range(world)]
    assert np.array_equal(np.concatenate(parts), np.load(tmp_path / "mv_all.npy")) and sum(p.shape[0] for p in parts) > 500
    assert [int(sum(c[i] for c in cnts)) for i in (0, 1)] == [int(v) for v in np.load(tmp_path / "mv_call.npy")]
    assert int(cnts[0][3]) == int(cnts[1][2])


class _Replica:
    """stands in for a device index replica: validate() fails on the rank that got a damaged copy"""

    def __init__(self, bad):
        self.bad = bad

    def validate(self):
        if self.bad:
            raise RuntimeError("row 17 of the move table points outside its run")


def _agree_worker(rank, world, port, tmp, bad_rank):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from columba_amd.dist import agree_on_replicas
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    msg = ""
    try:
        # rank 0 owns the original (nothing to validate), the others their replicas — as broadcast_device_[move_]index call it
        agree_on_replicas(None if rank == 0 else _Replica(rank == bad_rank), "b-move index")
    except RuntimeError as e:
        msg = str(e)
    # nobody is left behind in a collective: the next one completes on every rank
    t = torch.ones(1)
    dist.all_reduce(t)
    with open(os.path.join(tmp, f"agree{rank}.txt"), "w") as f:
        f.write(f"{int(t.item())}|{msg}")
    dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank", [1, 2, -1])
def test_a_bad_replica_makes_every_rank_raise(tmp_path, bad_rank):
    """broadcast_device_index / broadcast_device_move_index: one replica that fails validation is an error on ALL ranks (a rank
    that raised alone would leave the others hanging in the next collective); no bad replica, no error"""
    world = 3
    mp.spawn(_agree_worker, args=(world, 29600 + os.getpid() % 300 + 2 * (bad_rank + 1), str(tmp_path), bad_rank), nprocs=world, join=True)
    for r in range(world):
        n, msg = open(tmp_path / f"agree{r}.txt").read().split("|", 1)
        assert int(n) == world
        if bad_rank < 0:
            assert msg == ""
        else:
            assert "b-move index replica failed validation" in msg and f"{bad_rank}: row 17" in msg, (r, msg)
