"""Replication of the DEVICE layout of a b-move index (columba_amd.dist.broadcast_device_move_index): two processes share the
one GPU of the test box; rank 0 creates the index from the reference's files, rank 1 receives its device arrays through
torch.distributed (gloo here, which stages device tensors through the host — RCCL refuses two ranks on one device) and
validates them; both answer the same reads.  Run with `pytest -m gpu`."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import columba_amd as ca
    from columba_amd import movebuild
    from columba_amd.dist import broadcast_device_move_index
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(31)                     # (same seed on both ranks: only rank 0 indexes the text)
    base = rng.integers(0, 4, 8000)
    parts = []
    for _ in range(10):
        s = base.copy()
        m = rng.random(base.shape[0]) < 0.01
        s[m] = rng.integers(0, 4, int(m.sum()))
        parts.append(s)
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(parts)].tobytes()
    index = ca.MoveIndex(movebuild.build_move(text, device="cuda"), device=0) if rank == 0 else None
    index = broadcast_device_move_index(index, rank, 0)
    assert index.n == len(text) + 1 and int(index.layout().has_locate) == 1
    reads = [text[p:p + 60] for p in range(0, 60000, 37)]
    occ, offs, cnt = index.match_exact(reads)
    np.save(os.path.join(tmp, f"mocc{rank}.npy"), occ)
    np.save(os.path.join(tmp, f"moffs{rank}.npy"), offs)
    np.save(os.path.join(tmp, f"mcnt{rank}.npy"), np.array([cnt["NODE_COUNTER"], cnt["TOTAL_REPORTED_POSITIONS"]]))
    # the approximate search on the replica: every rank matches ITS shard of the reads (sharded job: no collective on the
    # data path), rank 0's index was created from the files, rank 1's arrived through the collective
    areads = [text[p:p + 100].replace(b"ACG", b"ATG", 1) for p in range(50, 70000, 53)]
    lo, hi = ca.shard_bounds(len(areads), world, rank)
    a_occ, a_offs, a_cnt = index.match_batch(ca.SearchStrategy("multiple_opt", "edit", "dynamic"), 4, areads[lo:hi], kmer_size=6)
    np.save(os.path.join(tmp, f"aocc{rank}.npy"), a_occ)
    np.save(os.path.join(tmp, f"aoffs{rank}.npy"), a_offs)
    if rank == 1:  # ... and the whole chunk on the replica, for the comparison with the concatenated shards
        w_occ, w_offs, _ = index.match_batch(ca.SearchStrategy("multiple_opt", "edit", "dynamic"), 4, areads, kmer_size=6)
        np.save(os.path.join(tmp, "awhole_occ.npy"), w_occ)
        np.save(os.path.join(tmp, "awhole_offs.npy"), w_offs)
    np.save(os.path.join(tmp, f"mkmer{rank}.npy"), index.kmer_table(5))
    np.save(os.path.join(tmp, f"mrows{rank}.npy"), index.rows(1))
    if rank == 1:  # a replica whose arrays are damaged afterwards is refused by the validation
        t = index.device_tensors()[0]
        t[16 * 7: 16 * 8] = t[16 * 3: 16 * 4].clone()
        try:
            index.validate()
            ok = False
        except ca.CmbError as e:
            ok = e.code == ca.CMB_ERR_INVALID
        np.save(os.path.join(tmp, "mrefused.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_device_layout_broadcast_of_the_move_index(tmp_path):
    import torch.multiprocessing as mp
    port = 29700 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    o0, o1 = np.load(tmp_path / "mocc0.npy"), np.load(tmp_path / "mocc1.npy")
    assert len(o0) > 5000 and np.array_equal(o0, o1)
    for name in ("moffs", "mcnt", "mkmer", "mrows"):
        assert np.array_equal(np.load(tmp_path / f"{name}0.npy"), np.load(tmp_path / f"{name}1.npy")), name
    assert bool(np.load(tmp_path / "mrefused.npy")[0])
    # read shards matched on the two replicas, concatenated in rank order = the whole chunk matched on one
    a0, a1, w = np.load(tmp_path / "aocc0.npy"), np.load(tmp_path / "aocc1.npy"), np.load(tmp_path / "awhole_occ.npy")
    assert len(w) > 1000 and np.array_equal(np.concatenate([a0, a1]), w)
    f0, f1, fw = np.load(tmp_path / "aoffs0.npy"), np.load(tmp_path / "aoffs1.npy"), np.load(tmp_path / "awhole_offs.npy")
    assert np.array_equal(np.concatenate([f0, f1[1:] + f0[-1]]), fw)
