"""Replication of the DEVICE layout of an index (columba_amd.dist.broadcast_device_index): two processes share the
one GPU of the test box; rank 0 builds the index, rank 1 receives its device arrays through torch.distributed
(gloo here, which stages device tensors through the host — RCCL refuses two ranks on one device; the calls are the
ones bench.py makes over RCCL) and both match the same reads.  Run with `pytest -m gpu`."""
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
    from columba_amd import indexbuild as ib, synth
    from columba_amd.dist import allreduce_counters, broadcast_device_index
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CMB_BCAST_CHUNK"] = str(1 << 20)   # (the device arrays travel in pieces: columba_amd.dist.broadcast_flat)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, starts = synth.genome_rep(seed=21, n=400_000, scale=2.0)   # (same seed on both ranks: only rank 0 indexes it)
    index = None
    if rank == 0:
        ix = ib.build_index(g.tobytes(), seq_starts=starts, device="cuda:0")
        index = ca.Index(ix, in_text_switch=4, kmer_size=9, device=0)
    index = broadcast_device_index(index, rank, 0)
    lay = index.layout()
    assert int(lay.kmer_size) == 9 and int(lay.text_length) == len(g) + 1
    assert index.seq_starts().tolist() == np.asarray(starts, np.uint32).tolist()
    reads = synth.sample_reads(g, 1500, 150, seed=5, n_frac=0.02)
    occ, offs, cnt = ca.match_batch(index, ca.SearchStrategy("multiple_opt", "edit", "dynamic"), 4, reads)
    np.save(os.path.join(tmp, f"occ{rank}.npy"), occ)
    np.save(os.path.join(tmp, f"offs{rank}.npy"), offs)
    tot = allreduce_counters(cnt, torch.device("cuda", 0))
    assert tot["NODE_COUNTER"] == 2 * cnt["NODE_COUNTER"]
    np.save(os.path.join(tmp, f"cnt{rank}.npy"), np.array([cnt[n] for n in sorted(cnt)]))
    np.save(os.path.join(tmp, f"kmer{rank}.npy"), index.kmer_table())
    dist.barrier()
    dist.destroy_process_group()


def test_device_layout_broadcast_gives_an_identical_index(tmp_path):
    import torch.multiprocessing as mp
    port = 29600 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    o0, o1 = np.load(tmp_path / "occ0.npy"), np.load(tmp_path / "occ1.npy")
    assert len(o0) > 1000 and np.array_equal(o0, o1)
    assert np.array_equal(np.load(tmp_path / "offs0.npy"), np.load(tmp_path / "offs1.npy"))
    assert np.array_equal(np.load(tmp_path / "cnt0.npy"), np.load(tmp_path / "cnt1.npy"))
    assert np.array_equal(np.load(tmp_path / "kmer0.npy"), np.load(tmp_path / "kmer1.npy"))
