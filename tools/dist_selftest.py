"""Development helper: exercise the RCCL code path of bench.py (broadcast_index / scatter_reads) with ONE rank on
the GPU (world sizes > 1 need a multi-GPU node; the two-rank logic is covered with gloo in tests/test_dist_gloo.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
from columba_amd import indexbuild as ib, synth
from columba_amd.dist import broadcast_index, scatter_reads
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g, starts = synth.genome_human_like(64_000_000, seed=1, device=dev)
ix = ib.build_index(g, seq_starts=starts, device=dev, with_bwt=False)
ix2 = broadcast_index(ix, 0, dev)
assert ix2 is ix
buf, _ = synth.sample_reads_fast(torch.from_numpy(ix.text[:-1]).to(dev), 100000, 150, seed=3, device=dev)
allr = torch.from_numpy(buf).to(dev).reshape(1, -1)
shard = scatter_reads(allr, allr.shape[1], 0, 1, dev)
assert np.array_equal(shard, buf)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); dist.destroy_process_group()
print("rccl single-rank self-test ok")
