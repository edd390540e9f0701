"""Quick GPU sanity/timing run (development helper, not the judged bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import columba_amd as ca
from columba_amd import indexbuild as ib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
t = time.time()
g, starts = synth.genome_human_like(n, seed=2025, device="cuda")
print("genome", round(time.time() - t, 2)); t = time.time()
ix = ib.build_index(g, seq_starts=starts, device="cuda", with_bwt=False)
print("index build", round(time.time() - t, 2), "s;", ix.nbytes() / 1e6, "MB"); t = time.time()
del g
torch.cuda.empty_cache()  # (the builder's temporaries would otherwise stay with torch's caching allocator)
dev = ca.Index(ix)
print("upload+kmer", round(time.time() - t, 2)); t = time.time()
buf, offs = synth.sample_reads_fast(ix.text[:-1], nreads, 150, seed=3, device="cuda")
reads = None
torch.cuda.empty_cache()
print("reads", round(time.time() - t, 2))
for spec, metric, k in (("multiple_opt", "edit", 4), ("kuch1", "edit", 0)):
    st = ca.SearchStrategy(spec, metric, "dynamic")
    b = ca.Batch(dev, st, k, packed=(buf, offs))
    for it in range(2):
        t = time.time(); b.run(); dt = time.time() - t
    occ, occ_offs, cnt = b.results()
    print(spec, metric, k, "reads/s", round(nreads / dt), "wall", round(dt, 3), "occ", len(occ))
    print("  timings", {k_: round(v, 2) for k_, v in b.timings().items()})
    print("  counters", cnt)
    b.close()
# extend microbench: random ranges
N = 1 << 24
rng = np.random.default_rng(1)
b_ = rng.integers(0, ix.n, N)
w = np.minimum((2.0 ** rng.uniform(0, np.log2(ix.n), N)).astype(np.int64), ix.n - b_)
b2 = rng.integers(0, ix.n, N)
r = np.stack([b_, b_ + w, b2, np.minimum(b2 + w, ix.n)], axis=1).astype(np.uint32)
din = torch.from_numpy(r.view(np.int32)).cuda()
dout = torch.empty((N, 16), dtype=torch.int32, device="cuda")
dok = torch.empty((N, 4), dtype=torch.uint8, device="cuda")
import ctypes as C
ms = C.c_float()
for mode in (0, 1, 2):
    rc = ca.lib().cmb_extend_bench(dev.h, mode, din.data_ptr(), N, dout.data_ptr(), dok.data_ptr(), 10, C.byref(ms))
    assert rc == 0
    print("extend mode", mode, "ms", round(ms.value, 3), "pairs/s %.3g" % (N / ms.value * 1e3), "algorithmic GB/s", round(192 * N / ms.value / 1e6, 1))
